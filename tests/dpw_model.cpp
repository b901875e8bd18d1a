// Host model of the wave-batch connection scorer (pyrodigal_amd/csrc/dp_wave.hip): the same decomposition -- 64 targets at
// a time, block maxima / in-block prefix and suffix maxima for the far gene ends, uniform carries for the forward-stop and
// reverse-stop relations, near steps, candidate chains, the in-batch walk -- with loops over the 64 lanes where the kernel has a
// wavefront.  It shares every scalar routine with the kernel (dpw_core.h).  TEST INFRASTRUCTURE: built and run by
// tests/test_dpw_model.py against the CPU oracle; nothing in the product links it.
#include "../pyrodigal_amd/csrc/dpw_core.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace {

const double NEG_INF = -__builtin_huge_val();

struct Lex { double v; int i; };
inline void lex_take(Lex& a, double v, int i) { if (v > a.v || (v == a.v && i > a.i)) { a.v = v; a.i = i; } }

struct Chain {
    int n;
    std::vector<uint8_t> kf; std::vector<int32_t> lo, q1, q2;
    std::vector<double> cs; std::vector<DpwExt> ext;
    const int32_t* ndx; const int32_t* stopv;
    // results
    std::vector<double> score; std::vector<int32_t> traceb, tbn; std::vector<int8_t> ov;
    std::vector<double> sfxv; std::vector<int32_t> sfxi;
};

DpwT load_target(const Chain& C, const int i, const double negc) {
    DpwT T; memset(&T, 0, sizeof T);
    T.dlo0 = T.dlo1 = T.dlo2 = INT_MAX; T.dhi0 = T.dhi1 = T.dhi2 = INT_MIN;
    T.i = i < C.n ? i : -1;
    if (T.i < 0) { T.lo = INT_MAX; T.kind = -1; return T; }
    const int kf = C.kf[i];
    T.kind = DPW_KIND(kf); T.frame = DPW_FRAME(kf); T.ndx = C.ndx[i]; T.stop_val = C.stopv[i]; T.lo = C.lo[i]; T.q1 = C.q1[i]; T.q2 = C.q2[i];
    T.cs = C.cs[i]; T.csd = T.cs + negc;
    if (T.kind & 1) {
        const DpwExt& e = C.ext[i];
        T.vm = e.vm; T.x0 = e.x[0]; T.x1 = e.x[1]; T.x2 = e.x[2];
        T.dlo0 = e.dlo[0]; T.dlo1 = e.dlo[1]; T.dlo2 = e.dlo[2]; T.dhi0 = e.dhi[0]; T.dhi1 = e.dhi[1]; T.dhi2 = e.dhi[2];
        T.cq0 = e.cq[0]; T.cq1 = e.cq[1]; T.cq2 = e.cq[2];
    }
    return T;
}

// a final node as a source, read back from memory
DpwS load_source(const Chain& C, const int j) {
    DpwS S; memset(&S, 0, sizeof S);
    const int kf = C.kf[j];
    S.j = j; S.kind = DPW_KIND(kf); S.frame = DPW_FRAME(kf); S.ndx = C.ndx[j]; S.stop_val = C.stopv[j];
    S.tbn = C.tbn[j]; S.score = C.score[j]; S.cs = C.cs[j];
    if (S.kind == 1) { const DpwExt& e = C.ext[j]; S.vm = e.vm; S.x0 = e.x[0]; S.x1 = e.x[1]; S.x2 = e.x[2]; }
    return S;
}


// ---- step schedule (dpw_core.h "Step schedule", round 6): per NODE two pairs of 64-bit words -- which lanes of its own batch (W0, W1)
// and of the batch behind it (N0, N1) it reaches as a source.  The host loop below builds the words from dpw_static_bits and takes them
// apart again with the kind masks of the target batch exactly as the kernel's assembly does (tools/gen_dpw_walk.py); k_dpw_sched
// builds the same words on the device.
struct Entry { int lane, s_ndx; uint32_t code; int j; uint64_t m[6]; };
struct BatchSched { bool miss; std::vector<Entry> near, walk; };
struct NodeWords { uint64_t w0, w1, n0, n1; };

// words of source j towards the 64 targets T[] (in_batch: a node of the batch itself, lane `lane`): r0 = every lane it reaches by any
// relation (for a forward stop of the batch also the forward starts it pulls, which sit BEFORE it), r1 = the lanes whose distance term
// comes from the table (R5 source: reverse stops nearby; F3 source: forward starts nearby)
void words_of(const Chain& C, const DpwST* T, const int j, const int lane, const bool in_batch, uint64_t& r0, uint64_t& r1) {
    r0 = r1 = 0;
    const int kf = C.kf[j], sk = DPW_KIND(kf), sf = DPW_FRAME(kf);
    if (sk == 0) return;
    const int s_ndx = C.ndx[j], s_stop = C.stopv[j];
    for (int t = 0; t < 64; t++) {
        const unsigned bits = dpw_static_bits(T[t], j, sk, sf, s_ndx, s_stop);
        if (sk == 3 ? bits != 0u : (bits & ~2u) != 0u) r0 |= 1ull << t;
        if ((sk == 2 || sk == 1) && (bits & 2u)) r1 |= 1ull << t;
        if (in_batch && sk == 1 && t < lane && dpw_static_pull(T[t], sf, s_stop)) r0 |= 1ull << t;
    }
    // A reverse start WITHOUT a W1 takes the walk's shortcut (entry_from): every gene begin behind it, no word read.  Where that is not its
    // W0 -- a reverse stop one or two bases on, which it does not reach (ref: _connection.h:337-342) -- the lane goes into W1 (outside W0:
    // priced by the table, masked out by W0), which sends the step down the path that reads the words.
    if (sk == 2 && !getenv("DPW_MODEL_NO_PLAIN_FIX"))
        for (int t = in_batch ? lane + 1 : 0; t < 64; t++)
            if (T[t].i >= 0 && (T[t].kind == 0 || T[t].kind == 3) && !((r0 >> t) & 1ull)) r1 |= 1ull << t;
}
// and back: the masks of a step from the words and the kinds of the target lanes (k[q] = lanes of kind q)
Entry entry_from(const Chain& C, const int j, const int lane, const uint64_t r0, const uint64_t r1, const uint64_t* k, const bool in_batch) {
    Entry e; memset(&e, 0, sizeof e);
    const int kf = C.kf[j], sk = DPW_KIND(kf), sf = DPW_FRAME(kf);
    e.lane = lane; e.code = DPW_E_CODE(sk, sf); e.s_ndx = C.ndx[j]; e.j = j;
    if (sk == 2) {
        // as the assembly (tools/gen_dpw_walk.py, "R5 (plain)"): without a W1 the mask is arithmetic -- the gene begins behind the source
        const uint64_t behind = in_batch ? (lane >= 63 ? 0ull : (~0ull << (lane + 1))) : ~0ull;
        e.m[0] = r1 ? r0 : ((k[0] | k[3]) & behind); e.m[1] = r1;
    }
    else if (sk == 3) { e.m[0] = r0 & k[2]; e.m[1] = r0 & k[3]; }
    else {
        const uint64_t later = in_batch ? (lane >= 63 ? 0ull : (~0ull << (lane + 1))) : ~0ull;
        e.m[0] = r0 & k[0] & later; e.m[1] = r1; e.m[2] = r0 & k[1]; e.m[3] = r0 & k[2]; e.m[4] = r0 & k[3];
        e.m[5] = in_batch ? (r0 & k[0] & ~later) : 0ull;
    }
    return e;
}

BatchSched compile_batch(const Chain& C, const int b) {
    BatchSched S; S.miss = false;
    const int i0 = b << 6;
    DpwST T[64];
    uint64_t k[4] = {0, 0, 0, 0};
    for (int t = 0; t < 64; t++) {
        const int i = i0 + t;
        T[t] = i < C.n ? dpw_st(i, C.kf[i], C.ndx[i], C.stopv[i], C.lo[i]) : dpw_st(-1, 0, 0, 0, 0);
        if (T[t].i >= 0) k[T[t].kind] |= 1ull << t;
    }
    int jm = i0;
    for (int t = 0; t < 64; t++) if (T[t].i >= 0 && (T[t].kind == 0 || T[t].kind == 3)) jm = std::min(jm, std::max(C.q1[T[t].i], T[t].lo));
    if (jm < i0 - 64) { S.miss = true; return S; }          // near sources older than the batch before: the launch falls back (k_dpw_dyn)
    uint64_t r0, r1;
    for (int j = jm; j < i0; j++) {
        words_of(C, T, j, j - (i0 - 64), false, r0, r1);
        if (r0) S.near.push_back(entry_from(C, j, j - (i0 - 64), r0, r1, k, false));
    }
    for (int l = 0; l < 64 && i0 + l < C.n; l++) {
        words_of(C, T, i0 + l, l, true, r0, r1);
        if (r0) S.walk.push_back(entry_from(C, i0 + l, l, r0, r1, k, true));
    }
    return S;
}
inline unsigned lane_bits(const Entry& e, const int t) {
    // the bits of dpw_static_bits again, from the step's masks (m[0] / m[1] of a reverse stop are its reverse-start / reverse-stop lanes)
    const int sk = DPW_E_KIND(e.code);
    unsigned b = 0;
    if (sk == 3) { if ((e.m[0] >> t) & 1ull) b |= 1u; if ((e.m[1] >> t) & 1ull) b |= 2u; return b; }
    for (int q = 0; q < 6; q++) b |= (unsigned)((e.m[q] >> t) & 1ull) << q;
    return b;
}

}  // namespace

extern "C" int dpw_model_run(int n, const int32_t* ndx, const int32_t* stop_val, const uint8_t* type, const int8_t* strand,
                             const double* cscore, const double* sscore, const double* rscore, const double* uscore,
                             const int32_t* star_ptr, double st_wt, double* score, int32_t* traceb, int8_t* ov_mark,
                             int32_t* max_index, int64_t* stats /* [8] */) {
    double igm[64] = {0};
    for (int d = 0; d <= DPW_OPER_DIST; d++) igm[d] = (2.0 - ((double)d / DPW_OPER_DIST)) * 0.15 * st_wt;
    const DpwModel M{st_wt, -0.15 * st_wt, igm};
    for (int k = 0; k < 8; k++) stats[k] = 0;
    *max_index = -1;
    if (n <= 0) return 0;

    Chain C;
    C.n = n; C.ndx = ndx; C.stopv = stop_val;
    C.kf.resize(n); C.lo.resize(n); C.q1.resize(n); C.q2.resize(n); C.cs.resize(n);
    C.score.assign(n, 0.0); C.traceb.assign(n, -1); C.tbn.assign(n, -1); C.ov.assign(n, -1);
    C.sfxv.assign(n, NEG_INF); C.sfxi.assign(n, -1);
    // ---- topology pass, then the per-chain pass
    for (int i = 0; i < n; i++) {
        const DpwTopo t = dpw_topo_node(ndx, stop_val, type, strand, n, i);
        C.kf[i] = t.kf; C.lo[i] = t.lo; C.q1[i] = t.q1; C.q2[i] = t.q2;
    }
    C.ext.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        C.cs[i] = cscore[i] + sscore[i];
        const int kind = DPW_KIND(C.kf[i]);
        if (kind & 1) dpw_chain_ext(ndx, stop_val, strand, C.q2.data(), cscore, sscore, rscore, uscore, star_ptr, i, kind == 3, M, C.ext[i]);
    }

    const int nb = (n + 63) >> 6;
    std::vector<Lex> bm((size_t)nb, Lex{NEG_INF, -1});            // lexicographic maximum of `a` over every finished block
    Lex pp[64];                                                   // inclusive prefix maxima of `a` inside the previous block
    for (int t = 0; t < 64; t++) pp[t] = Lex{NEG_INF, -1};
    Lex R[3] = {{NEG_INF, -1}, {NEG_INF, -1}, {NEG_INF, -1}}; int Rn[3] = {-1, -1, -1};      // v[f] since the last forward stop of frame f
    int L3i[3] = {-1, -1, -1}, L3stop[3] = {0, 0, 0}, L3ndx[3] = {0, 0, 0}; double L3score[3] = {0, 0, 0};   // last reverse stop of each frame
    double end_best = -1.0; int end_idx = -1;

    for (int b = 0; b < nb; b++) {
        const int i0 = b << 6;
        DpwT T[64]; DpwLT LT[64]; DpwLane L[64]; int tbn_pre[64];
        for (int t = 0; t < 64; t++) { T[t] = load_target(C, i0 + t, M.negc); LT[t] = dpw_lean(T[t]); L[t] = DpwLane{0.0, -1}; tbn_pre[t] = -1; }
        // explicit lexicographic take of a candidate older than the batch
        auto take = [&](const int t, const bool ok, const double val, const int j, const int ov1, const int s_ndx) {
            const int cur = dpw_tag_index(L[t].tag);
            if (ok && (val > L[t].val || (val == L[t].val && j > cur))) { L[t].val = val; L[t].tag = j | (ov1 << DPW_TAG_BITS); tbn_pre[t] = s_ndx; }
        };
        auto apply = [&](const DpwS& S, const int t) {       // the one-piece pair logic, for the chains of candidates
            bool ok; double w; int mf;
            dpw_pair(S, T[t], M, ok, w, mf);
            take(t, ok, S.score + w, S.j, mf + 1, S.ndx);
        };

        // ---- (2) near steps first (ascending sources onto an empty state: ">=" is the whole rule): the sources from the
        //      earliest p_near of a gene begin up to the batch, one at a time
        static const bool legacy = getenv("DPW_MODEL_LEGACY") != nullptr;      // the steps as the kernel made them before the schedule
        BatchSched SC = legacy ? BatchSched() : compile_batch(C, b);
        const bool legacy_b = legacy || SC.miss;        // (a batch whose near sources reach past the batch before: as the kernel's fallback does it)
        if (SC.miss) stats[0]++;
        // one scheduled step: source values (score, traceb position, the extras of a forward stop) from the caller
        auto sched_step = [&](const Entry& e, const DpwS& S) {
            const int sk = DPW_E_KIND(e.code), sf = DPW_E_FRAME(e.code);
            for (int t = 0; t < 64; t++) {
                const unsigned bits = lane_bits(e, t) & 31u;
                if (!bits) continue;
                if (sk == 2) dpw_take_ge(L[t], (bits & 1u) != 0u, dpw_sval_r5(LT[t], bits, e.s_ndx, S.score, M), e.j);
                else if (sk == 3) dpw_take_ge(L[t], dpw_sok_r3(LT[t], bits, sf), S.score + dpw_w_r3(LT[t], sf), e.j);
                else dpw_sstep_f3(LT[t], L[t], bits, e.j, e.s_ndx, S.vm, S.tbn, S.score, S.x0, S.x1, S.x2, M);
            }
        };
        if (legacy_b) {
        int jmin = i0;
        for (int t = 0; t < 64; t++) if (T[t].i >= 0 && (T[t].kind == 0 || T[t].kind == 3)) jmin = std::min(jmin, std::max(T[t].q1, T[t].lo));
        for (int j = jmin; j < i0; j++) {
            const DpwS S = load_source(C, j);
            if ((S.kind == 1 || S.kind == 2) && S.tbn == -1) continue;
            if (S.kind == 0) continue;          // as the kernel: a forward start before the batch offers nothing that the frame carries of (3) do not hold
            stats[1]++;
            for (int t = 0; t < 64; t++) dpw_step(S, LT[t], L[t], M);
        }
        } else {
            stats[7] += (int64_t)SC.near.size() + (int64_t)SC.walk.size();
            for (const Entry& e : SC.near) {
                const DpwS S = load_source(C, e.j);
                if ((S.kind == 1 || S.kind == 2) && S.tbn == -1) continue;
                stats[1]++;
                sched_step(e, S);
            }
        }
        for (int t = 0; t < 64; t++) if (L[t].tag >= 0) tbn_pre[t] = ndx[dpw_tag_index(L[t].tag)];
        // ---- (1) gene begins: far gene ends, `a` over [lo, min(p_near, i0))
        for (int t = 0; t < 64; t++) {
            if (T[t].i < 0 || !(T[t].kind == 0 || T[t].kind == 3)) continue;
            const int lo = T[t].lo, hi = std::min(T[t].q1, i0);
            if (hi <= lo) continue;
            Lex r{NEG_INF, -1};
            const int rb = hi >> 6, part = hi & 63, Bl = lo >> 6;
            bool generic = false;
            int E = rb;                                    // whole blocks end here (exclusive)
            if (rb == b) { /* hi == i0 */ }
            else if (rb == b - 1) { if (Bl >= rb && part > 0 && (lo & 63) != 0) generic = true; }
            else generic = true;
            if (!generic) {
                if (rb == b - 1 && part > 0) {
                    if (Bl < rb || (lo & 63) == 0) lex_take(r, pp[part - 1].v, pp[part - 1].i);
                }
                int x = Bl;
                if ((lo & 63) != 0) {
                    if (Bl < E) lex_take(r, C.sfxv[lo], C.sfxi[lo]);
                    x = Bl + 1;
                }
                if (x < E) {
                    if (E - 1 - x >= 64) generic = true;             // beyond what the kernel keeps of the block maxima
                    else for (int q = x; q < E; q++) lex_take(r, bm[q].v, bm[q].i);
                }
            }
            if (generic) {
                stats[0]++;
                r = Lex{NEG_INF, -1};
                for (int j = lo; j < hi; j++) {
                    const int k = DPW_KIND(C.kf[j]);
                    if ((k == 1 || k == 2) && C.traceb[j] != -1) lex_take(r, C.score[j] + M.negc, j);
                }
            }
            if (r.i >= 0) take(t, true, r.v, r.i, 0, ndx[r.i]);
        }
        // ---- (3) forward stops: the best start / operon partner of their ORF met before the batch
        for (int f = 0; f < 3; f++) {
            for (int t = 0; t < 64; t++) {
                if (T[t].i < 0 || T[t].kind != 1 || T[t].frame != f) continue;
                if (R[f].i >= 0) take(t, true, R[f].v, R[f].i, 0, Rn[f]);
                break;                                       // the first forward stop of frame f in the batch only
            }
        }
        // ---- (4) reverse nodes: the last reverse stop of a frame before the batch (own stop of a reverse start; operon)
        for (int t = 0; t < 64; t++) {
            if (T[t].i < 0) continue;
            if (T[t].kind == 2) {
                const int f = T[t].frame;
                if (L3i[f] >= T[t].lo && L3i[f] >= 0 && L3stop[f] > T[t].ndx) take(t, true, L3score[f] + T[t].cs, L3i[f], 0, L3ndx[f]);
            } else if (T[t].kind == 3) {
                for (int f = 0; f < 3; f++)
                    if (((T[t].vm >> f) & 1) && L3i[f] >= T[t].lo && L3i[f] >= 0 && L3stop[f] > T[t].ndx)
                        take(t, true, L3score[f] + dpw_sel3(f, T[t].x0, T[t].x1, T[t].x2), L3i[f], 0, L3ndx[f]);
            }
        }
        // ---- (5) reverse nodes: forward stops that overlap the 3' end of the gene (chains of forward stops).  Round 6: a candidate
        //      is priced through the chain's OWN overlapping start only (interval form of dpw_lean): a pair that is admissible through
        //      another overlapping start q' lies on chain q' as well, and the plain connection (no overlapping start) is what the far
        //      gene ends / the near steps already hold.
        for (int t = 0; t < 64; t++) {
            if (T[t].i < 0) continue;
            if (T[t].kind == 2) {
                for (int j = T[t].q2; j < i0; j = C.q2[j]) {
                    if (ndx[j] >= T[t].stop_val + DPW_MAX_OPP_OVLP - 5) break;
                    stats[2]++;
                    const int s_ndx = ndx[j], tbj = C.tbn[j];
                    const bool ok = j >= T[t].lo && tbj != -1 && s_ndx > LT[t].dlo0 && s_ndx < LT[t].dhi0 && tbj + s_ndx + 7 < LT[t].drhs0;
                    take(t, ok, C.score[j] + T[t].csd, j, 0, s_ndx);
                }
            } else if (T[t].kind == 3) {
                for (int q = 0; q < 3; q++) {
                    if (!((T[t].vm >> q) & 1)) continue;
                    const int dlo = dpw_sel3i(q, LT[t].dlo0, LT[t].dlo1, LT[t].dlo2), dhi = dpw_sel3i(q, LT[t].dhi0, LT[t].dhi1, LT[t].dhi2),
                              drhs = dpw_sel3i(q, LT[t].drhs0, LT[t].drhs1, LT[t].drhs2);
                    if (dlo == INT_MAX) continue;                       // an overlapping start worth nothing (x <= 0) is never taken
                    for (int j = dpw_sel3i(q, T[t].cq0, T[t].cq1, T[t].cq2); j < i0; j = C.q2[j]) {
                        const int s_ndx = ndx[j], tbj = C.tbn[j];
                        if (s_ndx >= dlo + DPW_MAX_OPP_OVLP) break;     // dlo = n3s - 5: the chain ends at n3s + MAX_OPP_OVLP - 5
                        stats[3]++;
                        const bool ok = j >= T[t].lo && tbj != -1 && s_ndx > dlo && s_ndx < dhi && tbj + s_ndx + 7 < drhs;
                        take(t, ok, C.score[j] + dpw_sel3(q, T[t].x0, T[t].x1, T[t].x2), j, q + 1, s_ndx);
                    }
                }
            }
        }
        // ---- (6') instead of the walk, if asked for (DPW_MODEL_FIXPOINT=1): every lane recomputes its best in-batch source from the
        //      CURRENT values of the lanes before it, all lanes at once, until nothing changes.  Lane k is right after k + 1 rounds
        //      at the latest (its sources are), so the rounds end, and a state that reproduces itself is the walk's result; in
        //      practice a lane is right one round after its own best source is, i.e. after as many rounds as the longest run of
        //      tracebs inside the batch.  stats[4] = rounds, stats[5] = batches, stats[6] = most rounds of a batch.
        static const bool fixpoint = getenv("DPW_MODEL_FIXPOINT") != nullptr;
        if (fixpoint) {
            DpwLane L0[64];
            for (int t = 0; t < 64; t++) L0[t] = L[t];
            int rounds = 0;
            for (;;) {
                DpwLane Ln[64];
                for (int t = 0; t < 64; t++) Ln[t] = L0[t];
                for (int k = 0; k < 64 && i0 + k < n; k++) {
                    DpwS S; memset(&S, 0, sizeof S);
                    S.j = i0 + k; S.kind = T[k].kind; S.frame = T[k].frame; S.ndx = T[k].ndx; S.stop_val = T[k].stop_val; S.vm = T[k].vm;
                    const int tbk = dpw_tag_index(L[k].tag);
                    S.tbn = tbk < 0 ? -1 : (tbk >= i0 ? T[tbk - i0].ndx : tbn_pre[k]);
                    S.score = L[k].val; S.cs = T[k].cs; S.x0 = T[k].x0; S.x1 = T[k].x1; S.x2 = T[k].x2;
                    if ((S.kind == 1 || S.kind == 2) && tbk == -1) continue;
                    for (int t = k + 1; t < 64; t++) dpw_step(S, LT[t], Ln[t], M);
                }
                rounds++;
                bool same = true;
                for (int t = 0; t < 64; t++) if (Ln[t].val != L[t].val || Ln[t].tag != L[t].tag) same = false;
                for (int t = 0; t < 64; t++) L[t] = Ln[t];
                if (same) break;
            }
            stats[4] += rounds; stats[5]++; if (rounds > stats[6]) stats[6] = rounds;
            static const bool log_rounds = getenv("DPW_MODEL_ROUNDS") != nullptr;
            if (log_rounds) fprintf(stderr, "R %d\n", rounds);
        } else
        if (!legacy_b) {
            // ---- (6) the walk from the schedule: the batch's own entries in lane order
            for (const Entry& e : SC.walk) {
                const int k = e.lane, sk = DPW_E_KIND(e.code);
                if (sk == 1) {
                    for (int c = 0; c < k; c++) {
                        if (!((e.m[5] >> c) & 1ull)) continue;
                        const double v = L[c].val + T[c].cs;
                        const int bi = dpw_tag_index(L[k].tag);
                        if (v > L[k].val || (v == L[k].val && i0 + c > bi)) { L[k].val = v; L[k].tag = i0 + c; }
                    }
                }
                const int tbk = dpw_tag_index(L[k].tag);
                if ((sk == 1 || sk == 2) && tbk == -1) continue;
                DpwS S; memset(&S, 0, sizeof S);
                S.j = i0 + k; S.kind = sk; S.vm = T[k].vm;
                S.tbn = tbk < 0 ? -1 : (tbk >= i0 ? T[tbk - i0].ndx : tbn_pre[k]);
                S.score = L[k].val; S.x0 = T[k].x0; S.x1 = T[k].x1; S.x2 = T[k].x2;
                sched_step(e, S);
            }
        } else
        // ---- (6) the walk: lane k is final when the walk reaches source i0 + k
        for (int k = 0; k < 64 && i0 + k < n; k++) {
            DpwS S; memset(&S, 0, sizeof S);
            S.j = i0 + k; S.kind = T[k].kind; S.frame = T[k].frame; S.ndx = T[k].ndx; S.stop_val = T[k].stop_val; S.vm = T[k].vm;
            // as the kernel: forward starts take no step; a forward stop pulls the forward starts of its ORF that sit before it in
            // the batch when the walk reaches it (they are final by then), (value, index) deciding, ties to the larger index
            if (S.kind == 0) continue;
            if (S.kind == 1) {
                for (int c = 0; c < k; c++) {
                    if (!(T[c].kind == 0 && T[c].frame == T[k].frame && T[c].ndx > T[k].stop_val)) continue;
                    const double v = L[c].val + T[c].cs;
                    const int bi = dpw_tag_index(L[k].tag);
                    if (v > L[k].val || (v == L[k].val && i0 + c > bi)) { L[k].val = v; L[k].tag = i0 + c; }
                }
            }
            const int tbk = dpw_tag_index(L[k].tag);
            S.tbn = tbk < 0 ? -1 : (tbk >= i0 ? T[tbk - i0].ndx : tbn_pre[k]);
            S.score = L[k].val; S.cs = T[k].cs; S.x0 = T[k].x0; S.x1 = T[k].x1; S.x2 = T[k].x2;
            if ((S.kind == 1 || S.kind == 2) && tbk == -1) continue;
            bool any = false;
            for (int t = k + 1; t < 64; t++) { const DpwLane before = L[t]; dpw_step(S, LT[t], L[t], M); any = any || before.tag != L[t].tag || before.val != L[t].val; }
            static const bool log_steps = getenv("DPW_MODEL_STEPS") != nullptr;      // "S <source kind> <taken by any lane>" per walk step
            if (log_steps) fprintf(stderr, "S %d %d\n", S.kind, any ? 1 : 0);
        }
        // ---- (7) the batch is final: results, block structures, carries
        DpwBest B[64];
        DpwOut O[64];
        for (int t = 0; t < 64; t++) {
            const int tb = dpw_tag_index(L[t].tag);
            B[t] = DpwBest{L[t].val, tb, dpw_tag_ov(L[t].tag), tb < 0 ? -1 : (tb >= i0 ? T[tb - i0].ndx : tbn_pre[t])};
            O[t] = dpw_outputs(T[t], T[t].i >= 0 ? C.kf[T[t].i] : 0, B[t], M.negc);
            if (T[t].i < 0) continue;
            const int i = T[t].i;
            C.score[i] = B[t].val; C.traceb[i] = B[t].tb; C.ov[i] = (int8_t)B[t].ov; C.tbn[i] = B[t].tbn;
            if ((T[t].kind == 1 || T[t].kind == 2) && B[t].val >= end_best) { end_best = B[t].val; end_idx = i; }
        }
        Lex run{NEG_INF, -1};
        for (int t = 0; t < 64; t++) { lex_take(run, O[t].a, O[t].a > NEG_INF ? i0 + t : -1); pp[t] = run; }
        bm[b] = run;
        run = Lex{NEG_INF, -1};
        for (int t = 63; t >= 0; t--) {
            lex_take(run, O[t].a, O[t].a > NEG_INF ? i0 + t : -1);
            if (i0 + t < n) { C.sfxv[i0 + t] = run.v; C.sfxi[i0 + t] = run.i; }
        }
        for (int f = 0; f < 3; f++) {
            int u = -1;
            for (int t = 0; t < 64; t++) if (T[t].i >= 0 && T[t].kind == 1 && T[t].frame == f) u = t;
            if (u >= 0) { R[f] = Lex{NEG_INF, -1}; Rn[f] = -1; }
            for (int t = u + 1; t < 64; t++) {
                const double v = f == 0 ? O[t].v0 : (f == 1 ? O[t].v1 : O[t].v2);
                if (v > NEG_INF && (v > R[f].v || (v == R[f].v && i0 + t > R[f].i))) { R[f] = Lex{v, i0 + t}; Rn[f] = T[t].ndx; }
            }
            int w = -1;
            for (int t = 0; t < 64; t++) if (T[t].i >= 0 && T[t].kind == 3 && T[t].frame == f) w = t;
            if (w >= 0) { L3i[f] = i0 + w; L3stop[f] = T[w].stop_val; L3ndx[f] = T[w].ndx; L3score[f] = B[w].val; }
        }
    }
    for (int i = 0; i < n; i++) { score[i] = C.score[i]; traceb[i] = C.traceb[i]; ov_mark[i] = C.ov[i]; }
    *max_index = end_idx;
    return 0;
}
