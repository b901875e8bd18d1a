// Scalar (per node / per lane) logic of the wave-batch connection scorer (dp_wave.hip), written once for the device and
// for the host: tests/dpw_model.cpp runs the same decomposition with loops over the 64 lanes and is compared with the
// plain restatement of the reference's loop on the CPU (tests/test_dpw_model.py), so the arithmetic and the case analysis below are pinned before a kernel
// ever runs.  No intrinsics in this file.
//
// What is computed (ref: lib.pyx:1205-1237 `_score_connections`, _connection.h:94-408, impl/generic.h:29-36): for every
// node i in position order, over the sources j in its window [lo_i, i):
//     score[i] = max(0, max_j (score[j] + w(j, i))),  ties -> largest j,  traceb[i] = that j,  ov_mark[i] as the pair leaves it.
// Node kinds: 0 = F5 forward start, 1 = F3 forward stop, 2 = R5 reverse start, 3 = R3 reverse stop.  In walk order a forward
// gene is F5 -> F3 and a reverse gene R3 -> R5: F5 / R3 are gene BEGINS, F3 / R5 gene ENDS.  A gene end that was never
// reached (traceb == -1) is not a source (ref: _connection.h:110-114).
#pragma once

#include <limits.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define DPW_HD __host__ __device__ __forceinline__
#else
#define DPW_HD inline
#endif

#define DPW_MAX_NODE_DIST 500   // ref: _connection.h:5
#define DPW_MAX_OPP_OVLP  200
#define DPW_OPER_DIST     60
#define DPW_NONE          0x7fffffff

// topology byte: kind | frame << 2 | (F3 only) bit 4 + f: this node lies in the ORF of the next forward stop of frame f
#define DPW_KIND(kf)    ((kf) & 3)
#define DPW_FRAME(kf)   (((kf) >> 2) & 3)
#define DPW_INORF(kf,f) (((kf) >> (4 + (f))) & 1)

// Per-chain extras of a stop node: a 64-byte record at the node's own index (only the records of stop nodes are ever touched).
struct alignas(64) DpwExt {
    double  x[3];      // F3 source: cs(n3_k) + igm(j, n3_k);  R3 target: cs(n3_k) + igm(n3_k, i)   (n3_k = nodes[star_ptr[k]])
    // R3, round 6 -- the interval of forward-stop positions that can reach the node through overlapping start k, made where the extras are
    // built (k_ovl_stops: a kernel that waits for memory) instead of by every batch of the connection scorer (dpw_lean):
    //   dlo[k] < s_ndx < dhi[k]   with dlo = n3s - 5, dhi = min(n3s + MAX_OPP_OVLP - 5, (n3n + n3s - 6) >> 1, ndx - 4)   (n3n / n3s: position
    //   and stop_val of overlapping start k); empty (INT_MAX, INT_MIN) where the start is worth nothing (x[k] <= 0: never taken) or absent.
    int32_t dlo[3];
    int32_t dhi[3];
    int32_t cq[3];     // R3: first forward stop that can overlap the 3' end of the gene of overlapping start k, or DPW_NONE
    int32_t vm;        // bit k: star_ptr[k] != -1
};

// Per-model constants (same layout as ModelConst in pga_internal.h).
struct DpwModel { double st_wt, negc; const double* igm; };

DPW_HD double dpw_sel3(int k, double a, double b, double c) { return k == 0 ? a : (k == 1 ? b : c); }
DPW_HD int dpw_sel3i(int k, int a, int b, int c) { return k == 0 ? a : (k == 1 ? b : c); }

// _intergenic_mod_same for two same-strand nodes `d` apart that neither overlap nor touch (ref: _connection.h:52-78)
DPW_HD double dpw_igm_apart(int d, double negc, const double* igm) {
    double r = 0.0;
    if (d > 3 * DPW_OPER_DIST) r = negc;
    else if (d <= DPW_OPER_DIST && d >= 0) r = igm[d];
    return r;
}

// General _intergenic_mod_same (ref: _connection.h:52-78); a = n1, b = n2, both on strand a_strand.
DPW_HD double dpw_igm_same(int a_ndx, int a_strand, double a_r, double a_u, int b_ndx, double b_r, double b_u, double st_wt, const double* igm) {
    const int dist = a_ndx > b_ndx ? a_ndx - b_ndx : b_ndx - a_ndx;
    const bool ovl = a_ndx + 2 * a_strand >= b_ndx;
    double r = 0.0;
    if (a_ndx + 2 == b_ndx || a_ndx == b_ndx + 1) {
        if (a_strand == 1) { if (b_r < 0) r -= b_r; if (b_u < 0) r -= b_u; }
        else               { if (a_r < 0) r -= a_r; if (a_u < 0) r -= a_u; }
    }
    if (dist > 3 * DPW_OPER_DIST) r -= 0.15 * st_wt;
    else if ((dist <= DPW_OPER_DIST && !ovl) || dist * 4 < DPW_OPER_DIST) r += igm[dist];
    return r;
}

// ------------------------------------------------------------------------------------------------------------------------
// Topology pass: what depends on positions and kinds only (shared by every model scored on the contig).
//   lo   window start (ref: lib.pyx:1221-1233): 500 nodes back, stretched to the far end of a giant ORF, then another 500
//   q1   F5 / R3: p_near = first index whose position is within 3 * OPER_DIST bases (closer gene ends need the exact
//        intergenic term)
//   q2   F3: the next forward stop;  R5: the first forward stop at or after stop_val - 4, i.e. the first one that can overlap
//        the 3' end of this gene (DPW_NONE: none before stop_val + MAX_OPP_OVLP)
struct DpwTopo { uint8_t kf; int32_t lo, q1, q2; };

// `f3` (device, optional): a forward stop's q2 and operon bits when the caller has them already (k_dpw_topo finds the next forward stop
// of every frame from bit masks of its workgroup's nodes instead of walking the nodes eight at a time)
struct DpwF3Hint { int q2, bits; };
DPW_HD DpwTopo dpw_topo_node(const int32_t* ndx, const int32_t* stopv, const uint8_t* type, const int8_t* strand, const int n, const int i,
                             const DpwF3Hint* f3 = nullptr) {
    DpwTopo t;
    const int my_ndx = ndx[i], my_stop = stopv[i];
    const bool rev = strand[i] != 1, stop = type[i] == 3;
    const int kind = (rev ? 2 : 0) | (stop ? 1 : 0);
    int kf = kind | ((my_ndx % 3) << 2);
    int lo = i < DPW_MAX_NODE_DIST ? 0 : i - DPW_MAX_NODE_DIST;
    if ((kind == 2 || kind == 1) && ndx[lo] > my_stop) {
        // the reference walks down to the highest index whose position equals stop_val, or to 0
        // (a position holds at most two nodes: everything before node lo - 2 d - 2 lies more than d positions left of node lo)
        int a = lo - 2 * (ndx[lo] - my_stop) - 2, b = lo;
        if (a < 0) a = 0;
        while (a < b) { const int m = (a + b) >> 1; if (ndx[m] <= my_stop) a = m + 1; else b = m; }
        lo = (a > 0 && ndx[a - 1] == my_stop) ? a - 1 : 0;
    }
    lo = lo < DPW_MAX_NODE_DIST ? 0 : lo - DPW_MAX_NODE_DIST;
    t.lo = lo; t.q1 = 0; t.q2 = 0;
    if (kind == 0 || kind == 3) {
        // first index in [0, i) with ndx >= my_ndx - 3 * OPER_DIST (positions ascend): a handful of nodes back, so look at the
        // eight before, then the eight before those ...: each round is eight independent loads, a binary search ten dependent ones
        const int v = my_ndx - 3 * DPW_OPER_DIST;
        int a = i;                              // every index in [a, i) has ndx >= v
        while (a > 0) {
            int x[8];
            for (int k = 0; k < 8; k++) x[k] = ndx[a - 1 - k >= 0 ? a - 1 - k : 0];
            int k = 0;
            while (k < 8 && a - 1 - k >= 0 && x[k] >= v) k++;
            a -= k;
            if (k < 8) break;
        }
        t.q1 = a > lo ? a : lo;
    } else if (kind == 1 && f3 != nullptr) {
        t.q2 = f3->q2; kf |= f3->bits;
    } else if (kind == 1) {
        t.q2 = n;
        int seen = 0;
        // the nodes after i, eight at a time (their loads do not wait for one another), until a forward stop of every frame was met
        for (int j0 = i + 1; j0 < n && seen != 7; j0 += 8) {
            int xs[8], xt[8], xn[8], xv[8];
            for (int k = 0; k < 8; k++) {
                const int j = j0 + k < n ? j0 + k : n - 1;
                xs[k] = strand[j]; xt[k] = type[j]; xn[k] = ndx[j]; xv[k] = stopv[j];
            }
            for (int k = 0; k < 8 && j0 + k < n && seen != 7; k++) {
                if (xs[k] != 1 || xt[k] != 3) continue;
                if (t.q2 == n) t.q2 = j0 + k;
                const int f = xn[k] % 3;
                if (seen & (1 << f)) continue;
                seen |= 1 << f;
                if (xv[k] < my_ndx) kf |= 1 << (4 + f);     // inside that stop's ORF: an operon candidate for it
            }
        }
    } else {
        const int v = my_stop - 4;
        int a = i - 2 * (my_ndx > v ? my_ndx - v : 0) - 2, b = i;                       // first index in [0, i) with ndx >= my_stop - 4
        if (a < 0) a = 0;
        while (a < b) { const int m = (a + b) >> 1; if (ndx[m] < v) a = m + 1; else b = m; }
        t.q2 = DPW_NONE;
        for (int j = a; j < i && ndx[j] < my_stop + DPW_MAX_OPP_OVLP - 5; j++)
            if (strand[j] == 1 && type[j] == 3) { t.q2 = j; break; }
    }
    t.kf = (uint8_t)kf;
    return t;
}

// ------------------------------------------------------------------------------------------------------------------------
// Per-chain pass: cs = cscore + sscore of every node; the extras of a stop node (third-node terms folded in,
// ref: _connection.h:166-176, 296-325, 345-356).
DPW_HD void dpw_chain_ext_sp(const int32_t* ndx, const int32_t* stopv, const int8_t* strand, const int32_t* topo_q2, const double* cscore,
                             const double* sscore, const double* rscore, const double* uscore, const int (&sp)[3], const int i,
                             const bool rev, const DpwModel& M, DpwExt& e, const double* css = nullptr);
DPW_HD void dpw_chain_ext(const int32_t* ndx, const int32_t* stopv, const int8_t* strand, const int32_t* topo_q2, const double* cscore,
                          const double* sscore, const double* rscore, const double* uscore, const int32_t* star_ptr /* [n][3] */, const int i,
                          const bool rev, const DpwModel& M, DpwExt& e) {
    const int sp[3] = {star_ptr[3 * i], star_ptr[3 * i + 1], star_ptr[3 * i + 2]};
    dpw_chain_ext_sp(ndx, stopv, strand, topo_q2, cscore, sscore, rscore, uscore, sp, i, rev, M, e);
}
// the same with the three overlapping starts of node i handed over
DPW_HD void dpw_chain_ext_sp(const int32_t* ndx, const int32_t* stopv, const int8_t* strand, const int32_t* topo_q2, const double* cscore,
                             const double* sscore, const double* rscore, const double* uscore, const int (&sp)[3], const int i,
                             const bool rev, const DpwModel& M, DpwExt& e, const double* css /* or nullptr: cscore + sscore, already added */) {
    e.vm = 0;
    const int my_ndx = ndx[i];
    for (int k = 0; k < 3; k++) {
        e.x[k] = 0.0; e.dlo[k] = INT_MAX; e.dhi[k] = INT_MIN; e.cq[k] = DPW_NONE;
        const int p = sp[k];
        if (p < 0) continue;
        e.vm |= 1 << k;
        const double cs3 = css != nullptr ? css[p] : cscore[p] + sscore[p];
        double ig;
        // (of the RBS / upstream scores only the start's enter, and only when the two nodes are adjacent: _connection.h:60-66)
        const int pn = ndx[p];
        const bool adj = !rev ? (my_ndx + 2 == pn || my_ndx == pn + 1) : (pn + 2 == my_ndx || pn == my_ndx + 1);
        const double pr = adj ? rscore[p] : 0.0, pu = adj ? uscore[p] : 0.0;
        if (!rev)   // F3 source j = i, n3 = forward start: igm(j, n3)       (ref: _connection.h:170-174)
            ig = (strand[p] == 1) ? dpw_igm_same(my_ndx, 1, 0.0, 0.0, pn, pr, pu, M.st_wt, M.igm) : M.negc;
        else        // R3 target i, n3 = reverse start: igm(n3, i)          (ref: _connection.h:313-320, 353-355)
            ig = (strand[p] == -1) ? dpw_igm_same(pn, -1, pr, pu, my_ndx, 0.0, 0.0, M.st_wt, M.igm) : M.negc;
        e.x[k] = cs3 + ig;
        if (rev && e.x[k] > 0.0) {
            const int n3n = ndx[p], n3s = stopv[p];
            int hi = n3s + DPW_MAX_OPP_OVLP - 5;
            const int h2 = (n3n + n3s - 6) >> 1;
            if (h2 < hi) hi = h2;
            if (my_ndx - 4 < hi) hi = my_ndx - 4;
            e.dlo[k] = n3s - 5; e.dhi[k] = hi;
        }
        // the overlapping start of a reverse stop is a reverse start: its own first candidate is this pair's (same stop_val)
        if (rev && strand[p] == -1) e.cq[k] = topo_q2[p];
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Pair logic.  T: a target lane; S: one source, the same for every lane of the wave.
struct DpwT {
    int i;                      // chain index, -1 = no node in this lane
    int kind, frame, ndx, stop_val, lo, q1, q2, vm;
    double cs, csd;             // csd = cs + negc
    double x0, x1, x2;          // stops
    int dlo0, dlo1, dlo2, dhi0, dhi1, dhi2, cq0, cq1, cq2;      // R3 (DpwExt)
};
struct DpwS {
    int j, kind, frame, ndx, stop_val, vm;
    int tbn;                    // position of the source's own traceb node, -1 = it has none
    double score, cs, x0, x1, x2;
};
// running result of a target lane
struct DpwBest { double val; int tb, ov, tbn; };

// the connection source S -> target T alone: allowed?, its weight, the ov_mark it leaves (ref: _connection.h:94-367 with
// the skip conditions of impl/generic.h:29-36 folded in)
DPW_HD void dpw_pair(const DpwS& S, const DpwT& T, const DpwModel& M, bool& ok, double& w, int& mf) {
    mf = -1; w = 0.0;
    ok = (S.j >= T.lo) & (S.j < T.i);
    if ((S.kind == 1 || S.kind == 2) && S.tbn == -1) { ok = false; return; }
    if (S.kind == 0) {
        // 5'fwd -> 3'fwd: a gene (ref: :166-174; skip condition 5: same frame only)
        ok = ok & (T.kind == 1) & (T.frame == S.frame) & (T.stop_val < S.ndx);
        w = S.cs;
    } else if (S.kind == 2) {
        // 5'rev -> 5'fwd (ref: :125-130) and 5'rev -> 3'rev (ref: :337-342)
        const bool a = (T.kind == 0) & (S.ndx < T.ndx);
        const bool b = (T.kind == 3) & (S.ndx < T.ndx - 2);
        ok = ok & (a | b);
        w = b ? dpw_igm_apart(T.ndx - S.ndx, M.negc, M.igm) : M.negc;
    } else if (S.kind == 3) {
        // 3'rev -> 5'rev: a gene (ref: :228-235; skip condition 6) and 3'rev -> 3'rev operon (ref: :345-356)
        const bool a = (T.kind == 2) & (T.frame == S.frame) & (S.stop_val > T.ndx);
        const bool b = (T.kind == 3) & (S.stop_val > T.ndx) & (((T.vm >> S.frame) & 1) != 0);
        ok = ok & (a | b);
        w = a ? T.cs : dpw_sel3(S.frame, T.x0, T.x1, T.x2);
    } else {
        if (T.kind == 0) {            // 3'fwd -> 5'fwd intergenic (ref: :117-124)
            ok = ok & (S.ndx + 2 < T.ndx);
            w = dpw_igm_apart(T.ndx - S.ndx, M.negc, M.igm);
        } else if (T.kind == 1) {     // 3'fwd -> 3'fwd operon through j's overlapping start (ref: :177-188)
            ok = ok & (T.stop_val < S.ndx) & (((S.vm >> T.frame) & 1) != 0);
            w = dpw_sel3(T.frame, S.x0, S.x1, S.x2);
        } else if (T.kind == 2) {     // 3'fwd -> 5'rev overlapping opposite 3' ends (ref: :238-254)
            const int ovlp = (S.ndx + 2) - (T.stop_val - 2) + 1;
            ok = ok & !(T.stop_val - 2 >= S.ndx + 2) & (ovlp < DPW_MAX_OPP_OVLP)
                    & ((S.ndx - T.stop_val) < (T.ndx - S.ndx + 3))
                    & ((S.ndx - T.stop_val) < (T.stop_val - 3 - S.tbn));
            w = T.csd;
        } else {                      // 3'fwd -> 3'rev, possibly through one of i's overlapping starts (ref: :288-336)
            ok = ok & (S.ndx < T.ndx - 4);
            double maxval = 0.0;
            for (int q = 0; q < 3; q++) {
                const int dlo = dpw_sel3i(q, T.dlo0, T.dlo1, T.dlo2), dhi = dpw_sel3i(q, T.dhi0, T.dhi1, T.dhi2);
                const double cur = dpw_sel3(q, T.x0, T.x1, T.x2);
                // (the interval form of the reference's four overlap tests, see DpwLT; an empty interval: no such start, or one worth nothing)
                const bool tk = (dlo != INT_MAX) & (S.ndx > dlo) & (S.ndx < dhi) & (S.tbn + S.ndx + 7 < 2 * (dlo + 5)) & (cur > maxval);
                if (tk) { mf = q; maxval = cur; }
            }
            w = mf != -1 ? maxval : M.negc;
        }
    }
}

// "val >= best" of the reference's ascending scan (ref: _connection.h:135-139) as a lexicographic (value, index) maximum:
// candidates may then come in any order, and partial maxima over disjoint source sets merge exactly.
DPW_HD void dpw_take(DpwBest& b, bool ok, double val, int j, int mf, int s_ndx) {
    if (ok && (val > b.val || (val == b.val && j > b.tb))) { b.val = val; b.tb = j; b.ov = mf; b.tbn = s_ndx; }
}
DPW_HD void dpw_apply(const DpwS& S, const DpwT& T, const DpwModel& M, DpwBest& B) {
    bool ok; double w; int mf;
    dpw_pair(S, T, M, ok, w, mf);
    dpw_take(B, ok, S.score + w, S.j, mf, S.ndx);
}

// Candidate values a finished node leaves for later targets: `a` for far gene begins (score + the constant intergenic
// term; -inf when the node is no gene end or was never reached), v[f] towards the forward stop of frame f whose ORF holds
// it (a forward start: score + cs; a reached forward stop with an overlapping start in that frame: score + x[f]).
struct DpwOut { double a, v0, v1, v2; };
DPW_HD DpwOut dpw_outputs(const DpwT& T, const int kf, const DpwBest& B, const double negc) {
    const double NEG_INF = -__builtin_huge_val();
    DpwOut o{NEG_INF, NEG_INF, NEG_INF, NEG_INF};
    if (T.i < 0) return o;
    const bool alive = B.tb != -1;
    if (T.kind == 0) {
        const double g = B.val + T.cs;
        if (T.frame == 0) o.v0 = g; else if (T.frame == 1) o.v1 = g; else o.v2 = g;
    } else if (T.kind == 1 && alive) {
        o.a = B.val + negc;
        if (((T.vm >> 0) & 1) && DPW_INORF(kf, 0)) o.v0 = B.val + T.x0;
        if (((T.vm >> 1) & 1) && DPW_INORF(kf, 1)) o.v1 = B.val + T.x1;
        if (((T.vm >> 2) & 1) && DPW_INORF(kf, 2)) o.v2 = B.val + T.x2;
    } else if (T.kind == 2 && alive) {
        o.a = B.val + negc;
    }
    return o;
}

// ------------------------------------------------------------------------------------------------------------------------
// Lean steps: the pair logic again, cut by source kind so that a step only touches what its kind needs.  A step applies
// ONE source (uniform over the wave: its fields come in as scalars) to the lane's target; sources arrive in ascending
// index order, which makes the reference's ">=" the whole tie rule.  The lane's running state is its value and a tag:
// source index | (ov_mark + 1) << 28, -1 = nothing taken yet.  dpw_pair above stays the one-piece statement of the same
// rules (the host model checks the steps against the reference's plain loop, the chains of candidates use dpw_pair directly).
#define DPW_TAG_BITS 28
#define DPW_TAG_MASK ((1 << DPW_TAG_BITS) - 1)

struct DpwLane { double val; int tag; };
DPW_HD int dpw_tag_index(int tag) { return tag < 0 ? -1 : (tag & DPW_TAG_MASK); }
DPW_HD int dpw_tag_ov(int tag) { return tag < 0 ? -1 : (tag >> DPW_TAG_BITS) - 1; }

// What a lane keeps of its target for the steps.  A forward-stop source reaches a reverse target through candidate q
// (the one candidate of a reverse start; one per overlapping start of a reverse stop) only when
//     dlo[q] < s_ndx < dhi[q]   and   tbn + s_ndx + 7 < drhs[q]          (tbn: position of the source's own traceb node)
// which restates, as interval tests on the source position (ref: _connection.h:238-254, 296-325):
//   reverse stop, candidate q, ovlp = s_ndx + 5 - n3s:   ovlp > 0;  ovlp < MAX_OPP_OVLP;  ovlp < n3n - (s_ndx + 2);
//                                                        ovlp < n3s - tbn - 2;  only a candidate worth more than 0 is taken
//   reverse start:  stop_val - 2 < s_ndx + 2;  s_ndx - stop_val + 5 < MAX_OPP_OVLP;  2 s_ndx < ndx + stop_val + 3;
//                   s_ndx - stop_val < stop_val - 3 - tbn
struct DpwLT {
    int i, kind, frame, ndx, stop_val, lo, vm;
    double cs, csd, x0, x1, x2;
    int okhi;                           // reverse stop: a forward stop connects at all only when s_ndx < ndx - 4
    int dlo0, dhi0, drhs0, dlo1, dhi1, drhs1, dlo2, dhi2, drhs2;
};
DPW_HD DpwLT dpw_lean(const DpwT& T) {
    DpwLT L;
    L.i = T.i; L.kind = T.kind; L.frame = T.frame; L.ndx = T.ndx; L.stop_val = T.stop_val; L.lo = T.lo; L.vm = T.vm;
    L.cs = T.cs; L.csd = T.csd; L.x0 = T.x0; L.x1 = T.x1; L.x2 = T.x2;
    L.okhi = INT_MIN;
    L.dlo0 = L.dlo1 = L.dlo2 = INT_MAX; L.dhi0 = L.dhi1 = L.dhi2 = INT_MIN; L.drhs0 = L.drhs1 = L.drhs2 = INT_MIN;
    if (T.kind == 3) {
        L.okhi = T.ndx - 4;
        // (the intervals come with the extras: DpwExt)
        L.dlo0 = T.dlo0; L.dlo1 = T.dlo1; L.dlo2 = T.dlo2; L.dhi0 = T.dhi0; L.dhi1 = T.dhi1; L.dhi2 = T.dhi2;
        L.drhs0 = T.dlo0 != INT_MAX ? 2 * T.dlo0 + 10 : INT_MIN; L.drhs1 = T.dlo1 != INT_MAX ? 2 * T.dlo1 + 10 : INT_MIN;
        L.drhs2 = T.dlo2 != INT_MAX ? 2 * T.dlo2 + 10 : INT_MIN;
    } else if (T.kind == 2) {
        L.dlo0 = T.stop_val - 4;
        L.dhi0 = T.stop_val + DPW_MAX_OPP_OVLP - 5;
        const int h2 = (T.ndx + T.stop_val + 4) >> 1;
        if (h2 < L.dhi0) L.dhi0 = h2;
        L.drhs0 = 2 * T.stop_val + 4;
    }
    return L;
}

DPW_HD void dpw_take_ge(DpwLane& L, const bool ok, const double val, const int tag) {
    if (ok && val >= L.val) { L.val = val; L.tag = tag; }
}
// source j is inside the lane's window and before the lane's node
DPW_HD bool dpw_inwin(const DpwLT& T, const int j) { return (j >= T.lo) & (j < T.i); }

// forward start j (frame sf, position s_ndx): the forward stop of its ORF takes score + cs          (ref: :166-174)
DPW_HD bool dpw_ok_f5(const DpwLT& T, const int j, const int sf, const int s_ndx) {
    return dpw_inwin(T, j) & (T.kind == 1) & (T.frame == sf) & (T.stop_val < s_ndx);
}
// reverse start j (a gene end): every later gene begin; the intergenic term depends on the distance only towards reverse
// stops, and there only within 3 * OPER_DIST bases                                                    (ref: :125-130, 337-342)
DPW_HD bool dpw_ok_r5(const DpwLT& T, const int j, const int s_ndx) {
    const bool a = (T.kind == 0) & (s_ndx < T.ndx), b = (T.kind == 3) & (s_ndx < T.ndx - 2);
    return dpw_inwin(T, j) & (a | b);
}
DPW_HD bool dpw_r5_needs_table(const DpwLT& T, const int s_ndx) { return (T.kind == 3) & (T.ndx - s_ndx <= 3 * DPW_OPER_DIST); }
DPW_HD double dpw_w_r5(const DpwLT& T, const int s_ndx, const DpwModel& M) {
    return T.kind == 3 ? dpw_igm_apart(T.ndx - s_ndx, M.negc, M.igm) : M.negc;
}
// reverse stop j (frame sf, far end of its ORF at s_stop): the reverse starts of its ORF take score + their cs; a reverse
// stop inside its ORF with an overlapping start in frame sf takes score + x[sf]                      (ref: :228-235, 345-356)
DPW_HD bool dpw_ok_r3(const DpwLT& T, const int j, const int sf, const int s_stop) {
    const bool a = (T.kind == 2) & (T.frame == sf), b = (T.kind == 3) & (((T.vm >> sf) & 1) != 0);
    return dpw_inwin(T, j) & (s_stop > T.ndx) & (a | b);
}
DPW_HD double dpw_w_r3(const DpwLT& T, const int sf) { return T.kind == 2 ? T.cs : dpw_sel3(sf, T.x0, T.x1, T.x2); }
// forward stop j (a gene end; s_vm / s_x*: its overlapping starts; s_tbn: position of its own traceb node): all four kinds
DPW_HD void dpw_step_f3(const DpwLT& T, DpwLane& L, const int j, const int s_ndx, const int s_vm, const int s_tbn, const double s_score,
                        const double s_x0, const double s_x1, const double s_x2, const DpwModel& M) {
    bool ok = dpw_inwin(T, j);
    double w; int ov1 = 0;
    if (T.kind == 0) {                        // intergenic step (ref: :117-124)
        ok = ok & (s_ndx + 2 < T.ndx);
        w = dpw_igm_apart(T.ndx - s_ndx, M.negc, M.igm);
    } else if (T.kind == 1) {                 // operon through j's overlapping start of the target's frame (ref: :177-188)
        ok = ok & (T.stop_val < s_ndx) & (((s_vm >> T.frame) & 1) != 0);
        w = dpw_sel3(T.frame, s_x0, s_x1, s_x2);
    } else {
        const int lhs = s_tbn + s_ndx + 7;
        const bool c0 = (s_ndx > T.dlo0) & (s_ndx < T.dhi0) & (lhs < T.drhs0);
        if (T.kind == 2) {                    // overlapping opposite 3' ends (ref: :238-254)
            ok = ok & c0;
            w = T.csd;
        } else {                              // towards a reverse stop, through the best admissible overlapping start (ref: :288-336)
            ok = ok & (s_ndx < T.okhi);
            const bool c1 = (s_ndx > T.dlo1) & (s_ndx < T.dhi1) & (lhs < T.drhs1);
            const bool c2 = (s_ndx > T.dlo2) & (s_ndx < T.dhi2) & (lhs < T.drhs2);
            double mv = 0.0; int m = -1;
            if (c0 & (T.x0 > mv)) { mv = T.x0; m = 0; }
            if (c1 & (T.x1 > mv)) { mv = T.x1; m = 1; }
            if (c2 & (T.x2 > mv)) { mv = T.x2; m = 2; }
            w = m >= 0 ? mv : M.negc;
            ov1 = m + 1;
        }
    }
    dpw_take_ge(L, ok, s_score + w, j | (ov1 << DPW_TAG_BITS));
}
// one whole step from a source record (what the host model runs; the kernel calls the pieces above around its wave votes)
DPW_HD void dpw_step(const DpwS& S, const DpwLT& T, DpwLane& L, const DpwModel& M) {
    if (S.kind == 0) dpw_take_ge(L, dpw_ok_f5(T, S.j, S.frame, S.ndx), S.score + S.cs, S.j | (0 << DPW_TAG_BITS));
    else if (S.kind == 2) dpw_take_ge(L, dpw_ok_r5(T, S.j, S.ndx), S.score + dpw_w_r5(T, S.ndx, M), S.j);
    else if (S.kind == 3) dpw_take_ge(L, dpw_ok_r3(T, S.j, S.frame, S.stop_val), S.score + dpw_w_r3(T, S.frame), S.j);
    else dpw_step_f3(T, L, S.j, S.ndx, S.vm, S.tbn, S.score, S.x0, S.x1, S.x2, M);
}

// ------------------------------------------------------------------------------------------------------------------------
// Step schedule.  Which sources take a pair step onto a batch of 64 targets, in which order, and which lanes each of them can
// reach at all is TOPOLOGY: kinds, frames, positions, stop positions, windows -- the same for every model scored on the contig
// under one translation table (ref: the six skip conditions of impl/generic.h:29-36 and the static tests of _connection.h:94-367
// read nothing else).  It is therefore compiled once per (contig, table) into lane masks (k_dpw_sched; the host loop of
// tests/dpw_model.cpp builds the same words): per step what is left is the source's value, one add, one compare against the lanes'
// running values under the step's lane mask -- plus, for the few relations that depend on a model (the overlapping starts of a stop
// node: star_ptr), the dynamic tests listed with each kind.
//
// Round 6 -- the schedule is per NODE, not per batch: node j (lane l = j & 63 of its batch) carries four 64-bit words,
//   W0   the lanes of its OWN batch it reaches as a source, by any relation (the relations are disjoint by the target's kind, so the
//        step takes the word apart with the lane masks of the batch's kinds); for a forward stop also the forward starts it PULLS,
//        which are forward-start lanes BEFORE it (the ones it reaches lie behind it)
//   W1   the lanes among them whose intergenic term depends on the distance (R5 source: reverse stops within 3 * OPER_DIST bases;
//        F3 source: forward starts within 3 * OPER_DIST bases)
//   N0, N1  the same towards the batch BEHIND its own (the near steps of that batch: sources from the earliest p_near of one of its
//        gene begins on, `jm`; a batch whose jm lies before the batch in front of it is marked DPW_SCHED_NONE and the launch repeated
//        by k_dpw_dyn -- more than 64 nodes within 3 * OPER_DIST bases, never on sequence)
// and a chain's wavefront loads them with two coalesced 16-byte reads per lane -- the near sources of a batch sit in the lanes the
// wave held them in one batch earlier, so their values stay in registers.  (Rounds 5's form was a list of 32-byte slots per batch,
// read line by line through the scalar cache: ten dependent scalar-load round trips per batch, 12 200 of the walk's 20 000 cycles.)
//   source kind      reaches (W0)                                                       dynamic part of the step
//   R5               gene begins in the window with s_ndx < key_r5 (ref: :125-130, 337-342)   --
//   R3               reverse starts of its frame inside its ORF (:228-235);                    --
//                    reverse stops inside its ORF (:345-356)                                   the lane has an overlapping start in its frame
//   F3               forward starts behind it (:117-124)                                       --
//                    forward stops whose ORF holds it (:177-188)                               the SOURCE has an overlapping start in the lane's frame
//                    reverse starts whose static interval holds s_ndx (:238-254)               tbn + s_ndx + 7 < drhs0
//                    reverse stops with s_ndx < ndx - 4 (:288-336)                             the candidates through the lane's overlapping starts
//                    (batch sources only) the forward starts of its ORF before it (:166-174)   pulled: (value, index) maximum
// Forward starts never step (a forward stop pulls the starts of its ORF).
struct DpwSchedHdr { uint32_t off; uint32_t cnt; int32_t jm; int32_t used; };    // per batch: off = 0, or DPW_SCHED_NONE (the batch's near sources did not fit); jm
struct DpwSlot { uint64_t w0, w1, n0, n1; };                                      // per node
#define DPW_SCHED_STRIDE 64u                                                      // records a batch owns (one per lane)
#define DPW_SCHED_NONE 0xffffffffu
#define DPW_E_CODE(kind, frame) ((uint32_t)((kind) | ((frame) << 2) | ((kind) != 2 ? 16 : 0)))      // (host model: kind / frame of an entry)
#define DPW_E_KIND(c)  ((int)((c) & 3))
#define DPW_E_FRAME(c) ((int)(((c) >> 2) & 3))

// what the topology fixes of a target
struct DpwST { int i, kind, frame, ndx, stop_val, lo, dlo0, dhi0; };
DPW_HD DpwST dpw_st(const int i /* -1: no node */, const int kf, const int ndx, const int stop_val, const int lo) {
    DpwST T;
    T.i = i; T.kind = i >= 0 ? DPW_KIND(kf) : -1; T.frame = DPW_FRAME(kf); T.ndx = ndx; T.stop_val = stop_val; T.lo = i >= 0 ? lo : INT_MAX;
    T.dlo0 = INT_MAX; T.dhi0 = INT_MIN;
    if (T.kind == 2) {          // as dpw_lean
        T.dlo0 = stop_val - 4;
        T.dhi0 = stop_val + DPW_MAX_OPP_OVLP - 5;
        const int h2 = (ndx + stop_val + 4) >> 1;
        if (h2 < T.dhi0) T.dhi0 = h2;
    }
    return T;
}
// the static bits of source (j, kind sk, frame sf, position s_ndx, stop_val s_stop) towards target T, in the order of the entry's masks
DPW_HD unsigned dpw_static_bits(const DpwST& T, const int j, const int sk, const int sf, const int s_ndx, const int s_stop) {
    if (!((j >= T.lo) & (j < T.i))) return 0u;
    unsigned b = 0;
    if (sk == 2) {
        const bool ok = ((T.kind == 0) & (s_ndx < T.ndx)) | ((T.kind == 3) & (s_ndx < T.ndx - 2));
        if (ok) b = 1u | (((T.kind == 3) & (T.ndx - s_ndx <= 3 * DPW_OPER_DIST)) ? 2u : 0u);
    } else if (sk == 3) {
        if (s_stop > T.ndx) b = (((T.kind == 2) & (T.frame == sf)) ? 1u : 0u) | (T.kind == 3 ? 2u : 0u);
    } else if (sk == 1) {
        if (T.kind == 0) { if (s_ndx + 2 < T.ndx) b = 1u | (T.ndx - s_ndx <= 3 * DPW_OPER_DIST ? 2u : 0u); }
        else if (T.kind == 1) { if (T.stop_val < s_ndx) b = 4u; }
        else if (T.kind == 2) { if ((s_ndx > T.dlo0) & (s_ndx < T.dhi0)) b = 8u; }
        else if (T.kind == 3) { if (s_ndx < T.ndx - 4) b = 16u; }
    }
    return b;
}
// lane c (target record Tc) is a forward start that forward stop k (frame fk, stop_val s_stop) pulls
DPW_HD bool dpw_static_pull(const DpwST& Tc, const int fk, const int s_stop) { return (Tc.kind == 0) & (Tc.frame == fk) & (Tc.ndx > s_stop); }

// The dynamic half of a step, per lane.  `b`: the lane's static bits of the entry.
// reverse start: the value it offers this lane
DPW_HD double dpw_sval_r5(const DpwLT& T, const unsigned b, const int s_ndx, const double s_score, const DpwModel& M) {
    return (b & 2u) ? s_score + dpw_igm_apart(T.ndx - s_ndx, M.negc, M.igm) : s_score + M.negc;
}
// reverse stop of frame sf
DPW_HD bool dpw_sok_r3(const DpwLT& T, const unsigned b, const int sf) { return (b & 1u) | ((b & 2u) && ((T.vm >> sf) & 1)); }
// forward stop: dpw_step_f3 with the static tests taken from the entry
DPW_HD void dpw_sstep_f3(const DpwLT& T, DpwLane& L, const unsigned b, const int j, const int s_ndx, const int s_vm, const int s_tbn, const double s_score,
                         const double s_x0, const double s_x1, const double s_x2, const DpwModel& M) {
    bool ok = false;
    double w = 0.0; int ov1 = 0;
    if (b & 1u) { ok = true; w = (b & 2u) ? dpw_igm_apart(T.ndx - s_ndx, M.negc, M.igm) : M.negc; }
    else if (b & 4u) { ok = ((s_vm >> T.frame) & 1) != 0; w = dpw_sel3(T.frame, s_x0, s_x1, s_x2); }
    else if (b & 8u) { ok = s_tbn + s_ndx + 7 < T.drhs0; w = T.csd; }
    else if (b & 16u) {
        const int lhs = s_tbn + s_ndx + 7;
        const bool c0 = (s_ndx > T.dlo0) & (s_ndx < T.dhi0) & (lhs < T.drhs0);
        const bool c1 = (s_ndx > T.dlo1) & (s_ndx < T.dhi1) & (lhs < T.drhs1);
        const bool c2 = (s_ndx > T.dlo2) & (s_ndx < T.dhi2) & (lhs < T.drhs2);
        double mv = 0.0; int m = -1;
        if (c0 & (T.x0 > mv)) { mv = T.x0; m = 0; }
        if (c1 & (T.x1 > mv)) { mv = T.x1; m = 1; }
        if (c2 & (T.x2 > mv)) { mv = T.x2; m = 2; }
        ok = true; w = m >= 0 ? mv : M.negc; ov1 = m + 1;
    }
    dpw_take_ge(L, ok, s_score + w, j | (ov1 << DPW_TAG_BITS));
}
