// Connection-scoring dynamic programme (Prodigal's dprog / score_connection) for gfx950.
//
// What is computed (ref: lib.pyx:1205-1237 `_score_connections`, _connection.h:94-408 the four
// split scorers, impl/generic.h:29-36 the six skip conditions): for every node i in position
// order, the best predecessor j in the window [lo_i, i):
//     score[i] = max(0, max_j (score[j] + w(j, i))),  ties -> largest j,  traceb[i] = that j.
//
// Mapping onto CDNA4 -- W wavefronts (1, 4 or 16) per (contig, model) chain, 64 targets at a time:
//   * lane t owns target node i0+t of the current batch and keeps (best, traceb, ov_mark) in
//     registers; a candidate source is wave-uniform: its fields sit in the registers of one lane
//     of a coalesced 64-source tile and are broadcast with v_readlane;
//   * a __ballot over the tile builds the visit mask (gene ends that were never reached connect
//     to nothing, ref: _connection.h:110-114) and a wave vote (__any) on the folded skip
//     conditions of impl/generic.h:29-36 drops a source before any f64 work;
//   * the reference's ascending ">=" scan is a lexicographic (value, index) maximum, so the W waves
//     scan disjoint tiles of the already-final sources concurrently and merge exactly in LDS;
//   * the 63 candidates inside the batch are the other lanes: when the walk reaches source i0+k,
//     lane k has met every j < i0+k, so its registers hold final values (this walk is the serial
//     critical path of a chain: one step per node);
//   * third-node indirections (star_ptr -> n3) never happen in the loop: dp_prepare folds
//     cs(n3)+igm(.,.) and n3.{ndx,stop_val} into the source/target records, and every lane tracks
//     the ndx of its own traceb node, which later pairs need (ref: _connection.h:249, 318).
//
// Few long chains (one genome) do not fill the chip with one serial walk each: they are cut into segments that are
// walked speculatively side by side, re-scored exactly and verified node by node -- see "Segmented chains" below.
//
// Arithmetic: IEEE double, compiled with -ffp-contract=off, same operation order as the
// reference; (2 - d/60)*0.15*st_wt comes from a host-computed 61-entry table.

#include "pga_internal.h"
#include "dev_common.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

namespace {

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m, 64));
    return v;
}

__device__ __forceinline__ double sel3(int k, double a, double b, double c) { return k == 0 ? a : (k == 1 ? b : c); }
__device__ __forceinline__ int sel3i(int k, int a, int b, int c) { return k == 0 ? a : (k == 1 ? b : c); }

// _intergenic_mod_same for two same-strand nodes `dist` apart that neither overlap nor touch
// (ref: _connection.h:52-78 with overlap == 0 and no adjacency bonus)
__device__ __forceinline__ double igm_apart(int d, double negc, const double* s_igm) {
    double r = 0.0;
    if (d > 3 * PGA_OPER_DIST) r = negc;
    else if (d <= PGA_OPER_DIST && d >= 0) r = s_igm[d];
    return r;
}

// One thread per node: build DpSrc / DpTgt (ref: lib.pyx:1126-1162 `_index`, 1221-1233 window,
// and the n3 terms of _connection.h:166-176, 296-325, 345-356).
__global__ void __launch_bounds__(256)
k_dp_prepare(const ChainDesc* __restrict__ chains, int n_chains, int64_t node_begin, int64_t total,
             NodeArrays nd, const ModelConst* __restrict__ models, DpSrc* __restrict__ src, DpTgt* __restrict__ tgt, int final) {
    __shared__ int s_c0;
    const int64_t blk0 = node_begin + (int64_t)blockIdx.x * blockDim.x;
    const int64_t g = blk0 + threadIdx.x;
    const bool in_range = g < node_begin + total;
    const int c = find_chain_block(chains, n_chains, blk0, in_range ? g : blk0, &s_c0);
    if (!in_range) return;
    const int64_t off = chains[c].off, toff = chains[c].topo_off;
    const int i = (int)(g - off);
    const ModelConst* mc = &models[chains[c].model];
    const int32_t* ndx = nd.ndx + toff; const int32_t* stopv = nd.stop_val + toff;
    const uint8_t* type = nd.type + toff; const int8_t* strand = nd.strand + toff;
    const double* cs_c = nd.cscore + off; const double* cs_s = nd.sscore + off;
    const double* rsc = nd.rscore + off; const double* usc = nd.uscore + off;
    const int32_t* sp = nd.star_ptr + off * 3;

    const int my_ndx = ndx[i], my_stop = stopv[i];
    const bool rev = strand[i] != 1, stop = type[i] == PGA_T_STOP;
    const int kind = (rev ? 2 : 0) | (stop ? 1 : 0);
    int meta = kind | ((my_ndx % 3) << 2);
    DpSrc s; DpTgt t;
    s.ndx = my_ndx; s.stop_val = my_stop; s._pad = 0;
    // training pass (final = 0): a connection is worth (length) x (bias . gc_score) of one of its nodes, so the
    // records carry that per-node factor where the gene-finding pass carries cscore + sscore  (ref: _connection.h:94-367)
    s.cs = final ? cs_c[i] + cs_s[i] : nd.gcb[off + i];
    s.n3src[0] = s.n3src[1] = s.n3src[2] = 0; s._pad2 = 0;
    for (int k = 0; k < 3; k++) {
        s.x[k] = 0.0; t.n3ndx[k] = 0; t.n3stop[k] = 0;
        if (!stop) continue;
        const int p = sp[i * 3 + k];
        if (p < 0) continue;
        meta |= 1 << (4 + k);
        if (!final) { s.x[k] = nd.gcb[off + p]; s.n3src[k] = ndx[p]; t.n3ndx[k] = ndx[p]; t.n3stop[k] = stopv[p]; continue; }
        const double cs3 = cs_c[p] + cs_s[p];
        double ig;
        if (!rev)   // F3 source j = i, n3 = forward start: igm(j, n3)       (ref: _connection.h:170-174)
            ig = (strand[p] == 1) ? igm_same_dev(my_ndx, 1, rsc[i], usc[i], ndx[p], rsc[p], usc[p], mc->st_wt, mc->igm) : mc->negc;
        else        // R3 target i, n3 = reverse start: igm(n3, i)          (ref: _connection.h:313-320, 353-355)
            ig = (strand[p] == -1) ? igm_same_dev(ndx[p], -1, rsc[p], usc[p], my_ndx, rsc[i], usc[i], mc->st_wt, mc->igm) : mc->negc;
        s.x[k] = cs3 + ig;
        t.n3ndx[k] = ndx[p]; t.n3stop[k] = stopv[p];
    }
    s.meta = meta;
    // window start (ref: lib.pyx:1221-1233): 500 nodes back, stretched to the far end of a giant ORF,
    // then another 500.  The reference's walk-down stops at the highest index whose ndx equals
    // stop_val, or at 0; positions are sorted, so a binary search finds the same index.
    int lo = i < PGA_MAX_NODE_DIST ? 0 : i - PGA_MAX_NODE_DIST;
    // (positions ascend and a position holds at most two nodes: node k - 2 d - 2 and everything before it lies more than d
    //  positions left of node k -- a search for position v starts there, not at 0: ten steps instead of twenty-three on a genome)
    if ((kind == 2 || kind == 1) && ndx[lo] > my_stop) {
        int a = max(0, lo - 2 * max(0, ndx[lo] - my_stop) - 2), b = lo;            // find last p in [0, lo) with ndx[p] <= my_stop
        while (a < b) { int m = (a + b) >> 1; if (ndx[m] <= my_stop) a = m + 1; else b = m; }
        lo = (a > 0 && ndx[a - 1] == my_stop) ? a - 1 : 0;
    }
    lo = lo < PGA_MAX_NODE_DIST ? 0 : lo - PGA_MAX_NODE_DIST;
    t.lo = lo; t.p_near = lo;
    for (int k = 0; k < 3; k++) { t.a[k] = 0; t.b[k] = 0; t.c[k] = -1; t._pad[k] = 0; }
    // static index ranges over the sorted positions of nodes [0, i)
    auto first_of = [&](int v) { return max(0, i - 2 * max(0, my_ndx - v) - 2); };      // every node before this index lies left of position v
    auto lower = [&](int v) { int a = first_of(v), b = i; while (a < b) { const int m = (a + b) >> 1; if (ndx[m] < v) a = m + 1; else b = m; } return a; };   // first ndx >= v
    auto upper = [&](int v) { int a = first_of(v), b = i; while (a < b) { const int m = (a + b) >> 1; if (ndx[m] <= v) a = m + 1; else b = m; } return a; };  // first ndx > v
    if (kind == 0 || kind == 3) t.p_near = max(lo, lower(my_ndx - 3 * PGA_OPER_DIST));
    if (kind == 1) t.a[0] = max(lo, upper(my_stop));
    if (kind == 2) {
        const int u = upper(my_stop) - 1;     // last node at position stop_val: the reverse one if both strands have a node there
        t.a[0] = (u >= lo && ndx[u] == my_stop && strand[u] != 1 && type[u] == PGA_T_STOP) ? u : -1;
        t.a[1] = max(lo, upper(my_stop - 4)); t.a[2] = max(t.a[1], lower(my_stop + PGA_MAX_OPP_OVLP - 5));
    }
    if (kind == 3) {
        for (int k = 0; k < 3; k++) {
            if (!((meta >> (4 + k)) & 1)) continue;
            t.a[k] = max(lo, upper(t.n3stop[k] - 5));
            t.b[k] = min(t.p_near, max(t.a[k], lower(t.n3stop[k] + PGA_MAX_OPP_OVLP - 5)));
            if (t.b[k] < t.a[k]) t.b[k] = t.a[k];
        }
        int seen = 0;                          // latest reverse stop of each frame before i
        for (int j = i - 1; j >= lo && seen != 7; j--) {
            if (strand[j] == 1 || type[j] != PGA_T_STOP) continue;
            const int fj = ndx[j] % 3;
            if (seen & (1 << fj)) continue;
            seen |= 1 << fj;
            if (stopv[j] > my_ndx && ((meta >> (4 + fj)) & 1)) t.c[fj] = j;
        }
    }
    src[g] = s; tgt[g] = t;
}

struct Target {
    int kind, frame, ndx, stop_val, meta, lo, i, p_near;
    int a0, a1, a2, b0, b1, b2, c0, c1, c2;
    double cs, csd, x0, x1, x2;
    int n3n0, n3n1, n3n2, n3s0, n3s1, n3s2;
};
// Per-lane copy of one candidate source (lane k holds source t0+k); fields are broadcast with
// v_readlane when the wave visits that source.
struct SrcLane {
    int ndx, stop_val, meta, tbn;     // tbn: ndx of the source's own traceb node, -1 if it has none
    double cs, x0, x1, x2, score;
    int n3a, n3b, n3c;                // training pass only: DpSrc::n3src
};
struct Best { double val; int tb, ov, tbn; };   // running result of a target + ndx of its traceb node

// "val >= best" of the reference's ascending scan (ref: _connection.h:135-139), written as a
// lexicographic (value, index) maximum so that partial results over disjoint source sets merge exactly.
__device__ __forceinline__ void take(Best& b, bool ok, double val, int j, int mf, int s_ndx) {
    if (ok && (val > b.val || (val == b.val && j > b.tb))) { b.val = val; b.tb = j; b.ov = mf; b.tbn = s_ndx; }
}

// Visit source j = lane k of S for all 64 targets of the wave.
// One call = one iteration of the loop in _connection.h:386-408, with the six skip conditions of
// impl/generic.h:29-36 folded into the per-kind predicates; a wave-wide vote (__any, i.e. a ballot)
// drops the source before any floating-point work when no lane can connect to it.
// FINAL = false is the training pass (ref: the `final` flag of _connection.h:94-367): same admissibility, but a connection
// is worth (right - left + 1 - 2 overlap) x (bias . gc_score of one of its nodes), 0 for intergenic steps.  In that
// pass S.cs / T.cs hold the node's own factor, S.x* / T.x* those of its overlapping starts, S.n3* their positions.
template <bool FINAL = true>
__device__ __forceinline__ void visit_source(const int k, const int j, const SrcLane& S, const Target& T,
                                             const double negc, const double* s_igm, Best& B, const double st_wt = 0.0) {
    const int s_meta = __builtin_amdgcn_readlane(S.meta, k);
    const int s_ndx = __builtin_amdgcn_readlane(S.ndx, k);
    const int sk = PGA_KIND(s_meta), sf = PGA_FRAME(s_meta);
    const bool inwin = (j >= T.lo) && (j < T.i);
    if (sk == 0) {
        // 5'fwd -> 3'fwd: a gene (ref: _connection.h:166-174; skip condition 5: same frame only)
        const bool ok = inwin && T.kind == 1 && T.frame == sf && T.stop_val < s_ndx;
        if (!__any(ok)) return;
        const double w = FINAL ? readlane_f64(S.cs, k) : (double)(T.ndx + 2 - s_ndx + 1) * readlane_f64(S.cs, k);
        take(B, ok, readlane_f64(S.score, k) + w, j, -1, s_ndx);
    } else if (sk == 2) {
        // 5'rev -> 5'fwd (ref: :125-130) and 5'rev -> 3'rev (ref: :337-342)
        const bool a = T.kind == 0 && s_ndx < T.ndx;
        const bool b = T.kind == 3 && s_ndx < T.ndx - 2;
        const bool ok = inwin && (a || b);
        if (!__any(ok)) return;
        const double w = !FINAL ? 0.0 : (b ? igm_apart(T.ndx - s_ndx, negc, s_igm) : negc);
        take(B, ok, readlane_f64(S.score, k) + w, j, -1, s_ndx);
    } else if (sk == 3) {
        // 3'rev -> 5'rev: a gene (ref: :228-235; skip condition 6) and 3'rev -> 3'rev operon (ref: :345-356)
        const int s_stop = __builtin_amdgcn_readlane(S.stop_val, k);
        const bool a = T.kind == 2 && T.frame == sf && s_stop > T.ndx;
        const bool b = T.kind == 3 && s_stop > T.ndx && PGA_SPVALID(T.meta, sf);
        const bool ok = inwin && (a || b);
        if (!__any(ok)) return;
        double w = a ? T.cs : sel3(sf, T.x0, T.x1, T.x2);
        if (!FINAL) w = (double)((a ? T.ndx : sel3i(sf, T.n3n0, T.n3n1, T.n3n2)) - (s_ndx - 2) + 1) * w;
        take(B, ok, readlane_f64(S.score, k) + w, j, -1, s_ndx);
    } else {
        // forward stop as source: connects to all four target kinds
        const int tbnj = __builtin_amdgcn_readlane(S.tbn, k);
        const double sj = readlane_f64(S.score, k);
        bool ok = inwin; double w; int mf = -1;
        if (T.kind == 0) {            // 3'fwd -> 5'fwd intergenic (ref: :117-124)
            ok = ok && (s_ndx + 2 < T.ndx);
            w = FINAL ? igm_apart(T.ndx - s_ndx, negc, s_igm) : 0.0;
        } else if (T.kind == 1) {     // 3'fwd -> 3'fwd operon through j's overlapping start (ref: :177-188)
            ok = ok && T.stop_val < s_ndx && PGA_SPVALID(s_meta, T.frame);
            w = sel3(T.frame, readlane_f64(S.x0, k), readlane_f64(S.x1, k), readlane_f64(S.x2, k));
            if (!FINAL) {
                const int n3 = sel3i(T.frame, __builtin_amdgcn_readlane(S.n3a, k), __builtin_amdgcn_readlane(S.n3b, k), __builtin_amdgcn_readlane(S.n3c, k));
                w = (double)(T.ndx + 2 - n3 + 1) * w;
            }
        } else if (T.kind == 2) {     // 3'fwd -> 5'rev overlapping opposite 3' ends (ref: :238-254)
            const int ovlp = (s_ndx + 2) - (T.stop_val - 2) + 1;
            ok = ok && !(T.stop_val - 2 >= s_ndx + 2) && ovlp < PGA_MAX_OPP_OVLP
                    && (s_ndx - T.stop_val) < (T.ndx - s_ndx + 3)
                    && (s_ndx - T.stop_val) < (T.stop_val - 3 - tbnj);
            w = FINAL ? T.csd : (double)(T.ndx - (T.stop_val - 2) + 1 - ovlp * 2) * T.cs;
        } else {                      // 3'fwd -> 3'rev, possibly through one of i's overlapping starts (ref: :288-336)
            const int left = s_ndx + 2, right = T.ndx - 2;
            ok = ok && left < right;
            double maxval = 0.0;
            int ovlp_last = 0;        // training pass: the reference's `ovlp` keeps the value of the last candidate it looked at
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const int n3s = sel3i(q, T.n3s0, T.n3s1, T.n3s2), n3n = sel3i(q, T.n3n0, T.n3n1, T.n3n2);
                const double xq = sel3(q, T.x0, T.x1, T.x2);
                const int ovlp = left - n3s + 3;
                const bool valid = PGA_SPVALID(T.meta, q) != 0;
                if (valid) ovlp_last = ovlp;
                const bool adm = valid && ovlp > 0 && ovlp < PGA_MAX_OPP_OVLP && ovlp < n3n - left && ovlp < n3s - tbnj - 2;
                if (FINAL) { if (adm && xq > maxval) { mf = q; maxval = xq; } }
                else if (adm && xq > maxval) {
                    // the reference compares the factor but remembers the gene-finding value, here the bare intergenic term
                    mf = q; maxval = igm_same_dev(n3n, -1, 0.0, 0.0, T.ndx, 0.0, 0.0, st_wt, s_igm);
                }
            }
            if (FINAL) w = mf != -1 ? maxval : negc;
            else w = (double)(right - left + 1 - ovlp_last * 2) * (mf != -1 ? sel3(mf, T.x0, T.x1, T.x2) : 0.0);
        }
        take(B, ok, sj + w, j, mf, s_ndx);
    }
}

// Per-lane (non-uniform) evaluation of one (source j, target) pair: same arithmetic as visit_source,
// every field loaded by the lane itself.  Used for the few pairs that need the exact pairwise term.
// Source fields of final nodes, read from global memory.
struct GlobalAcc {
    const DpSrc* __restrict__ src; const double* score; const int* tbn;
    __device__ __forceinline__ int meta(int j) const { return src[j].meta; }
    __device__ __forceinline__ int ndx(int j) const { return src[j].ndx; }
    __device__ __forceinline__ int stop_val(int j) const { return src[j].stop_val; }
    __device__ __forceinline__ double cs(int j) const { return src[j].cs; }
    __device__ __forceinline__ double x(int j, int f) const { return src[j].x[f]; }
    __device__ __forceinline__ double sc(int j) const { return score[j]; }
    __device__ __forceinline__ int tb_ndx(int j) const { return tbn[j]; }
};
// weight of the pair alone: ok = the connection is allowed, w its weight, mf the ov_mark it leaves; tbnj = ndx of the
// source's own traceb node (-1: none)
template <class Acc>
__device__ __forceinline__ void pair_weight(const int j, const Acc& S, const int tbnj, const Target& T, const double negc, const double* s_igm,
                                            bool& ok, double& w, int& mf) {
    const int s_meta = S.meta(j), s_ndx = S.ndx(j);
    const int sk = PGA_KIND(s_meta), sf = PGA_FRAME(s_meta);
    ok = false; w = 0.0; mf = -1;
    if ((sk == 1 || sk == 2) && tbnj == -1) return;
    ok = (j >= T.lo) && (j < T.i);
    if (sk == 0) {
        ok = ok && T.kind == 1 && T.frame == sf && T.stop_val < s_ndx;
        w = S.cs(j);
    } else if (sk == 2) {
        const bool a = T.kind == 0 && s_ndx < T.ndx;
        const bool b = T.kind == 3 && s_ndx < T.ndx - 2;
        ok = ok && (a || b);
        w = b ? igm_apart(T.ndx - s_ndx, negc, s_igm) : negc;
    } else if (sk == 3) {
        const int s_stop = S.stop_val(j);
        const bool a = T.kind == 2 && T.frame == sf && s_stop > T.ndx;
        const bool b = T.kind == 3 && s_stop > T.ndx && PGA_SPVALID(T.meta, sf);
        ok = ok && (a || b);
        w = a ? T.cs : sel3(sf, T.x0, T.x1, T.x2);
    } else {
        if (T.kind == 0) {
            ok = ok && (s_ndx + 2 < T.ndx);
            w = igm_apart(T.ndx - s_ndx, negc, s_igm);
        } else if (T.kind == 1) {
            ok = ok && T.stop_val < s_ndx && PGA_SPVALID(s_meta, T.frame);
            w = S.x(j, T.frame);
        } else if (T.kind == 2) {
            const int ovlp = (s_ndx + 2) - (T.stop_val - 2) + 1;
            ok = ok && !(T.stop_val - 2 >= s_ndx + 2) && ovlp < PGA_MAX_OPP_OVLP
                    && (s_ndx - T.stop_val) < (T.ndx - s_ndx + 3)
                    && (s_ndx - T.stop_val) < (T.stop_val - 3 - tbnj);
            w = T.csd;
        } else {
            const int left = s_ndx + 2, right = T.ndx - 2;
            ok = ok && left < right;
            double maxval = 0.0;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const int n3s = sel3i(q, T.n3s0, T.n3s1, T.n3s2), n3n = sel3i(q, T.n3n0, T.n3n1, T.n3n2);
                const double cur = sel3(q, T.x0, T.x1, T.x2);
                const int ovlp = left - n3s + 3;
                const bool tk = PGA_SPVALID(T.meta, q) && ovlp > 0 && ovlp < PGA_MAX_OPP_OVLP && ovlp < n3n - left
                                && ovlp < n3s - tbnj - 2 && cur > maxval;
                if (tk) { mf = q; maxval = cur; }
            }
            w = mf != -1 ? maxval : negc;
        }
    }
}
template <class Acc>
__device__ __forceinline__ void pair_eval(const int j, const Acc& S, const Target& T, const double negc, const double* s_igm, Best& B) {
    bool ok; double w; int mf;
    pair_weight(j, S, S.tb_ndx(j), T, negc, s_igm, ok, w, mf);
    if (ok) take(B, true, S.sc(j) + w, j, mf, S.ndx(j));
}

// ---------------------------------------------------------------------------------------------
// Tree kernels: the far field of the window is never scanned source by source.
//
// Every candidate value score[j] + w(j, i) whose w does not depend on the target is stored once,
// when node j becomes final:
//     A[j]    = score[j] + igm_diff          gene end j  -> any gene begin further than 3*OPER_DIST
//     V[f][j] = score[j] + cs[j]             forward start j of frame f -> the forward stop of its ORF
//             = score[j] + x[j][f]           forward stop j -> forward stop of frame f (operon)
// and the ascending ">=" scan of the reference equals a lexicographic (value, index) maximum, so
//   * a gene-begin target (F5 / R3) takes the maximum of A over [lo, p_near) from an 8-ary max tree
//     (O(log) block reads instead of ~1000 pair evaluations) and evaluates exactly only the sources
//     within 3*OPER_DIST bases, plus -- for R3 -- the forward stops that can overlap one of its
//     overlapping starts and the one reverse stop per frame whose ORF covers it;
//   * a forward stop scans V[frame] over its own ORF; a reverse start reads its own stop and the few
//     forward stops that can overlap its 3' end;
//   * sources inside the current 64-node batch are not final yet: they are walked in order, one
//     wave-uniform source at a time, which is the serial critical path of a chain.
// For an R3 target the tree may also return A[j] of a forward stop whose exact term is larger
// (the overlapping-start case adds a positive score): the exact pair is evaluated as well and wins.
struct ChainPtrs {
    const DpSrc* __restrict__ src; const DpTgt* __restrict__ tgt;
    int rebase;        // > 0: a segment's sub-chain that starts at this node of its chain (see load_target)
    double* score; int32_t* traceb; int32_t* tbn; int8_t* ovm;
    double* A; double* V0; double* V1; double* V2; double* hv; int32_t* hi;
};

__device__ __forceinline__ ChainPtrs chain_ptrs(const ChainDesc& cd, const DpSrc* g_src, const DpTgt* g_tgt, const DpBuffers& buf) {
    ChainPtrs P;
    const int64_t rec = cd.rec_off >= 0 ? cd.rec_off : cd.off;
    P.src = g_src + rec; P.tgt = g_tgt + rec; P.rebase = cd.rebase;
    P.score = buf.score + cd.off; P.traceb = buf.traceb + cd.off; P.tbn = buf.tbn + cd.off; P.ovm = buf.ov_mark + cd.off;
    P.A = buf.A + cd.off; P.V0 = buf.V[0] + cd.off; P.V1 = buf.V[1] + cd.off; P.V2 = buf.V[2] + cd.off;
    P.hv = buf.hv + cd.off; P.hi = buf.hi + cd.off;
    return P;
}

__device__ __forceinline__ void load_target(Target& T, const ChainPtrs& P, int i0, int lane, int n, double negc) {
    T.i = i0 + lane;
    const bool act = T.i < n;
    const int ii = act ? T.i : n - 1;
    const DpSrc me = P.src[ii]; const DpTgt mt = P.tgt[ii];
    T.kind = PGA_KIND(me.meta); T.frame = PGA_FRAME(me.meta); T.meta = me.meta;
    T.ndx = me.ndx; T.stop_val = me.stop_val; T.cs = me.cs; T.csd = me.cs + negc;
    T.x0 = me.x[0]; T.x1 = me.x[1]; T.x2 = me.x[2];
    T.n3n0 = mt.n3ndx[0]; T.n3n1 = mt.n3ndx[1]; T.n3n2 = mt.n3ndx[2];
    T.n3s0 = mt.n3stop[0]; T.n3s1 = mt.n3stop[1]; T.n3s2 = mt.n3stop[2];
    T.lo = mt.lo; T.p_near = mt.p_near;
    T.a0 = mt.a[0]; T.a1 = mt.a[1]; T.a2 = mt.a[2]; T.b0 = mt.b[0]; T.b1 = mt.b[1]; T.b2 = mt.b[2];
    T.c0 = mt.c[0]; T.c1 = mt.c[1]; T.c2 = mt.c[2];
    if (P.rebase > 0) {
        // chain indices -> sub-chain indices; what lies before the sub-chain does not exist for it: a range starts at 0 at
        // the earliest, a single node that lies before it becomes "none"
        const int a = P.rebase;
        auto rb = [a](const int v) { return v > a ? v - a : 0; };
        auto one = [a](const int v) { return v >= a ? v - a : -1; };
        T.lo = rb(T.lo); T.p_near = rb(T.p_near);
        if (T.kind == 1) T.a0 = rb(T.a0);
        else if (T.kind == 2) { T.a0 = one(T.a0); T.a1 = rb(T.a1); T.a2 = rb(T.a2); }
        else if (T.kind == 3) {
            T.a0 = rb(T.a0); T.a1 = rb(T.a1); T.a2 = rb(T.a2); T.b0 = rb(T.b0); T.b1 = rb(T.b1); T.b2 = rb(T.b2);
            T.c0 = one(T.c0); T.c1 = one(T.c1); T.c2 = one(T.c2);
        }
    }
    if (!act) { T.lo = INT_MAX; T.i = -1; }          // j < T.i is never true: the lane stays idle
}

// A and its block-of-8 maxima for the most recent PGA_RING nodes of a chain, kept in LDS by the wave that
// stores the batches.
#define PGA_RING 1024
#define PGA_RING_BLOCKS (PGA_RING / 8)
struct RingLds {
    double A[PGA_RING];
    int ndx[PGA_RING];                 // position of the node, for the ndx-of-traceb of a far-field result
    double l1v[PGA_RING_BLOCKS]; int l1i[PGA_RING_BLOCKS];
};
// Suffix maxima over the ring's blocks: v/i[e - ebase] = lexicographic (value, index) maximum of A over the
// blocks [e, eend).  Nearly every far-field range of a batch ends at the batch boundary 8 * eend, so one
// scan per batch replaces a tree descent per target: a range [p, 8 eend) is its ragged head (ring A) plus
// one suffix entry.  Built and read by one wave.
struct SuffixLds { double v[PGA_RING_BLOCKS]; int i[PGA_RING_BLOCKS]; };
__device__ __forceinline__ void suffix_build(const RingLds* ring, SuffixLds* sfx, const int ebase, const int eend, const int lane) {
    const double NEG_INF = -__builtin_huge_val();
    constexpr int PER = PGA_RING_BLOCKS / 64;
    double sv[PER]; int si[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int e = ebase + PER * lane + u;
        const bool in = e < eend;
        sv[u] = in ? ring->l1v[e & (PGA_RING_BLOCKS - 1)] : NEG_INF; si[u] = in ? ring->l1i[e & (PGA_RING_BLOCKS - 1)] : -1;
    }
#pragma unroll
    for (int u = PER - 2; u >= 0; u--)      // inside the lane: a later node wins a tie
        if (sv[u + 1] >= sv[u]) { sv[u] = sv[u + 1]; si[u] = si[u + 1]; }
    double tv = sv[0]; int ti = si[0];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {      // inclusive suffix scan over the lanes
        const double xv = __shfl_down(tv, d, 64); const int xi = __shfl_down(ti, d, 64);
        if (lane + d < 64 && xv >= tv) { tv = xv; ti = xi; }
    }
    double ev = __shfl_down(tv, 1, 64); int ei = __shfl_down(ti, 1, 64);
    if (lane == 63) { ev = NEG_INF; ei = -1; }
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const bool c = ev >= sv[u];
        sfx->v[PER * lane + u] = c ? ev : sv[u]; sfx->i[PER * lane + u] = c ? ei : si[u];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// lexicographic maximum of A over [l, r) from the 8-ary tree; the (up to 7 + 7) ragged entries of a
// level are loaded together so that a level costs one memory round trip.
__device__ __forceinline__ void tree_range(int l, int r, const ChainPtrs& P, const int* s_levbase, Best& B) {
    int lev = 0;
    while (l < r) {
        const int l_end = min(r, (l + 7) & ~7);
        const int r_beg = max(l_end, r & ~7);
        const int base = s_levbase[lev];
        const double* __restrict__ vp = lev == 0 ? P.A : P.hv + base;
        const int* __restrict__ ip = P.hi + base;
        double v[14]; int ix[14];
#pragma unroll
        for (int q = 0; q < 7; q++) {
            const int pl = l + q, pr = r_beg + q;
            const bool okl = pl < l_end, okr = pr < r;
            const int cl = okl ? pl : l, cr = okr ? pr : l;
            v[q] = vp[cl]; v[7 + q] = vp[cr];
            ix[q] = lev == 0 ? cl : ip[cl]; ix[7 + q] = lev == 0 ? cr : ip[cr];
            if (!okl) v[q] = -__builtin_huge_val();
            if (!okr) v[7 + q] = -__builtin_huge_val();
        }
#pragma unroll
        for (int q = 0; q < 14; q++) take(B, true, v[q], ix[q], -1, 0);
        l = l_end >> 3; r = r_beg >> 3; lev++;
    }
}

// Everything target T can take from final sources j in [clo, chi): tree + exact near pairs + special
// ranges.  chi must not exceed the number of final nodes.
__device__ __forceinline__ void far_field(const Target& T, int clo, int chi, const ChainPtrs& P, const int* s_levbase,
                                          const double negc, const double* s_igm, Best& B,
                                          const RingLds* ring = nullptr, const SuffixLds* sfx = nullptr, const int ebase = 0, const int eend = 0) {
    if (T.i < 0) return;
    chi = min(chi, T.i);
    clo = max(clo, T.lo);
    if (clo >= chi) return;
    const GlobalAcc G{P.src, P.score, P.tbn};
    if (T.kind == 0 || T.kind == 3) {
        // (1) far gene ends
        const int fhi = min(T.p_near, chi);
        if (sfx != nullptr && fhi == 8 * eend) {
            int p = clo;
            if (p < 8 * ebase) { tree_range(p, 8 * ebase, P, s_levbase, B); p = 8 * ebase; }
            const int e0 = (p + 7) >> 3;
            for (; p < min(8 * e0, fhi); p++) take(B, true, ring->A[p & (PGA_RING - 1)], p, -1, 0);      // ragged head
            if (e0 < eend) take(B, true, sfx->v[e0 - ebase], sfx->i[e0 - ebase], -1, 0);
        } else tree_range(clo, fhi, P, s_levbase, B);
        for (int j = max(T.p_near, clo); j < chi; j++) pair_eval(j, G, T, negc, s_igm, B);   // (2) near: exact pairs
        if (T.kind == 3) {
            // (3) forward stops that can overlap one of this node's overlapping starts (ref: _connection.h:296-325)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (!PGA_SPVALID(T.meta, k)) continue;
                const int zb = min(sel3i(k, T.b0, T.b1, T.b2), chi);
                for (int j = max(sel3i(k, T.a0, T.a1, T.a2), clo); j < zb; j++)
                    if (PGA_KIND(P.src[j].meta) == 1) pair_eval(j, G, T, negc, s_igm, B);
            }
            // (4) the reverse stop of each frame whose ORF covers this node (ref: _connection.h:345-356)
#pragma unroll
            for (int f = 0; f < 3; f++) {
                const int j = sel3i(f, T.c0, T.c1, T.c2);
                if (j >= clo && j < chi) pair_eval(j, G, T, negc, s_igm, B);
            }
        }
    } else if (T.kind == 1) {
        // forward stop: starts of its own ORF and operon partners, precomputed per target frame
        for (int j = max(T.a0, clo); j < chi; j++) {
            const double v = T.frame == 0 ? P.V0[j] : (T.frame == 1 ? P.V1[j] : P.V2[j]);
            take(B, true, v, j, -1, 0);
        }
    } else {
        // reverse start: its own stop, then forward stops overlapping its 3' end (ref: _connection.h:228-254)
        if (T.a0 >= clo && T.a0 < chi) pair_eval(T.a0, G, T, negc, s_igm, B);
        const int zb = min(T.a2, chi);
        for (int j = max(T.a1, clo); j < zb; j++)
            if (PGA_KIND(P.src[j].meta) == 1) pair_eval(j, G, T, negc, s_igm, B);
    }
}

// The batch [i0, i0+64) is final in wave registers: store it with its far-field candidate values and
// extend the tree.  Returns nothing; updates the running _find_max_index state.
// PART selects what this call stores (the chain kernel deals the parts to different waves):
//   1  score, traceb, ov_mark, ndx of the traceb node; the running _find_max_index state
//   2  A (far gene-end candidate value), also into the LDS ring
//   4  V0..V2 (candidate values towards forward stops)
//   8  the lowest tree level above A (blocks of 8, also into the LDS ring)
//  16  the tree levels above that
template <int PART = 31>
__device__ __forceinline__ void finalize_batch(const Target& T, const Best& B, int i0, int lane, int n, const ChainPtrs& P,
                                               const int* s_levbase, const double negc,
                                               double& end_best, int& end_idx, int& end_tb, RingLds* ring = nullptr) {
    const double NEG_INF = -__builtin_huge_val();
    const bool act = T.i >= 0;
    double a_val = NEG_INF;
    if (act) {
        const bool alive = B.tb != -1;
        if (PART & 1) {
            P.score[T.i] = B.val; P.traceb[T.i] = B.tb; P.ovm[T.i] = (int8_t)B.ov; P.tbn[T.i] = alive ? B.tbn : -1;
            if ((T.kind == 1 || T.kind == 2) && B.val >= end_best) { end_best = B.val; end_idx = T.i; end_tb = B.tb; }
        }
        double v0 = NEG_INF, v1 = NEG_INF, v2 = NEG_INF;
        if (T.kind == 0) {
            const double g = B.val + T.cs;
            if (T.frame == 0) v0 = g; else if (T.frame == 1) v1 = g; else v2 = g;
        } else if (T.kind == 1 && alive) {
            a_val = B.val + negc;
            if (PGA_SPVALID(T.meta, 0)) v0 = B.val + T.x0;
            if (PGA_SPVALID(T.meta, 1)) v1 = B.val + T.x1;
            if (PGA_SPVALID(T.meta, 2)) v2 = B.val + T.x2;
        } else if (T.kind == 2 && alive) {
            a_val = B.val + negc;
        }
        if (PART & 4) { P.V0[T.i] = v0; P.V1[T.i] = v1; P.V2[T.i] = v2; }
        if (PART & 2) {
            P.A[T.i] = a_val;
            if (ring) { ring->A[T.i & (PGA_RING - 1)] = a_val; ring->ndx[T.i & (PGA_RING - 1)] = T.ndx; }
        }
    }
    if (!(PART & 24) || i0 + 64 > n) return;
    double rv = a_val; int ri = i0 + lane;
#pragma unroll
    for (int m = 1; m <= 4; m <<= 1) {
        const double ov2 = __shfl_xor(rv, m, 64); const int oi = __shfl_xor(ri, m, 64);
        if (ov2 > rv || (ov2 == rv && oi > ri)) { rv = ov2; ri = oi; }
    }
    const int tile_no = i0 >> 6;
    if ((PART & 8) && (lane & 7) == 0) {
        P.hv[s_levbase[1] + (i0 >> 3) + (lane >> 3)] = rv; P.hi[s_levbase[1] + (i0 >> 3) + (lane >> 3)] = ri;
        if (ring) { ring->l1v[((i0 >> 3) + (lane >> 3)) & (PGA_RING_BLOCKS - 1)] = rv; ring->l1i[((i0 >> 3) + (lane >> 3)) & (PGA_RING_BLOCKS - 1)] = ri; }
    }
    if (!(PART & 16)) return;
#pragma unroll
    for (int m = 8; m <= 32; m <<= 1) {
        const double ov2 = __shfl_xor(rv, m, 64); const int oi = __shfl_xor(ri, m, 64);
        if (ov2 > rv || (ov2 == rv && oi > ri)) { rv = ov2; ri = oi; }
    }
    if (lane == 0) { P.hv[s_levbase[2] + tile_no] = rv; P.hi[s_levbase[2] + tile_no] = ri; }
    // higher levels: a block of 8 children closes when its last child does
    int child = tile_no, lev = 2;
    while ((child & 7) == 7 && (lev + 1) * 3 < 31 && (n >> (3 * (lev + 1))) > 0) {
        double cv = NEG_INF; int ci = -1;
        if (lane < 8) { cv = P.hv[s_levbase[lev] + child - 7 + lane]; ci = P.hi[s_levbase[lev] + child - 7 + lane]; }
#pragma unroll
        for (int m = 1; m <= 4; m <<= 1) {
            const double ov2 = __shfl_xor(cv, m, 64); const int oi = __shfl_xor(ci, m, 64);
            if (ov2 > cv || (ov2 == cv && oi > ci)) { cv = ov2; ci = oi; }
        }
        child >>= 3; lev++;
        if (lane == 0) { P.hv[s_levbase[lev] + child] = cv; P.hi[s_levbase[lev] + child] = ci; }
    }
}

__device__ __forceinline__ void init_levbase(int* s_levbase, int n) {
    int base = 0;
    s_levbase[0] = 0;
    for (int lev = 1; lev < 12; lev++) { s_levbase[lev] = base; base += (lev * 3 < 31) ? (n >> (3 * lev)) : 0; }
}

__device__ __forceinline__ void publish_max(double end_best, int end_idx, int end_tb, int lane, const DpBuffers& buf, const int slot) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ob = __shfl_xor(end_best, m, 64);
        const int oi = __shfl_xor(end_idx, m, 64);
        const int ot = __shfl_xor(end_tb, m, 64);
        if (ob > end_best || (ob == end_best && oi > end_idx)) { end_best = ob; end_idx = oi; end_tb = ot; }
    }
    if (lane == 0) {   // highest score among gene ends, ties to the largest index (ref: lib.pyx:1239-1251, 1311)
        buf.max_index[slot] = end_idx; buf.max_score[slot] = end_idx >= 0 ? end_best : 0.0;
        buf.ipath[slot] = (end_idx >= 0 && end_tb != -1) ? end_idx : -1;
    }
}

// In-batch walk, general form: source k's registers are broadcast and every pair is evaluated in full.
__device__ __forceinline__ void walk_batch(const Target& T, Best& B, int i0, int n, const double negc, const double* s_igm) {
    const int kmax = min(63, n - 1 - i0);
    for (int k = 0; k < kmax; k++) {
        const int sk = PGA_KIND(__builtin_amdgcn_readlane(T.meta, k));
        const int tbk = __builtin_amdgcn_readlane(B.tb, k);
        if ((sk == 1 || sk == 2) && tbk == -1) continue;
        SrcLane S;
        S.ndx = T.ndx; S.stop_val = T.stop_val; S.meta = T.meta; S.tbn = B.tbn;
        S.cs = T.cs; S.x0 = T.x0; S.x1 = T.x1; S.x2 = T.x2; S.score = B.val;
        visit_source(k, i0 + k, S, T, negc, s_igm, B);
    }
}

// One wavefront per chain: used when there are enough chains to fill the chip.
__global__ void __launch_bounds__(64)
k_dp_tree(const ChainDesc* __restrict__ chains, const DpSrc* __restrict__ g_src, const DpTgt* __restrict__ g_tgt,
          const ModelConst* __restrict__ models, DpBuffers buf) {
    __shared__ double s_igm[64];
    __shared__ int s_levbase[12];
    const ChainDesc cd = chains[blockIdx.x];
    const int lane = threadIdx.x;
    const int n = cd.n;
    const ModelConst* mc = &models[cd.model];
    s_igm[lane] = mc->igm[lane];
    if (lane == 0) init_levbase(s_levbase, n);
    __syncthreads();
    const double negc = mc->negc;
    const ChainPtrs P = chain_ptrs(cd, g_src, g_tgt, buf);
    double end_best = -1.0; int end_idx = -1, end_tb = -1;
    for (int i0 = 0; i0 < n; i0 += 64) {
        Target T;
        load_target(T, P, i0, lane, n, negc);
        Best B{0.0, -1, -1, -1};
        const bool prof = buf.prof != nullptr && blockIdx.x == 0;
        const unsigned long long tp0 = prof ? __builtin_readcyclecounter() : 0;
        far_field(T, 0, i0, P, s_levbase, negc, s_igm, B);
        if (B.tb >= 0) B.tbn = P.src[B.tb].ndx;
        const unsigned long long tp1 = prof ? __builtin_readcyclecounter() : 0;
        walk_batch(T, B, i0, n, negc, s_igm);
        const unsigned long long tp2 = prof ? __builtin_readcyclecounter() : 0;
        finalize_batch(T, B, i0, lane, n, P, s_levbase, negc, end_best, end_idx, end_tb);
        if (prof && lane == 0) {
            const unsigned long long tp3 = __builtin_readcyclecounter();
            buf.prof[0] += tp1 - tp0; buf.prof[1] += tp2 - tp1; buf.prof[2] += tp3 - tp2; buf.prof[5] += 1;
        }
    }
    publish_max(end_best, end_idx, end_tb, lane, buf, blockIdx.x);
}

// Reverse targets against a forward-stop source at s_ndx: every static condition of the dynamic rule is an
// interval test on s_ndx, precomputed once per batch (candidate q is possible iff lo[q] < s_ndx < hi[q]).
//   reverse stop, candidate q  (ref: _connection.h:296-325), ovlp = s_ndx + 5 - n3s:
//       ovlp > 0  <=>  s_ndx > n3s - 5;   ovlp < MAX_OPP_OVLP  <=>  s_ndx < n3s + MAX_OPP_OVLP - 5;
//       ovlp < n3n - (s_ndx + 2)  <=>  2 s_ndx < n3n + n3s - 7;   left < ndx - 2  <=>  s_ndx < ndx - 4
//   reverse start, its one candidate  (ref: _connection.h:238-254):
//       stop_val - 2 < s_ndx + 2;   s_ndx - stop_val + 5 < MAX_OPP_OVLP;   2 s_ndx < ndx + stop_val + 3
struct RevRegs { int lo0, hi0, lo1, hi1, lo2, hi2, okhi; };
__device__ __forceinline__ RevRegs rev_regs(const Target& T) {
    RevRegs R;
    R.lo0 = R.lo1 = R.lo2 = INT_MAX; R.hi0 = R.hi1 = R.hi2 = INT_MIN; R.okhi = INT_MIN;
    if (T.kind == 3) {
        R.okhi = T.ndx - 4;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int n3s = sel3i(q, T.n3s0, T.n3s1, T.n3s2), n3n = sel3i(q, T.n3n0, T.n3n1, T.n3n2);
            if (PGA_SPVALID(T.meta, q) && sel3(q, T.x0, T.x1, T.x2) > 0.0) {
                const int lo = n3s - 5;
                const int hi = min(min(n3s + PGA_MAX_OPP_OVLP - 5, (n3n + n3s - 6) >> 1), R.okhi);
                if (q == 0) { R.lo0 = lo; R.hi0 = hi; } else if (q == 1) { R.lo1 = lo; R.hi1 = hi; } else { R.lo2 = lo; R.hi2 = hi; }
            }
        }
    } else if (T.kind == 2) {
        R.lo0 = T.stop_val - 4;
        R.hi0 = min(T.stop_val + PGA_MAX_OPP_OVLP - 5, (T.ndx + T.stop_val + 4) >> 1);
    }
    return R;
}

// Static part of a pair: source = lane k of batch S (chain index j), target = this lane of batch T.
// Weight and admissibility that do not depend on the source's running state.  What remains dynamic is
// a forward-stop source (position s_ndx, its traceb node at tbnj) towards a reverse target: it goes
// through candidate q only when "tbnj + s_ndx + 7 < rhs[q]" (DynRegs), and `flags` bit q says whether
// candidate q passes every static condition:
//   reverse start target   one candidate, the overlapping 3' ends rule       (ref: _connection.h:238-254)
//   reverse stop target    one candidate per overlapping start of the target (ref: _connection.h:296-325)
// ok/w describe the pair when no candidate is taken (w = NaN-able base weight).
// Written with selects only: the source kind is uniform (one scalar branch per row), everything per target
// lane is branch-free.
__device__ __forceinline__ double igm_apart_sel(const int d, const double negc, const double* s_igm) {
    const double tab = s_igm[min(max(d, 0), PGA_OPER_DIST)];
    const double r = (unsigned)d <= (unsigned)PGA_OPER_DIST ? tab : 0.0;
    return d > 3 * PGA_OPER_DIST ? negc : r;
}
__device__ __forceinline__ void static_pair(const int k, const int j, const Target& S, const Target& T, const RevRegs& R,
                                            const double negc, const double* s_igm, bool& ok, double& w, int& flags) {
    const int s_meta = __builtin_amdgcn_readlane(S.meta, k);
    const int s_ndx = __builtin_amdgcn_readlane(S.ndx, k);
    const int sk = PGA_KIND(s_meta), sf = PGA_FRAME(s_meta);
    const bool inwin = (j >= T.lo) & (j < T.i);
    flags = 0;
    if (sk == 0) {
        ok = inwin & (T.kind == 1) & (T.frame == sf) & (T.stop_val < s_ndx);
        w = readlane_f64(S.cs, k);
    } else if (sk == 2) {
        const bool a = (T.kind == 0) & (s_ndx < T.ndx);
        const bool b = (T.kind == 3) & (s_ndx < T.ndx - 2);
        ok = inwin & (a | b);
        const double g = igm_apart_sel(T.ndx - s_ndx, negc, s_igm);
        w = b ? g : negc;
    } else if (sk == 3) {
        const int s_stop = __builtin_amdgcn_readlane(S.stop_val, k);
        const bool a = (T.kind == 2) & (T.frame == sf) & (s_stop > T.ndx);
        const bool b = (T.kind == 3) & (s_stop > T.ndx) & (PGA_SPVALID(T.meta, sf) != 0);
        ok = inwin & (a | b);
        const double tx0 = T.x0, tx1 = T.x1, tx2 = T.x2;
        const double xs = sel3(sf, tx0, tx1, tx2);      // uniform choice
        w = a ? T.cs : xs;
    } else {
        const double sx0 = readlane_f64(S.x0, k), sx1 = readlane_f64(S.x1, k), sx2 = readlane_f64(S.x2, k);
        const double g = igm_apart_sel(T.ndx - s_ndx, negc, s_igm);
        const double sxf = sel3(T.frame, sx0, sx1, sx2);
        const bool ok0 = s_ndx + 2 < T.ndx;
        const bool ok1 = (T.stop_val < s_ndx) & (((s_meta >> (4 + T.frame)) & 1) != 0);
        const bool ok2 = s_ndx < R.okhi;         // a reverse start is admissible only through its candidate
        const bool rev = T.kind >= 2;
        ok = inwin & (T.kind == 0 ? ok0 : (T.kind == 1 ? ok1 : ok2));
        w = T.kind == 0 ? g : (T.kind == 1 ? sxf : negc);
        const int f = (((s_ndx > R.lo0) & (s_ndx < R.hi0)) ? 1 : 0) | (((s_ndx > R.lo1) & (s_ndx < R.hi1)) ? 2 : 0) |
                      (((s_ndx > R.lo2) & (s_ndx < R.hi2)) ? 4 : 0);
        flags = (inwin & rev) ? f : 0;
    }
}

// Per-target registers of the dynamic rule (see static_pair):
//   reverse stop   ovlp < n3s - tbnj - 2,  ovlp = s_ndx + 5 - n3s       <=>  tbnj + s_ndx + 7 < 2 n3s
//   reverse start  (s_ndx - stop_val) < (stop_val - 3 - tbnj)            <=>  tbnj + s_ndx + 7 < 2 stop_val + 4
struct DynRegs { int rhs0, rhs1, rhs2; double cur0, floor0; bool is_r3; };
__device__ __forceinline__ DynRegs dyn_regs(const Target& T) {
    DynRegs D;
    D.is_r3 = T.kind == 3;
    D.rhs0 = D.is_r3 ? 2 * T.n3s0 : 2 * T.stop_val + 4; D.rhs1 = 2 * T.n3s1; D.rhs2 = 2 * T.n3s2;
    D.cur0 = D.is_r3 ? T.x0 : T.csd;
    D.floor0 = D.is_r3 ? 0.0 : -__builtin_huge_val();   // a reverse start takes its candidate whatever its value
    return D;
}
// weight of a dynamic pair: the best admitted candidate, else the base weight; ov1 = ov_mark + 1
__device__ __forceinline__ void dyn_weight(const int s_ndx, const int tbnj, const int flags, const Target& T, const DynRegs& D,
                                           double& w, int& ov1) {
    const int lhs = tbnj + s_ndx + 7;
    double mv = D.floor0; int m = -1;
    if (((flags & 1) != 0) & (lhs < D.rhs0) & (D.cur0 > mv)) { mv = D.cur0; m = 0; }
    if (((flags & 2) != 0) & (lhs < D.rhs1) & (T.x1 > mv)) { mv = T.x1; m = 1; }
    if (((flags & 4) != 0) & (lhs < D.rhs2) & (T.x2 > mv)) { mv = T.x2; m = 2; }
    w = m >= 0 ? mv : w;
    ov1 = D.is_r3 ? m + 1 : 0;
}

// Workgroup barrier that orders LDS traffic only: global loads issued before it (next batch's records) stay in
// flight across it instead of being waited for, as __syncthreads() would.
__device__ __forceinline__ void barrier_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// What the other waves need of the batch the serial wave finalized last.
struct TileFin {
    double score[64];
    int ndx[64], tbn[64];               // tbn: ndx of the node's traceb node, -1 if none
    int tb[64], ov[64];
    int meta[64]; double cs[64], x0[64], x1[64], x2[64];    // static fields the storing waves need
    unsigned long long dead;            // gene ends without a traceb: they connect to nothing
};

// Sixteen wavefronts per chain, for the latency-bound case of few long chains.  Every pair inside the
// last 128 nodes is reduced to "score[j] + w(j, i)" with w precomputed off the critical path; each role
// runs its own loop and all of them meet at two workgroup barriers per 64-node batch:
//   wave 0        the serial wave: merges the partial results, walks the batch (lane k is final once the
//                 walk reaches source i0+k) and leaves it in LDS;
//   waves 1-11    precompute, for the NEXT batch, the weights of its in-batch pairs and of the pairs
//                 from this batch into it (no dependence on scores);
//   waves 12-15   one per target kind: far field of the NEXT batch over everything older than this
//                 batch (block suffix maxima for far gene ends, exact pairs and V arrays from global
//                 memory); before that, while the others wait for nothing else, they store the batch
//                 the serial wave finalized last and extend the tree, one part each;
//   waves 0-7     once a batch is final, each applies a slice of its 64 nodes to the next batch's targets.
#define PGA_MW_WAVES 16
#define PGA_MW_HELPERS 11
#define PGA_MW_SLICES 8        // waves that take a slice of the previous batch
__global__ void __launch_bounds__(64 * PGA_MW_WAVES)
k_dp_tree_mw(const ChainDesc* __restrict__ chains, const DpSrc* __restrict__ g_src, const DpTgt* __restrict__ g_tgt,
             const ModelConst* __restrict__ models, DpBuffers buf, const int32_t* __restrict__ gate, const int32_t* __restrict__ slot_map) {
    __shared__ double s_igm[64];
    __shared__ int s_levbase[12];
    __shared__ double s_w[2][64][64];                 // in-batch weights [slot][source k][target lane], NaN = pair not allowed
    __shared__ double s_wp[64][64];                   // weights from the previous batch's sources into this batch
    __shared__ unsigned char s_fl[2][64][64], s_flp[64][64];      // candidate flags of the dynamic pairs
    __shared__ unsigned long long s_dyn[2][64], s_dynp[64];       // per source: lanes with a dynamic pair
    __shared__ double s_eval[2][64];                  // early far-field result of the next batch
    __shared__ int s_etb[2][64], s_eov[2][64], s_etbn[2][64];
    __shared__ double s_pval[PGA_MW_SLICES][64];      // partial results over the previous batch, one slice per wave
    __shared__ int s_ptbx[PGA_MW_SLICES][64];
    __shared__ TileFin s_fin;
    __shared__ RingLds s_ring;
    __shared__ SuffixLds s_sfx[2];                    // one per far-field wave that queries far gene ends
    if (gate != nullptr && gate[blockIdx.x] == 0) return;          // segmented launches: this chain is not (or no longer) wanted
    const int slot = slot_map != nullptr ? slot_map[blockIdx.x] : (int)blockIdx.x;
    const ChainDesc cd = chains[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = cd.n;
    const ModelConst* mc = &models[cd.model];
    if (wave == 0) { s_igm[lane] = mc->igm[lane]; if (lane == 0) init_levbase(s_levbase, n); }
    __syncthreads();
    const double negc = mc->negc;
    const ChainPtrs P = chain_ptrs(cd, g_src, g_tgt, buf);
    const double QNAN = __builtin_nan("");
    double end_best = -1.0; int end_idx = -1, end_tb = -1;
    const int nb = (n + 63) >> 6;
    const bool prof = buf.prof != nullptr && blockIdx.x == 0;
    constexpr int SLICE = 64 / PGA_MW_SLICES;

    if (n <= 0) {                                    // empty chain (contig without nodes): nothing to walk
        if (wave == 0) publish_max(end_best, end_idx, end_tb, lane, buf, slot);
        return;
    }
    // weights of batch `bn` (targets Tq at chain index iq) in slot `sl`, and of the pairs from batch Tp into it
    auto helper_weights = [&](const Target& Tp, const Target& Tq, const int iq, const int sl, const bool with_prev) {
        const int kmax = min(63, n - 1 - iq);
        const RevRegs R = rev_regs(Tq);
        for (int r = wave - 1; r < (with_prev ? 128 : 64); r += PGA_MW_HELPERS) {
            bool ok = false; double w = 0.0; int fl = 0;
            if (r < 64) {
                if (r < kmax) static_pair(r, iq + r, Tq, Tq, R, negc, s_igm, ok, w, fl);
                s_w[sl][r][lane] = ok ? w : QNAN;
                const unsigned long long dm = __ballot(fl != 0);
                if (dm) s_fl[sl][r][lane] = (unsigned char)fl;
                if (lane == 0) s_dyn[sl][r] = dm;
            } else {
                const int k = r - 64;
                static_pair(k, iq - 64 + k, Tp, Tq, R, negc, s_igm, ok, w, fl);
                s_wp[k][lane] = ok ? w : QNAN;
                const unsigned long long dm = __ballot(fl != 0);
                if (dm) s_flp[k][lane] = (unsigned char)fl;
                if (lane == 0) s_dynp[k] = dm;
            }
        }
    };
    // store the batch at chain index ib (final in s_fin) with its far-field candidate values, extend the tree,
    // track _find_max_index
    auto flush = [&](const int ib, const int part) {
        Target Tf;
        Tf.i = ib + lane < n ? ib + lane : -1;
        const int meta = s_fin.meta[lane];
        Tf.kind = PGA_KIND(meta); Tf.frame = PGA_FRAME(meta); Tf.meta = meta; Tf.ndx = s_fin.ndx[lane];
        Tf.cs = s_fin.cs[lane]; Tf.x0 = s_fin.x0[lane]; Tf.x1 = s_fin.x1[lane]; Tf.x2 = s_fin.x2[lane];
        const Best Bf{s_fin.score[lane], s_fin.tb[lane], s_fin.ov[lane], s_fin.tbn[lane]};
        // four waves, about the same work each: results + running maximum | upper tree levels | V arrays + lowest
        // tree level | A with the ring
        if (part == 0) finalize_batch<1>(Tf, Bf, ib, lane, n, P, s_levbase, negc, end_best, end_idx, end_tb);
        else if (part == 1) finalize_batch<16>(Tf, Bf, ib, lane, n, P, s_levbase, negc, end_best, end_idx, end_tb);
        else if (part == 2) finalize_batch<4 | 8>(Tf, Bf, ib, lane, n, P, s_levbase, negc, end_best, end_idx, end_tb, &s_ring);
        else finalize_batch<2>(Tf, Bf, ib, lane, n, P, s_levbase, negc, end_best, end_idx, end_tb, &s_ring);
    };
    // one slice of the batch finalized last, applied to this batch's targets (ascending inside the slice)
    auto consume = [&](const Target& T, const DynRegs& D, const int i0) {
        double pv = 0.0; int ptx = -1;
        if (i0 > 0) {
            const unsigned long long deadp = s_fin.dead;
            const unsigned long long dynp = __ballot(s_dynp[lane] != 0ull);
            const int q0 = wave * SLICE;
            double wq[SLICE];
#pragma unroll
            for (int u = 0; u < SLICE; u++) wq[u] = s_wp[q0 + u][lane];
#pragma unroll
            for (int u = 0; u < SLICE; u++) {
                const int q = q0 + u;
                if ((deadp >> q) & 1ull) continue;
                double wk = wq[u]; int tag = i0 - 64 + q;
                if ((dynp >> q) & 1ull) {
                    const int s_ndx = __builtin_amdgcn_readfirstlane(s_fin.ndx[q]);
                    const int tbnj = __builtin_amdgcn_readfirstlane(s_fin.tbn[q]);
                    int ov1;
                    dyn_weight(s_ndx, tbnj, s_flp[q][lane], T, D, wk, ov1);
                    tag |= ov1 << 28;
                }
                const double val = s_fin.score[q] + wk;
                const bool c = val >= pv;
                pv = c ? val : pv; ptx = c ? tag : ptx;
            }
        }
        s_pval[wave][lane] = pv; s_ptbx[wave][lane] = ptx;
    };

    // Each role runs its own loop (its own register budget); all of them meet at the same two barriers per batch:
    //   barrier B  the slices of the previous batch are in LDS, the previous batch is in global memory
    //   barrier A  this batch is final in s_fin, the next batch's weights and early far field are in LDS
    if (wave == 0) {
        // ---------------------------------------------------------------- the serial wave
        __builtin_amdgcn_s_setprio(3);           // it shares its SIMD with three helper waves: issue it first
        Target T;
        load_target(T, P, 0, lane, n, negc);
        __syncthreads();
        for (int b = 0; b < nb; b++) {
            const int i0 = b << 6, slot = b & 1, nx = i0 + 64;
            const unsigned long long tq0 = prof ? __builtin_readcyclecounter() : 0;
            const DynRegs D = dyn_regs(T);
            consume(T, D, i0);
            if (prof && lane == 0) buf.prof[14] += __builtin_readcyclecounter() - tq0;
            barrier_lds();
            Target Tnext;
            if (nx < n) load_target(Tnext, P, nx, lane, n, negc);      // in flight during the walk
            const unsigned long long tq1 = prof ? __builtin_readcyclecounter() : 0;
            // merge, oldest sources first: the early far field, then the slices in order.  Ascending order makes
            // the lexicographic test the reference's plain ">=" (ref: _connection.h:135-139).
            double bv = s_eval[slot][lane];
            int tbx = s_etb[slot][lane] < 0 ? -1 : (s_etb[slot][lane] | ((s_eov[slot][lane] + 1) << 28));
#pragma unroll
            for (int w = 0; w < PGA_MW_SLICES; w++) {
                const double v = s_pval[w][lane]; const int t = s_ptbx[w][lane];
                const bool c = t >= 0 && v >= bv;
                bv = c ? v : bv; tbx = c ? t : tbx;
            }
            Best B;
            B.val = bv; B.tb = tbx < 0 ? -1 : (tbx & 0x0fffffff); B.ov = tbx < 0 ? -1 : (tbx >> 28) - 1; B.tbn = -1;
            // lean in-batch walk, fully unrolled: the static weight w(k, lane) comes from LDS one chunk (4 steps)
            // ahead; only the recurrence (broadcast value of lane k, add, compare, select) is left on the serial path.
            //   bv   running score of the target lane.  A gene end without a traceb connects to nothing
            //        (ref: impl/generic.h:29-36): while it has none its bv is -inf, so that as a source it loses every
            //        comparison without a test; what a target has to beat is therefore max(bv, 0)
            //   lk   in-batch source taken last (| (ov_mark + 1) << 8), -1 while the pre-walk result stands
            // Rows k >= kmax of the weight tile are NaN, so every batch runs the same 64 steps.
            const unsigned long long dynm = __ballot(s_dyn[slot][lane] != 0ull);
            const bool endlane = T.kind == 1 || T.kind == 2;
            // ndx of the pre-walk traceb node: in the previous batch (LDS), or found by the early far-field wave
            const int tbn_pre = B.tb < 0 ? -1 : (B.tb >= i0 - 64 ? s_fin.ndx[B.tb - (i0 - 64)] : s_etbn[slot][lane]);
            if (endlane && B.tb < 0) bv = -__builtin_huge_val();
            // thr = max(bv, 0); bv is never NaN, so the plain instruction is exact (fmax() would add a canonicalisation)
            auto floor0 = [](const double v) { double r; asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(v)); return r; };
            double thr = floor0(bv);
            int lk = -1;
            const double* wp = &s_w[slot][0][lane];
            const unsigned char* fp = &s_fl[slot][0][lane];
            auto step = [&](const int k, const double w) {
                double wk = w;
                if (__builtin_expect((int)((dynm >> k) & 1ull), 0)) {
                    // forward-stop source towards reverse targets: the admission depends on where the source's
                    // own traceb node lies
                    const int s_ndx = __builtin_amdgcn_readlane(T.ndx, k);
                    const int lkk = __builtin_amdgcn_readlane(lk, k);
                    const int tbnj = lkk >= 0 ? __builtin_amdgcn_readlane(T.ndx, lkk) : __builtin_amdgcn_readlane(tbn_pre, k);
                    int ov1;
                    dyn_weight(s_ndx, tbnj, fp[k * 64], T, D, wk, ov1);
                }
                const double val = readlane_f64(bv, k) + wk;
                const bool c = val >= thr;
                bv = c ? val : bv; lk = c ? k : lk;
                thr = floor0(bv);
            };
            double w0 = wp[0], w1 = wp[64], w2 = wp[128], w3 = wp[192];
#pragma unroll
            for (int k0 = 0; k0 < 64; k0 += 4) {
                const int kn = k0 + 4 < 64 ? k0 + 4 : 60;
                const double n0 = wp[kn * 64], n1 = wp[kn * 64 + 64], n2 = wp[kn * 64 + 128], n3 = wp[kn * 64 + 192];
                step(k0, w0); step(k0 + 1, w1); step(k0 + 2, w2); step(k0 + 3, w3);
                w0 = n0; w1 = n1; w2 = n2; w3 = n3;
            }
            // after the walk (every lane takes part in the shuffles): ndx of the traceb node, and the ov_mark of a
            // connection taken through a dynamic pair, re-derived from the source's final state
            const int lkc = lk & 63;
            const int ndx_lk = __shfl(T.ndx, lkc, 64);
            const int tbn_fin = lk >= 0 ? ndx_lk : tbn_pre;
            const int tbn_src = __shfl(tbn_fin, lkc, 64);
            if (lk >= 0) {
                B.val = bv; B.tb = i0 + lk; B.ov = -1; B.tbn = ndx_lk;
                if (D.is_r3 && ((dynm >> lk) & 1ull)) {
                    double wk = 0.0; int ov1;
                    dyn_weight(ndx_lk, tbn_src, fp[lk * 64], T, D, wk, ov1);
                    B.ov = ov1 - 1;
                }
            } else {
                B.tbn = tbn_pre;               // B.val stands: the walk changed nothing for this lane
                if (B.tb < 0) { B.tb = -1; B.ov = -1; B.tbn = -1; }
            }
            const unsigned long long tq2 = prof ? __builtin_readcyclecounter() : 0;
            s_fin.score[lane] = B.val; s_fin.ndx[lane] = T.ndx; s_fin.tbn[lane] = B.tb != -1 ? B.tbn : -1;
            s_fin.tb[lane] = B.tb; s_fin.ov[lane] = B.ov;
            s_fin.meta[lane] = T.meta; s_fin.cs[lane] = T.cs; s_fin.x0[lane] = T.x0; s_fin.x1[lane] = T.x1; s_fin.x2[lane] = T.x2;
            const unsigned long long dm = __ballot(endlane && B.tb == -1);
            if (lane == 0) s_fin.dead = dm;
            if (nx < n) T = Tnext;
            if (prof && lane == 0) {
                const unsigned long long tq3 = __builtin_readcyclecounter();
                buf.prof[0] += tq1 - tq0; buf.prof[1] += tq2 - tq1; buf.prof[2] += tq3 - tq2; buf.prof[5] += 1;
            }
            barrier_lds();
            if (prof && lane == 0) buf.prof[6] += __builtin_readcyclecounter() - tq0;
        }
    } else if (wave <= PGA_MW_HELPERS) {
        // ---------------------------------------------------------------- the weight helpers
        Target Tq;
        load_target(Tq, P, 0, lane, n, negc);
        helper_weights(Tq, Tq, 0, 0, false);
        __syncthreads();
        for (int b = 0; b < nb; b++) {
            const int i0 = b << 6, slot = b & 1, nx = i0 + 64, pb = slot ^ 1;
            Target Tn;
            if (nx < n) load_target(Tn, P, nx, lane, n, negc);     // needed after the barrier
            const unsigned long long th0 = prof ? __builtin_readcyclecounter() : 0;
            if (wave < PGA_MW_SLICES) {
                const DynRegs D = dyn_regs(Tq);
                consume(Tq, D, i0);
            }
            if (prof && lane == 0 && wave == 1) buf.prof[15] += __builtin_readcyclecounter() - th0;
            barrier_lds();
            const unsigned long long tq1 = prof ? __builtin_readcyclecounter() : 0;
            if (nx < n) {
                helper_weights(Tq, Tn, nx, pb, true);
                Tq = Tn;
                if (prof && lane == 0 && wave == 1) buf.prof[3] += __builtin_readcyclecounter() - tq1;
            }
            barrier_lds();
        }
    } else {
        // ---------------------------------------------------------------- the far-field waves, one per target kind
        const int mykind = wave - (PGA_MW_HELPERS + 1);
        if (mykind == 0) { s_eval[0][lane] = 0.0; s_etb[0][lane] = -1; s_eov[0][lane] = -1; s_etbn[0][lane] = -1; }
        __syncthreads();
        for (int b = 0; b < nb; b++) {
            const int i0 = b << 6, slot = b & 1, nx = i0 + 64, pb = slot ^ 1;
            // the batch finalized last is stored by these waves, idle in this phase (a part each); the far fields
            // that read it start after the barrier
            if (i0 > 0) {
                const unsigned long long tf0 = prof ? __builtin_readcyclecounter() : 0;
                flush(i0 - 64, mykind);
                if (prof && lane == 0 && mykind == 1) buf.prof[12] += __builtin_readcyclecounter() - tf0;
            }
            __syncthreads();
            const unsigned long long tq1 = prof ? __builtin_readcyclecounter() : 0;
            if (nx < n) {
                Target Tn;
                load_target(Tn, P, nx, lane, n, negc);      // in flight during the suffix scan
                // a single wave would run the four kinds' paths one after the other
                // forward starts and reverse stops look for far gene ends: their ranges nearly always end at i0
                SuffixLds* sfx = mykind == 0 ? &s_sfx[0] : (mykind == 3 ? &s_sfx[1] : nullptr);
                const int eend = i0 >> 3, ebase = max(0, eend - PGA_RING_BLOCKS);
                if (sfx != nullptr && i0 > 0) suffix_build(&s_ring, sfx, ebase, eend, lane);
                if (prof && lane == 0 && mykind == 0) buf.prof[13] += __builtin_readcyclecounter() - tq1;
                if (Tn.kind == mykind) {
                    Best B{0.0, -1, -1, -1};
                    if (i0 > 0) far_field(Tn, 0, i0, P, s_levbase, negc, s_igm, B, &s_ring, sfx, ebase, eend);   // every tile finalized before this iteration
                    s_eval[pb][lane] = B.val; s_etb[pb][lane] = B.tb; s_eov[pb][lane] = B.ov;
                    // ndx of the traceb node: from the ring when it is recent enough (nearly always), else from memory
                    int tbn = -1;
                    if (B.tb >= 0) tbn = B.tb >= i0 - PGA_RING ? s_ring.ndx[B.tb & (PGA_RING - 1)] : P.src[B.tb].ndx;
                    s_etbn[pb][lane] = tbn;
                }
                if (prof && lane == 0) buf.prof[8 + mykind] += __builtin_readcyclecounter() - tq1;
            }
            barrier_lds();
        }
        flush((nb - 1) << 6, mykind);
        if (mykind == 0) publish_max(end_best, end_idx, end_tb, lane, buf, slot);
    }
}

// W wavefronts per chain.  Per 64-target batch:
//   phase F  every wave scans its share of the already-final sources (tiles of 64, coalesced loads,
//            ballot-built visit mask) against the same 64 targets;
//   merge    the W partial (value, index) maxima meet in LDS, wave 0 folds them;
//   phase I  wave 0 walks the 63 in-batch sources: lane k is final once the walk reaches source
//            i0+k, because by then it has met every j < i0+k;
//   wave 0 stores the batch (score, traceb, ov_mark, ndx-of-traceb) and the next batch starts.
template <int W, bool FINAL = true>
__global__ void __launch_bounds__(64 * W)
k_dp_chain(const ChainDesc* __restrict__ chains, const DpSrc* __restrict__ g_src, const DpTgt* __restrict__ g_tgt,
           const ModelConst* __restrict__ models, double* g_score, int32_t* g_traceb, int32_t* g_tbn, int8_t* g_ov,
           int32_t* __restrict__ max_index, double* __restrict__ max_score, int32_t* __restrict__ ipath) {
    __shared__ double s_igm[64];
    __shared__ double s_pval[W > 1 ? W : 1][64];
    __shared__ int s_ptb[W > 1 ? W : 1][64], s_pov[W > 1 ? W : 1][64], s_ptbn[W > 1 ? W : 1][64];
    const ChainDesc cd = chains[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = cd.n;
    const ModelConst* mc = &models[cd.model];
    if (wave == 0) s_igm[lane] = mc->igm[lane];
    __syncthreads();
    const double negc = mc->negc, st_wt = mc->st_wt;
    const DpSrc* __restrict__ src = g_src + cd.off;
    const DpTgt* __restrict__ tgt = g_tgt + cd.off;
    double* score = g_score + cd.off; int32_t* traceb = g_traceb + cd.off;
    int32_t* tbn = g_tbn + cd.off; int8_t* ovm = g_ov + cd.off;

    double end_best = -1.0; int end_idx = -1, end_tb = -1;     // _find_max_index (ref: lib.pyx:1239-1251)

    for (int i0 = 0; i0 < n; i0 += 64) {
        Target T;
        T.i = i0 + lane;
        const bool act = T.i < n;
        DpSrc me;
        {
            const int ii = act ? T.i : n - 1;
            me = src[ii]; const DpTgt mt = tgt[ii];
            T.kind = PGA_KIND(me.meta); T.frame = PGA_FRAME(me.meta); T.meta = me.meta;
            T.ndx = me.ndx; T.stop_val = me.stop_val; T.cs = me.cs; T.csd = me.cs + negc;
            T.x0 = me.x[0]; T.x1 = me.x[1]; T.x2 = me.x[2];
            T.n3n0 = mt.n3ndx[0]; T.n3n1 = mt.n3ndx[1]; T.n3n2 = mt.n3ndx[2];
            T.n3s0 = mt.n3stop[0]; T.n3s1 = mt.n3stop[1]; T.n3s2 = mt.n3stop[2];
            T.lo = act ? mt.lo : INT_MAX;
            T.a0 = me.n3src[0]; T.a1 = me.n3src[1]; T.a2 = me.n3src[2];     // carried for the in-batch walk of the training pass
            if (!act) T.i = -1;       // j < T.i is never true: lane stays idle
        }
        Best B{0.0, -1, -1, -1};
        const int wlo = wave_min_i32(T.lo);

        // ---- phase F: sources of earlier batches, tiles dealt round-robin to the W waves
        for (int t0 = (wlo & ~63) + 64 * wave; t0 < i0; t0 += 64 * W) {
            const int sidx = t0 + lane;                       // < i0 <= n
            SrcLane S;
            {
                const DpSrc r = src[sidx];
                S.ndx = r.ndx; S.stop_val = r.stop_val; S.meta = r.meta; S.cs = r.cs; S.x0 = r.x[0]; S.x1 = r.x[1]; S.x2 = r.x[2];
                S.n3a = r.n3src[0]; S.n3b = r.n3src[1]; S.n3c = r.n3src[2];
                S.score = score[sidx]; S.tbn = tbn[sidx];
            }
            const int sk = PGA_KIND(S.meta);
            const bool dead = (sk == 1 || sk == 2) && S.tbn == -1;   // gene end never reached: connects to nothing (ref: :110-114)
            unsigned long long visit = __ballot(sidx >= wlo && !dead);
            while (visit) {
                const int k = __builtin_ctzll(visit);
                visit &= visit - 1;
                visit_source<FINAL>(k, t0 + k, S, T, negc, s_igm, B, st_wt);
            }
        }
        if (W > 1) {
            s_pval[wave][lane] = B.val; s_ptb[wave][lane] = B.tb; s_pov[wave][lane] = B.ov; s_ptbn[wave][lane] = B.tbn;
            __syncthreads();
        }
        if (wave == 0) {
            if (W > 1) {
#pragma unroll 4
                for (int w = 1; w < W; w++) {
                    const double v = s_pval[w][lane]; const int t = s_ptb[w][lane];
                    if (t != -1 && (v > B.val || (v == B.val && t > B.tb))) { B.val = v; B.tb = t; B.ov = s_pov[w][lane]; B.tbn = s_ptbn[w][lane]; }
                }
            }
            // ---- phase I: sources inside this batch are the lanes themselves
            const int kmax = min(63, n - 1 - i0);
            for (int k = 0; k < kmax; k++) {
                const int sk = PGA_KIND(__builtin_amdgcn_readlane(T.meta, k));
                const int tbk = __builtin_amdgcn_readlane(B.tb, k);
                if ((sk == 1 || sk == 2) && tbk == -1) continue;
                SrcLane S;
                S.ndx = T.ndx; S.stop_val = T.stop_val; S.meta = T.meta; S.tbn = B.tbn;
                S.cs = T.cs; S.x0 = T.x0; S.x1 = T.x1; S.x2 = T.x2; S.score = B.val;
                S.n3a = T.a0; S.n3b = T.a1; S.n3c = T.a2;
                visit_source<FINAL>(k, i0 + k, S, T, negc, s_igm, B, st_wt);
            }
            if (act) {
                score[T.i] = B.val; traceb[T.i] = B.tb; ovm[T.i] = (int8_t)B.ov; tbn[T.i] = B.tb < 0 ? -1 : B.tbn;
                if ((T.kind == 1 || T.kind == 2) && B.val >= end_best) { end_best = B.val; end_idx = T.i; end_tb = B.tb; }
            }
        }
        if (W > 1) __syncthreads();      // the batch's stores are visible to the whole workgroup from here on
    }
    if (wave != 0) return;
    // highest score among gene-end nodes, ties to the largest index (the reference scans from the end with '>')
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ob = __shfl_xor(end_best, m, 64);
        const int oi = __shfl_xor(end_idx, m, 64);
        const int ot = __shfl_xor(end_tb, m, 64);
        if (ob > end_best || (ob == end_best && oi > end_idx)) { end_best = ob; end_idx = oi; end_tb = ot; }
    }
    if (lane == 0) {
        max_index[blockIdx.x] = end_idx; max_score[blockIdx.x] = end_idx >= 0 ? end_best : 0.0;
        ipath[blockIdx.x] = (end_idx >= 0 && end_tb != -1) ? end_idx : -1;
    }
}


// ---------------------------------------------------------------------------------------------
// Segmented chains: speculate, re-score exactly, verify.
//
// The walk of a chain is serial, one node after the other, so a launch with few long chains (one genome, or
// one genome x 16 metagenomic models) leaves the chip idle.  The choice of a node's predecessor only depends on
// score DIFFERENCES inside its window, and those settle a few genes after any starting point, so:
//   1  speculate  every segment [s, e) of a long chain is walked as an independent sub-chain that starts
//                 `warm` nodes early from the empty state (the unchanged chain kernel, one workgroup per segment);
//                 only the traceb of [s, e) is kept -- the CLAIM;
//   2  re-score   the exact scores that belong to the claimed tracebs: score[i] = score[tb[i]] + w(tb[i], i), the
//                 additions in chain order as the serial walk makes them (one wave per chain, no search);
//   3  verify     with exact scores of ALL earlier nodes in memory, every node re-evaluates its whole window in
//                 parallel (the far-field routine of the one-wave kernel over [lo, i)) and compares (score, traceb,
//                 ov_mark) with the claim.  If every node of a chain agrees the claim IS the serial result, by
//                 induction over the node index: node i's claim equals the recurrence applied to the exact values
//                 of the nodes before it.
// A mismatch (an unsettled warm-up, or a tie decided by the last bit of a score) makes the verified choice the
// next claim and steps 2-3 run again; a chain that still disagrees after PGA_SEG_ROUNDS rounds is walked
// serially.  Nothing here is approximate: a result is only ever published after a verification that found no
// mismatch, or by the serial kernel.
__global__ void __launch_bounds__(256)
k_seg_gather(const DpSeg* __restrict__ segs, const ChainDesc* __restrict__ chains, const int32_t* __restrict__ traceb, int32_t* __restrict__ ctb) {
    const DpSeg sd = segs[blockIdx.y];
    const int li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= sd.e - sd.s) return;
    const int node = sd.s + li;
    const int tb = traceb[sd.off + (node - sd.a)];
    ctb[chains[sd.chain].off + node] = tb >= 0 ? tb + sd.a : -1;
}

// the claim of a round: traceb, the ov_mark and ndx-of-traceb that go with it, and the weight of the connection
__global__ void __launch_bounds__(256)
k_seg_weights(const ChainDesc* __restrict__ chains, const int32_t* __restrict__ big, const int32_t* __restrict__ gate,
              const DpSrc* __restrict__ g_src, const DpTgt* __restrict__ g_tgt, const ModelConst* __restrict__ models, DpBuffers buf,
              const int32_t* __restrict__ g_ctb, double* __restrict__ g_cw, uint32_t* __restrict__ g_hb) {
    __shared__ double s_igm[64];
    const int chain = big[blockIdx.y];
    if (gate != nullptr && gate[chain] == 0) return;
    const ChainDesc cd = chains[chain];
    if ((int)(blockIdx.x * blockDim.x) >= cd.n) return;
    const ModelConst* mc = &models[cd.model];
    if (threadIdx.x < 64) s_igm[threadIdx.x] = mc->igm[threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cd.n) return;
    const double negc = mc->negc;
    const ChainPtrs P = chain_ptrs(cd, g_src, g_tgt, buf);
    const int32_t* ctb = g_ctb + cd.off;
    int tb = ctb[i];
    if (tb >= i || tb < 0) tb = -1;
    double w = 0.0; int mf = -1, tbn = -1;
    if (tb >= 0) {
        Target T;
        load_target(T, P, i, 0, cd.n, negc);
        int t2 = ctb[tb];
        if (t2 >= tb || t2 < 0) t2 = -1;
        const int tbnj = t2 >= 0 ? P.src[t2].ndx : -1;
        const GlobalAcc G{P.src, nullptr, nullptr};
        bool ok;
        pair_weight(tb, G, tbnj, T, negc, s_igm, ok, w, mf);
        if (ok) tbn = P.src[tb].ndx; else { tb = -1; w = 0.0; mf = -1; }
    }
    P.traceb[i] = tb; P.ovm[i] = (int8_t)mf; P.tbn[i] = tbn; g_cw[cd.off + i] = w;
    if (tb >= 0 && !(g_hb[cd.off + tb] & 2u)) atomicOr(&g_hb[cd.off + tb], 2u);     // height >= 1: some node continues from tb
}

// Height classes of the claimed traceb forest (a node's height = the longest chain of nodes continuing from it):
// bit k of hb[i] = "height >= k".  Pass k reads bit k-1 (final since the pass before) and sets bit k of the parent.
__global__ void __launch_bounds__(256)
k_seg_height(const ChainDesc* __restrict__ chains, const int32_t* __restrict__ big, const int32_t* __restrict__ gate,
             const int32_t* __restrict__ g_traceb, uint32_t* __restrict__ g_hb, const int k) {
    const int chain = big[blockIdx.y];
    if (gate != nullptr && gate[chain] == 0) return;
    const ChainDesc cd = chains[chain];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cd.n) return;
    const int tb = g_traceb[cd.off + i];
    if (tb < 0 || !(g_hb[cd.off + i] & (1u << (k - 1)))) return;
    if (!(g_hb[cd.off + tb] & (1u << k))) atomicOr(&g_hb[cd.off + tb], 1u << k);
}

// scores of the nodes of height class c (exactly): their parents are higher and already final
__global__ void __launch_bounds__(256)
k_seg_leaves(const ChainDesc* __restrict__ chains, const int32_t* __restrict__ big, const int32_t* __restrict__ gate,
             const int32_t* __restrict__ first_bad, DpBuffers buf, const double* __restrict__ g_cw, const uint32_t* __restrict__ g_hb, const int c) {
    const int chain = big[blockIdx.y];
    if (gate != nullptr && gate[chain] == 0) return;
    const ChainDesc cd = chains[chain];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cd.n) return;
    if (first_bad != nullptr && i < (first_bad[chain] & ~63)) return;
    const uint32_t hb = g_hb[cd.off + i] | 1u;
    if (31 - __builtin_clz(hb) != c) return;
    const int tb = buf.traceb[cd.off + i];
    buf.score[cd.off + i] = tb < 0 ? 0.0 : buf.score[cd.off + tb] + g_cw[cd.off + i];
}

// Exact scores of the claimed tracebs: score[i] = score[tb[i]] + w(tb[i], i), every addition as the serial walk
// makes it.  Only the dependence parent -> child is left, and only long chains of it need a serial pass: the
// SPINE = nodes of height >= PGA_SEG_HEIGHT (the path and what keeps up with it for a while; about 3% of the
// nodes of a genome, each one's parent nearly always the spine node just before it).  The spine is compacted into
// a list in index order (k_spine_count / _scan / _fill), one wave per chain walks the list 64 entries at a time
// (parents in earlier batches come from an LDS ring, parents in the same batch are met one after the other through
// v_readlane), and every other node is scored afterwards, one parallel pass per height class (k_seg_leaves).
// A later round restarts at the first node the verification rejected: everything before it is already exact.
#define PGA_SEG_HEIGHT 4          // default; PGA_DP_SEG_HEIGHT = 2 .. 24 (seg_height()): a higher bar means a shorter spine and more leaf passes
                                  // (measured on configs 2 and 5: 8 changes nothing, 12 and 16 are 3-8 % slower -- the spine is mostly the path itself)
static int seg_height() {
    static int h = 0;
    if (!h) { const char* e = getenv("PGA_DP_SEG_HEIGHT"); h = e ? atoi(e) : PGA_SEG_HEIGHT; if (h < 2 || h > 24) h = PGA_SEG_HEIGHT; }
    return h;
}
#define PGA_RS_RING 4096

__device__ __forceinline__ int64_t seg_tile(const ChainDesc& cd, const int chain, const int i) { return (cd.off >> 6) + chain + (i >> 6); }

__global__ void __launch_bounds__(256)
k_spine_count(const ChainDesc* __restrict__ chains, const int32_t* __restrict__ big, const int32_t* __restrict__ gate,
              const uint32_t* __restrict__ g_hb, unsigned long long* __restrict__ g_tmask, const int spine_h) {
    const int chain = big[blockIdx.y];
    if (gate != nullptr && gate[chain] == 0) return;
    const ChainDesc cd = chains[chain];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i - (int)(threadIdx.x & 63) >= cd.n) return;
    const bool spine = i < cd.n && ((g_hb[cd.off + i] >> spine_h) & 1u);
    const unsigned long long m = __ballot(spine);
    if ((threadIdx.x & 63) == 0) g_tmask[seg_tile(cd, chain, i)] = m;
}

// exclusive scan of the tiles' spine counts; one workgroup per chain
__global__ void __launch_bounds__(1024)
k_spine_scan(const ChainDesc* __restrict__ chains, const int32_t* __restrict__ big, const int32_t* __restrict__ gate,
             const unsigned long long* __restrict__ g_tmask, int32_t* __restrict__ g_toff, int32_t* __restrict__ g_nsp) {
    __shared__ int s_wsum[16];
    __shared__ int s_base;
    const int chain = big[blockIdx.x];
    if (gate != nullptr && gate[chain] == 0) return;
    const ChainDesc cd = chains[chain];
    const int nt = (cd.n + 63) >> 6, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t tq0 = seg_tile(cd, chain, 0);
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int q0 = 0; q0 < nt; q0 += 1024) {
        const int q = q0 + t;
        const int c = q < nt ? __popcll(g_tmask[tq0 + q]) : 0;
        int inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < wave; k++) woff += s_wsum[k];
        const int base = s_base;
        if (q < nt) g_toff[tq0 + q] = base + woff + inc - c;
        __syncthreads();
        if (t == 1023) s_base = base + woff + inc;
        __syncthreads();
    }
    if (t == 0) g_nsp[chain] = s_base;
}

// the list: chain index, claimed traceb, list position of the traceb node, weight of the connection
__global__ void __launch_bounds__(256)
k_spine_fill(const ChainDesc* __restrict__ chains, const int32_t* __restrict__ big, const int32_t* __restrict__ gate,
             const int32_t* __restrict__ g_traceb, const double* __restrict__ g_cw, const unsigned long long* __restrict__ g_tmask,
             const int32_t* __restrict__ g_toff, int32_t* __restrict__ sp_idx, int32_t* __restrict__ sp_tb, int32_t* __restrict__ sp_pp,
             double* __restrict__ sp_w) {
    const int chain = big[blockIdx.y];
    if (gate != nullptr && gate[chain] == 0) return;
    const ChainDesc cd = chains[chain];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cd.n) return;
    const int64_t tq = seg_tile(cd, chain, i);
    const unsigned long long m = g_tmask[tq];
    if (!((m >> (i & 63)) & 1ull)) return;
    const int64_t pos = cd.off + g_toff[tq] + __popcll(m & ((1ull << (i & 63)) - 1ull));
    const int tb = g_traceb[cd.off + i];
    int pp = -1;
    if (tb >= 0) {
        const int64_t tp = seg_tile(cd, chain, tb);
        pp = g_toff[tp] + __popcll(g_tmask[tp] & ((1ull << (tb & 63)) - 1ull));     // the parent of a spine node is a spine node
    }
    sp_idx[pos] = i; sp_tb[pos] = tb; sp_pp[pos] = pp; sp_w[pos] = g_cw[cd.off + i];
}

__global__ void __launch_bounds__(64)
k_dp_rescore(const ChainDesc* __restrict__ chains, const int32_t* __restrict__ big, const int32_t* __restrict__ gate,
             const int32_t* __restrict__ first_bad, DpBuffers buf, const int32_t* __restrict__ g_toff, const int32_t* __restrict__ g_nsp,
             const int32_t* __restrict__ sp_idx, const int32_t* __restrict__ sp_tb, const int32_t* __restrict__ sp_pp,
             const double* __restrict__ sp_w) {
    __shared__ double s_ring[PGA_RS_RING];
    __shared__ double s_out[64], s_term[64];
    __shared__ int s_code[64];
    const int chain = big[blockIdx.x];
    if (gate != nullptr && gate[chain] == 0) return;
    const ChainDesc cd = chains[chain];
    const int lane = threadIdx.x;
    const int m = g_nsp[chain];
    double* score = buf.score + cd.off;
    const int32_t* __restrict__ lidx = sp_idx + cd.off; const int32_t* __restrict__ ltb = sp_tb + cd.off;
    const int32_t* __restrict__ lpp = sp_pp + cd.off; const double* __restrict__ lw = sp_w + cd.off;
    int p0 = 0;
    if (first_bad != nullptr) {
        const int b0 = min(max(first_bad[chain], 0), cd.n) & ~63;
        p0 = b0 >= cd.n ? m : g_toff[seg_tile(cd, chain, b0)];
    }
    if (m <= 0) return;
    struct Ent { int ix, tb, pp; double w; };
    // branch-free (clamped index, then a select): with the loads under a lane mask the compiler waits for every load in
    // flight at each join, which would serialise a memory round trip per batch
    auto load = [&](const int ps) {
        const int e = ps + lane;
        const bool in = e < m;
        const int ec = min(e, m - 1);
        Ent r;
        const int a = lidx[ec], b = ltb[ec], c = lpp[ec]; const double d = lw[ec];
        r.ix = in ? a : -1; r.tb = in ? b : -1; r.pp = in ? c : -1; r.w = in ? d : 0.0;
        return r;
    };
    Ent nx1 = load(p0), nx2 = load(p0 + 64);
    for (int ps = p0; ps < m; ps += 64) {
        const Ent cur = nx1;
        nx1 = nx2; nx2 = load(ps + 128);                 // two batches ahead: in flight during this one
        const int e = ps + lane;
        const bool valid = e < m;
        const int pp = cur.pp; const double w = cur.w;
        // entries whose parent lies in an earlier batch (or that have none) are final at once
        const bool inb = valid && pp >= ps;
        const bool early = valid && pp >= 0 && pp < ps;
        const bool near = pp >= p0 && e - pp <= PGA_RS_RING;
        double sj = s_ring[pp & (PGA_RS_RING - 1)];
        if (__any(early && !near))                        // far back, or left by the round before: rare
            if (early && !near) sj = __hip_atomic_load(&score[cur.tb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double s = early ? sj + w : 0.0;
        if (__any(inb)) {
            // (tried: every lane from t up adds entry t's term, fetched by v_readlane, so that nothing goes through LDS -- exact, but
            // 20-30 % slower: the exec write and the SGPR hand-over per step cost more than the LDS traffic of this form)
            // the others in list order, by ONE lane (a lone wavefront issues an instruction every four cycles whatever
            // its width, so what counts is the instruction count per entry: here a scalar test, the addition, and half an LDS
            // read and write).  The parent is nearly always the entry just before: r carries the chain.  `special` entries --
            // final already (the chain restarts from their value), or with another lane of the batch as parent -- take the
            // rare branch.  Every r is left in LDS, where each lane picks its own afterwards.
            const int code = inb ? pp - ps : -1;
            const unsigned long long special = __ballot(!inb || code != lane - 1);
            s_term[lane] = inb ? w : s; s_code[lane] = code;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
            if (lane == 0) {
                double xs[64];                      // all terms first: the chain then never waits for LDS
#pragma unroll
                for (int t = 0; t < 64; t++) xs[t] = s_term[t];
                double r = 0.0;
                // Eight entries at a time.  A group without a special entry (nine in ten: 1.5 % of the entries are special) is eight
                // additions back to back, each waiting only for the one before, and their results leave for LDS behind them; with a
                // test, a branch and a store between any two additions (the form kept for the other groups) an entry cost some 55
                // cycles of a lone wavefront -- 5.7 ms for the 224 000 spine entries of config 5.
#pragma unroll
                for (int t0 = 0; t0 < 64; t0 += 8) {
                    const unsigned sp8 = (unsigned)(special >> t0) & 0xffu;
                    if (__builtin_expect(sp8 == 0u, 1)) {
                        const double r0 = r + xs[t0], r1 = r0 + xs[t0 + 1], r2 = r1 + xs[t0 + 2], r3 = r2 + xs[t0 + 3];
                        const double r4 = r3 + xs[t0 + 4], r5 = r4 + xs[t0 + 5], r6 = r5 + xs[t0 + 6], r7 = r6 + xs[t0 + 7];
                        s_out[t0] = r0; s_out[t0 + 1] = r1; s_out[t0 + 2] = r2; s_out[t0 + 3] = r3;
                        s_out[t0 + 4] = r4; s_out[t0 + 5] = r5; s_out[t0 + 6] = r6; s_out[t0 + 7] = r7;
                        r = r7;
                    } else {
#pragma unroll
                        for (int t = t0; t < t0 + 8; t++) {
                            const double x = xs[t];
                            if (__builtin_expect((int)((special >> t) & 1ull), 0)) {
                                const int c = s_code[t];
                                r = c < 0 ? x : s_out[c] + x;
                            } else r = r + x;
                            s_out[t] = r;
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
            s = s_out[lane];
        }
        if (valid) { score[cur.ix] = s; s_ring[e & (PGA_RS_RING - 1)] = s; }
        // the ring is read by other lanes of this wave in a later batch: LDS operations of a wave execute in order, so only
        // the compiler has to keep them in place (a memory fence here would also wait for the loads in flight)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    }
}

__device__ __forceinline__ int levbase_of(const int lev, const int n) {     // init_levbase, one entry
    int base = 0;
    for (int l = 1; l < lev; l++) base += (l * 3 < 31) ? (n >> (3 * l)) : 0;
    return base;
}

// far-field candidate values of every node (finalize_batch's A and V) and the two lowest tree levels, from the
// exact scores; one wave per 64 nodes
__global__ void __launch_bounds__(256)
k_seg_build_far(const ChainDesc* __restrict__ chains, const int32_t* __restrict__ big, const int32_t* __restrict__ gate,
                const DpSrc* __restrict__ g_src, const DpTgt* __restrict__ g_tgt, const ModelConst* __restrict__ models, DpBuffers buf,
                double* __restrict__ g_tv, int32_t* __restrict__ g_ti) {
    const int chain = big[blockIdx.y];
    if (gate != nullptr && gate[chain] == 0) return;
    const ChainDesc cd = chains[chain];
    const int n = cd.n;
    const int i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const int i0 = i - lane;
    if (i0 >= n) return;
    const double NEG_INF = -__builtin_huge_val();
    const double negc = models[cd.model].negc;
    const ChainPtrs P = chain_ptrs(cd, g_src, g_tgt, buf);
    double a_val = NEG_INF;
    double ev = -1.0; int ei = -1;              // _find_max_index over this tile: gene ends, ties to the larger index
    if (i < n) {
        const DpSrc me = P.src[i];
        const int kind = PGA_KIND(me.meta), frame = PGA_FRAME(me.meta);
        const double val = P.score[i];
        if (kind == 1 || kind == 2) { ev = val; ei = i; }
        const bool alive = P.traceb[i] != -1;
        double v0 = NEG_INF, v1 = NEG_INF, v2 = NEG_INF;
        if (kind == 0) {
            const double g = val + me.cs;
            if (frame == 0) v0 = g; else if (frame == 1) v1 = g; else v2 = g;
        } else if (kind == 1 && alive) {
            a_val = val + negc;
            if (PGA_SPVALID(me.meta, 0)) v0 = val + me.x[0];
            if (PGA_SPVALID(me.meta, 1)) v1 = val + me.x[1];
            if (PGA_SPVALID(me.meta, 2)) v2 = val + me.x[2];
        } else if (kind == 2 && alive) {
            a_val = val + negc;
        }
        P.A[i] = a_val; P.V0[i] = v0; P.V1[i] = v1; P.V2[i] = v2;
    }
    double rv = a_val; int ri = i;
#pragma unroll
    for (int m = 1; m <= 4; m <<= 1) {
        const double ov2 = __shfl_xor(rv, m, 64); const int oi = __shfl_xor(ri, m, 64);
        if (ov2 > rv || (ov2 == rv && oi > ri)) { rv = ov2; ri = oi; }
    }
    if ((lane & 7) == 0 && i + 8 <= n) { P.hv[levbase_of(1, n) + (i >> 3)] = rv; P.hi[levbase_of(1, n) + (i >> 3)] = ri; }
#pragma unroll
    for (int m = 8; m <= 32; m <<= 1) {
        const double ov2 = __shfl_xor(rv, m, 64); const int oi = __shfl_xor(ri, m, 64);
        if (ov2 > rv || (ov2 == rv && oi > ri)) { rv = ov2; ri = oi; }
    }
    if (lane == 0 && i0 + 64 <= n) { P.hv[levbase_of(2, n) + (i0 >> 6)] = rv; P.hi[levbase_of(2, n) + (i0 >> 6)] = ri; }
#pragma unroll
    for (int m = 1; m <= 32; m <<= 1) {
        const double ov2 = __shfl_xor(ev, m, 64); const int oi = __shfl_xor(ei, m, 64);
        if (ov2 > ev || (ov2 == ev && oi > ei)) { ev = ov2; ei = oi; }
    }
    if (lane == 0) { const int64_t tq = (cd.off >> 6) + chain + (i0 >> 6); g_tv[tq] = ev; g_ti[tq] = ei; }
}

// the tree levels above that, and _find_max_index (ref: lib.pyx:1239-1251); one workgroup per chain
__global__ void __launch_bounds__(256)
k_seg_build_upper(const ChainDesc* __restrict__ chains, const int32_t* __restrict__ big, const int32_t* __restrict__ gate,
                  const DpSrc* __restrict__ g_src, const DpTgt* __restrict__ g_tgt, DpBuffers buf,
                  const double* __restrict__ g_tv, const int32_t* __restrict__ g_ti) {
    __shared__ double s_v[256];
    __shared__ int s_i[256];
    const int chain = big[blockIdx.x];
    if (gate != nullptr && gate[chain] == 0) return;
    const ChainDesc cd = chains[chain];
    const int n = cd.n, t = threadIdx.x;
    const ChainPtrs P = chain_ptrs(cd, g_src, g_tgt, buf);
    for (int lev = 3; lev * 3 < 31 && (n >> (3 * lev)) > 0; lev++) {
        const int m = n >> (3 * lev), cb = levbase_of(lev - 1, n), ob = levbase_of(lev, n);
        for (int e = t; e < m; e += 256) {
            double bv = P.hv[cb + 8 * e]; int bi = P.hi[cb + 8 * e];
            for (int q = 1; q < 8; q++) {
                const double v = P.hv[cb + 8 * e + q]; const int ix = P.hi[cb + 8 * e + q];
                if (v > bv || (v == bv && ix > bi)) { bv = v; bi = ix; }
            }
            P.hv[ob + e] = bv; P.hi[ob + e] = bi;
        }
        __threadfence_block();
        __syncthreads();
    }
    double eb = -1.0; int ei = -1;
    const int64_t tq0 = (cd.off >> 6) + chain;
    for (int q = t; q < ((n + 63) >> 6); q += 256) {
        const double v = g_tv[tq0 + q]; const int ix = g_ti[tq0 + q];
        if (v > eb || (v == eb && ix > ei)) { eb = v; ei = ix; }
    }
    s_v[t] = eb; s_i[t] = ei;
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {
        if (t < h) {
            const double v = s_v[t + h]; const int ix = s_i[t + h];
            if (v > s_v[t] || (v == s_v[t] && ix > s_i[t])) { s_v[t] = v; s_i[t] = ix; }
        }
        __syncthreads();
    }
    if (t == 0) {
        const int end_idx = s_i[0];
        const int end_tb = end_idx >= 0 ? P.traceb[end_idx] : -1;
        buf.max_index[chain] = end_idx; buf.max_score[chain] = end_idx >= 0 ? s_v[0] : 0.0;
        buf.ipath[chain] = (end_idx >= 0 && end_tb != -1) ? end_idx : -1;
    }
}

// every node against its whole window, all earlier nodes taken as final
__global__ void __launch_bounds__(256)
k_dp_verify(const ChainDesc* __restrict__ chains, const int32_t* __restrict__ big, const int32_t* __restrict__ gate,
            const DpSrc* __restrict__ g_src, const DpTgt* __restrict__ g_tgt, const ModelConst* __restrict__ models, DpBuffers buf,
            int32_t* __restrict__ g_ctb, int32_t* __restrict__ flags, int32_t* __restrict__ first_bad, const int32_t* __restrict__ from) {
    __shared__ double s_igm[64];
    __shared__ int s_levbase[12];
    const int chain = big[blockIdx.y];
    if (gate != nullptr && gate[chain] == 0) return;
    const ChainDesc cd = chains[chain];
    const int n = cd.n;
    if ((int)(blockIdx.x * blockDim.x) >= n) return;
    const ModelConst* mc = &models[cd.model];
    if (threadIdx.x < 64) s_igm[threadIdx.x] = mc->igm[threadIdx.x];
    if (threadIdx.x == 64) init_levbase(s_levbase, n);
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // a later round: everything before the first node the round before rejected was verified then and has not changed
    if (from != nullptr && i < (from[chain] & ~63)) return;
    const double negc = mc->negc;
    const ChainPtrs P = chain_ptrs(cd, g_src, g_tgt, buf);
    Target T;
    load_target(T, P, i, 0, n, negc);
    Best B{0.0, -1, -1, -1};
    far_field(T, 0, n, P, s_levbase, negc, s_igm, B);
    if (B.tb < 0) { B.tb = -1; B.ov = -1; }
    g_ctb[cd.off + i] = B.tb;
    if (!(B.val == P.score[i]) || B.tb != P.traceb[i] || (int8_t)B.ov != P.ovm[i]) { atomicAdd(&flags[chain], 1); atomicMin(&first_bad[chain], i); }
}

}  // namespace

void pga_launch_dp_prepare(const ChainDesc* d_chains, int n_chains, int64_t node_begin, int64_t total_nodes,
                           const NodeArrays& nodes, const ModelConst* d_models, DpBuffers buf, hipStream_t st, int final) {
    if (total_nodes <= 0) return;
    const int threads = 256;
    const int64_t blocks = (total_nodes + threads - 1) / threads;
    hipLaunchKernelGGL(k_dp_prepare, dim3((unsigned)blocks), dim3(threads), 0, st,
                       d_chains, n_chains, node_begin, total_nodes, nodes, d_models, buf.src, buf.tgt, final);
}

bool pga_dp_use_wave(int n_chains) {
    const char* kern = getenv("PGA_DP_KERNEL");
    if (kern && *kern) return strcmp(kern, "wave") == 0;
    return n_chains >= 2048;
}
static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

bool pga_dp_plan(const ChainDesc* h, int n_chains, int64_t tot_nodes, DpSegPlan& plan, const bool wave_walk) {
    plan = DpSegPlan();
    if (n_chains <= 0 || n_chains >= 2048 || getenv("PGA_DP_KERNEL") || env_int("PGA_DP_SEG", 1) == 0) return false;
    const int min_chain = std::max(256, env_int("PGA_DP_SEG_MIN", 16384));
    const int warm = std::max(64, env_int("PGA_DP_SEG_WARM", 4096));      // 2048 left 8 (config 5) / 938 (config 2) nodes for a second round; 4096 none
    int64_t cand = 0;
    for (int c = 0; c < n_chains; c++) if (h[c].n >= min_chain) cand += h[c].n;
    if (cand == 0) return false;
    // one workgroup per compute unit is what the chain kernel's LDS footprint allows: every sub-chain and every chain that
    // is not cut must be resident at once (256 CUs), a workgroup too many would wait for a free CU and double the time
    int n_short = 0;                                      // chains that are walked whole whatever the segment length
    for (int c = 0; c < n_chains; c++) n_short += h[c].n < min_chain;
    auto count_segs = [&](const int64_t len) {            // workgroups of the candidates: segments, or 1 when left whole
        int64_t k = 0;
        for (int c = 0; c < n_chains; c++) {
            const int n = h[c].n;
            if (n < min_chain) continue;
            if (n < 2 * len) { k++; continue; }
            for (int64_t s0 = 0; s0 < n;) { int64_t e = std::min<int64_t>(n, s0 + len); if (n - e < len / 4) e = n; k++; s0 = e; }
        }
        return k;
    };
    int64_t len = env_int("PGA_DP_SEG_LEN", 0);
    if (len <= 0) {
        // what the short chains leave of the compute units, but never less than a quarter of them: a batch of one genome and
        // hundreds of small contigs still cuts the genome (its segments then share the chip with the small chains' workgroups)
        // (wave_walk: a wavefront per segment, four to a SIMD -- 4096 at once (a wavefront takes 18.5 us per 64-node batch with two on a SIMD, 26 us with six: measured on config 5, 2048 / 4096 / 6144 segments: 11.8 / 11.4 / 11.6 ms of connection scoring), however long the chains that are not cut: those go to the chain
        //  kernel in a launch of their own)
        const int64_t slots = wave_walk ? std::max(64, env_int("PGA_DP_SEG_WSLOTS", 4096)) : std::max(64, env_int("PGA_DP_SEG_SLOTS", 256) - 4);
        const int64_t budget = wave_walk ? slots : std::max<int64_t>(slots / 4, slots - n_short);
        len = std::max<int64_t>(512, ((cand + budget - 1) / budget + 63) & ~63ll);
        for (int it = 0; count_segs(len) > budget && len < cand; it++) len += it < 256 ? 64 : std::max<int64_t>(64, (len / 8) & ~63ll);
    }
    len = (len + 63) & ~63ll;
    int64_t cursor = tot_nodes;
    for (int c = 0; c < n_chains; c++) {
        const int n = h[c].n;
        if (n < min_chain || n < 2 * len) continue;
        plan.big.push_back(c);
        plan.max_big_n = std::max(plan.max_big_n, n);
        for (int64_t s0 = 0; s0 < n;) {
            int64_t e = std::min<int64_t>(n, s0 + len);
            if (n - e < len / 4) e = n;
            DpSeg sd;
            sd.chain = c; sd.s = (int32_t)s0; sd.e = (int32_t)e; sd.a = (int32_t)std::max<int64_t>(0, s0 - warm); sd.off = cursor;
            cursor += sd.e - sd.a;
            plan.max_seg_nodes = std::max(plan.max_seg_nodes, sd.e - sd.a);
            plan.max_seg_len = std::max(plan.max_seg_len, sd.e - sd.s);
            ChainDesc sc = h[c];
            sc.off = sd.off; sc.n = sd.e - sd.a; sc.rec_off = h[c].off + sd.a; sc.rebase = sd.a;
            plan.p1_chains.push_back(sc);
            plan.p1_slot.push_back(n_chains + (int32_t)plan.segs.size());
            plan.segs.push_back(sd);
            s0 = e;
        }
    }
    if (plan.big.empty()) { plan = DpSegPlan(); return false; }
    std::vector<char> is_big((size_t)n_chains, 0);
    for (int c : plan.big) is_big[(size_t)c] = 1;
    for (int c = 0; c < n_chains; c++) if (!is_big[(size_t)c]) { plan.p1_chains.push_back(h[c]); plan.p1_slot.push_back(c); }
    plan.extra = cursor - tot_nodes;
    return true;
}

namespace {
struct SegLayout {
    size_t segs, p1_chains, p1_slot, big, flags, first_bad, ctb, cw, hb, tv, ti, tmask, toff, nsp, sp_idx, sp_tb, sp_pp, sp_w, total;
};
SegLayout seg_layout(const DpSegPlan& plan, int n_chains, int64_t tot_nodes) {
    SegLayout L{};
    size_t off = 0;
    auto take = [&off](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
    const size_t n = (size_t)tot_nodes + 1, nt = (size_t)tot_nodes / 64 + (size_t)n_chains + 2, nc = (size_t)n_chains + 1;
    L.segs = take(sizeof(DpSeg) * plan.segs.size()); L.p1_chains = take(sizeof(ChainDesc) * plan.p1_chains.size());
    L.p1_slot = take(4 * plan.p1_slot.size()); L.big = take(4 * plan.big.size());
    L.flags = take(4 * PGA_SEG_ROUNDS * nc); L.first_bad = take(4 * PGA_SEG_ROUNDS * nc);
    L.ctb = take(4 * n); L.cw = take(8 * n); L.hb = take(4 * n);
    L.tv = take(8 * nt); L.ti = take(4 * nt); L.tmask = take(8 * nt); L.toff = take(4 * nt); L.nsp = take(4 * nc);
    L.sp_idx = take(4 * n); L.sp_tb = take(4 * n); L.sp_pp = take(4 * n); L.sp_w = take(8 * n);
    L.total = off;
    return L;
}
}  // namespace

size_t pga_dp_seg_bytes(const DpSegPlan& plan, int n_chains, int64_t tot_nodes) { return seg_layout(plan, n_chains, tot_nodes).total; }

hipError_t pga_dp_seg_bind(const DpSegPlan& plan, int n_chains, int64_t tot_nodes, void* arena, hipStream_t st, DpSegDev* out) {
    const SegLayout L = seg_layout(plan, n_chains, tot_nodes);
    char* b = (char*)arena;
    DpSegDev d{};
    d.segs = (const DpSeg*)(b + L.segs); d.p1_chains = (const ChainDesc*)(b + L.p1_chains); d.p1_slot = (const int32_t*)(b + L.p1_slot);
    d.big = (const int32_t*)(b + L.big); d.flags = (int32_t*)(b + L.flags); d.first_bad = (int32_t*)(b + L.first_bad);
    d.ctb = (int32_t*)(b + L.ctb); d.cw = (double*)(b + L.cw); d.hb = (uint32_t*)(b + L.hb);
    d.tv = (double*)(b + L.tv); d.ti = (int32_t*)(b + L.ti); d.tmask = (unsigned long long*)(b + L.tmask); d.toff = (int32_t*)(b + L.toff);
    d.nsp = (int32_t*)(b + L.nsp); d.sp_idx = (int32_t*)(b + L.sp_idx); d.sp_tb = (int32_t*)(b + L.sp_tb); d.sp_pp = (int32_t*)(b + L.sp_pp);
    d.sp_w = (double*)(b + L.sp_w);
    d.n_nodes = tot_nodes; d.n_segs = (int32_t)plan.segs.size(); d.n_p1 = (int32_t)plan.p1_chains.size(); d.n_big = (int32_t)plan.big.size();
    d.max_seg_nodes = plan.max_seg_nodes; d.max_seg_len = plan.max_seg_len; d.max_big_n = plan.max_big_n;
    // the plan and the cleared verification flags (flags: 0, first_bad: 0x7f7f7f7f) in ONE copy: they are the head of the workspace
    // (it was four copies here and two memsets in the launcher; the staging lives in the plan, which outlives the copy)
    plan.stage.assign(L.ctb, 0);
    char* h = plan.stage.data();
    if (!plan.segs.empty()) memcpy(h + L.segs, plan.segs.data(), sizeof(DpSeg) * plan.segs.size());
    if (!plan.p1_chains.empty()) memcpy(h + L.p1_chains, plan.p1_chains.data(), sizeof(ChainDesc) * plan.p1_chains.size());
    if (!plan.p1_slot.empty()) memcpy(h + L.p1_slot, plan.p1_slot.data(), 4 * plan.p1_slot.size());
    if (!plan.big.empty()) memcpy(h + L.big, plan.big.data(), 4 * plan.big.size());
    memset(h + L.first_bad, 0x7f, L.ctb - L.first_bad);
    hipError_t e = hipMemcpyAsync(b, h, L.ctb, hipMemcpyHostToDevice, st);
    *out = d;
    return e;
}

// C-ABI view of the plan (include/pyrodigal_amd.h): pure host arithmetic, usable without a device
extern "C" int pga_dp_plan_summary(int32_t n_chains, const int32_t* nodes_per_chain, int64_t out[4]) {
    if (n_chains < 0 || (n_chains > 0 && !nodes_per_chain) || !out) return PGA_EINVAL;
    std::vector<ChainDesc> h((size_t)n_chains);
    int64_t tot = 0;
    for (int c = 0; c < n_chains; c++) {
        if (nodes_per_chain[c] < 0) return PGA_EINVAL;
        h[(size_t)c] = ChainDesc{tot, 0, nodes_per_chain[c], 0, c, 1};
        tot += nodes_per_chain[c];
    }
    DpSegPlan plan;
    pga_dp_plan(h.data(), n_chains, tot, plan);
    out[0] = (int64_t)plan.big.size(); out[1] = (int64_t)plan.segs.size(); out[2] = plan.max_seg_nodes; out[3] = plan.extra;
    // every node of a cut chain belongs to exactly one segment, segments are in order and start their walk at most `warm` early
    for (size_t k = 0; k < plan.segs.size(); k++) {
        const DpSeg& s = plan.segs[k];
        const bool first = k == 0 || plan.segs[k - 1].chain != s.chain;
        if (s.a > s.s || s.s >= s.e || s.e > h[(size_t)s.chain].n || (first ? s.s != 0 : s.s != plan.segs[k - 1].e)) return PGA_EDEVICE;
        const bool last = k + 1 == plan.segs.size() || plan.segs[k + 1].chain != s.chain;
        if (last && s.e != h[(size_t)s.chain].n) return PGA_EDEVICE;
    }
    return PGA_OK;
}

static void launch_dp_segmented(const ChainDesc* d_chains, int n_chains, const ModelConst* d_models, DpBuffers buf, hipStream_t st,
                                const DpSegDev& sg) {
    const dim3 blk(256);
    auto blocks = [](int n) { return (unsigned)((n + 255) / 256); };
    // (flags and first_bad arrive cleared with the plan: pga_dp_seg_bind)
    if (sg.wave_groups != nullptr && sg.wave_buf != nullptr) {
        // the segments (the first n_segs sub-chains of the plan) a wavefront each; the chains that are not cut by the chain kernel
        pga_launch_dp_wave_sub(sg.p1_chains, sg.n_segs, *sg.wave_groups, d_models, buf, *sg.wave_buf, st, sg.p1_slot);
        if (sg.n_p1 > sg.n_segs)
            hipLaunchKernelGGL(k_dp_tree_mw, dim3(sg.n_p1 - sg.n_segs), dim3(64 * PGA_MW_WAVES), 0, st, sg.p1_chains + sg.n_segs, buf.src, buf.tgt, d_models, buf,
                               (const int32_t*)nullptr, sg.p1_slot + sg.n_segs);
    } else
    hipLaunchKernelGGL(k_dp_tree_mw, dim3(sg.n_p1), dim3(64 * PGA_MW_WAVES), 0, st, sg.p1_chains, buf.src, buf.tgt, d_models, buf,
                       (const int32_t*)nullptr, sg.p1_slot);
    hipLaunchKernelGGL(k_seg_gather, dim3(blocks(sg.max_seg_len), sg.n_segs), blk, 0, st, sg.segs, d_chains, buf.traceb, sg.ctb);
    for (int r = 0; r < PGA_SEG_ROUNDS; r++) {
        const int32_t* gate = r == 0 ? nullptr : sg.flags + (size_t)(r - 1) * n_chains;
        const int32_t* from = r == 0 ? nullptr : sg.first_bad + (size_t)(r - 1) * n_chains;
        const dim3 per_node(blocks(sg.max_big_n), sg.n_big);
        hipMemsetAsync(sg.hb, 0, sizeof(uint32_t) * (size_t)sg.n_nodes, st);
        hipLaunchKernelGGL(k_seg_weights, per_node, blk, 0, st, d_chains, sg.big, gate, buf.src, buf.tgt, d_models, buf, sg.ctb, sg.cw, sg.hb);
        const int H = seg_height();
        for (int k = 2; k <= H; k++)
            hipLaunchKernelGGL(k_seg_height, per_node, blk, 0, st, d_chains, sg.big, gate, (const int32_t*)buf.traceb, sg.hb, k);
        hipLaunchKernelGGL(k_spine_count, per_node, blk, 0, st, d_chains, sg.big, gate, (const uint32_t*)sg.hb, sg.tmask, H);
        hipLaunchKernelGGL(k_spine_scan, dim3(sg.n_big), dim3(1024), 0, st, d_chains, sg.big, gate, (const unsigned long long*)sg.tmask, sg.toff, sg.nsp);
        hipLaunchKernelGGL(k_spine_fill, per_node, blk, 0, st, d_chains, sg.big, gate, (const int32_t*)buf.traceb, (const double*)sg.cw,
                           (const unsigned long long*)sg.tmask, (const int32_t*)sg.toff, sg.sp_idx, sg.sp_tb, sg.sp_pp, sg.sp_w);
        hipLaunchKernelGGL(k_dp_rescore, dim3(sg.n_big), dim3(64), 0, st, d_chains, sg.big, gate, from, buf, (const int32_t*)sg.toff,
                           (const int32_t*)sg.nsp, (const int32_t*)sg.sp_idx, (const int32_t*)sg.sp_tb, (const int32_t*)sg.sp_pp, (const double*)sg.sp_w);
        for (int cl = H - 1; cl >= 0; cl--)
            hipLaunchKernelGGL(k_seg_leaves, per_node, blk, 0, st, d_chains, sg.big, gate, from, buf, sg.cw, sg.hb, cl);
        hipLaunchKernelGGL(k_seg_build_far, per_node, blk, 0, st, d_chains, sg.big, gate, buf.src, buf.tgt, d_models, buf, sg.tv, sg.ti);
        hipLaunchKernelGGL(k_seg_build_upper, dim3(sg.n_big), blk, 0, st, d_chains, sg.big, gate, buf.src, buf.tgt, buf, sg.tv, sg.ti);
        hipLaunchKernelGGL(k_dp_verify, per_node, blk, 0, st, d_chains, sg.big, gate, buf.src, buf.tgt, d_models, buf, sg.ctb,
                           sg.flags + (size_t)r * n_chains, sg.first_bad + (size_t)r * n_chains, from);
        if (sg.h_round != nullptr && r == 0) {       // (after round 0 only, the common clean case: the later rounds run gated on the device, without a host round trip each)
            // Nearly every launch passes its first verification (configs 2 and 5: no rejection with the 4096-node warm-up), and the
            // rounds behind it then are sixteen gated launches each that find their gate shut -- 5 us apiece, 0.17 ms of a 3.3 ms
            // call.  One small read-back instead: the host sees the round's verdict and stops issuing.
            if (hipMemcpyAsync(sg.h_round, sg.flags + (size_t)r * n_chains, sizeof(int32_t) * (size_t)n_chains, hipMemcpyDeviceToHost, st) == hipSuccess &&
                hipStreamSynchronize(st) == hipSuccess) {
                bool clean = true;
                for (int k = 0; k < n_chains; k++) clean = clean && sg.h_round[k] == 0;
                if (clean) return;              // (the flags of the rounds not run stay zero: nothing is left for the serial walk either)
            }
        }
    }
    // chains that never verified clean: the serial walk
    hipLaunchKernelGGL(k_dp_tree_mw, dim3(n_chains), dim3(64 * PGA_MW_WAVES), 0, st, d_chains, buf.src, buf.tgt, d_models, buf,
                       (const int32_t*)(sg.flags + (size_t)(PGA_SEG_ROUNDS - 1) * n_chains), (const int32_t*)nullptr);
}

void pga_launch_dp(const ChainDesc* d_chains, int n_chains, const ModelConst* d_models, DpBuffers buf,
                   int final, hipStream_t st, const DpSegDev* seg) {
    if (n_chains <= 0) return;
    if (final && seg != nullptr && seg->n_segs > 0) { launch_dp_segmented(d_chains, n_chains, d_models, buf, st, *seg); return; }
    if (!final) {     // training pass: once per genome, the window-scanning kernel is plenty
        hipLaunchKernelGGL((k_dp_chain<16, false>), dim3(n_chains), dim3(64 * 16), 0, st, d_chains, buf.src, buf.tgt, d_models,
                           buf.score, buf.traceb, buf.tbn, buf.ov_mark, buf.max_index, buf.max_score, buf.ipath);
        return;
    }
#define PGA_DP_LAUNCH(WAVES) hipLaunchKernelGGL(k_dp_chain<WAVES>, dim3(n_chains), dim3(64 * WAVES), 0, st, d_chains, buf.src, buf.tgt, \
        d_models, buf.score, buf.traceb, buf.tbn, buf.ov_mark, buf.max_index, buf.max_score, buf.ipath)
    const char* kern = getenv("PGA_DP_KERNEL");
    if (!kern || strcmp(kern, "scan") != 0) {
        // many chains: one wave each fills the chip; few chains: latency-bound, 3 cooperating waves per chain
        bool mw = n_chains < 2048;
        if (kern && strcmp(kern, "tree1") == 0) mw = false;
        if (kern && strcmp(kern, "tree3") == 0) mw = true;
        if (mw) hipLaunchKernelGGL(k_dp_tree_mw, dim3(n_chains), dim3(64 * PGA_MW_WAVES), 0, st, d_chains, buf.src, buf.tgt, d_models, buf,
                                   (const int32_t*)nullptr, (const int32_t*)nullptr);
        else hipLaunchKernelGGL(k_dp_tree, dim3(n_chains), dim3(64), 0, st, d_chains, buf.src, buf.tgt, d_models, buf);
        return;
    }
    // PGA_DP_KERNEL=scan: the window-scanning kernels (kept as an independent cross-check of the tree kernel)
    // few chains: latency-bound, give each chain a whole workgroup; many chains: one wave each fills the chip
    int waves = n_chains >= 4096 ? 1 : (n_chains >= 1024 ? 4 : 16);
    if (const char* e = getenv("PGA_DP_WAVES")) { const int v = atoi(e); if (v == 1 || v == 4 || v == 16) waves = v; }   // tuning / tests
    if (waves == 1) PGA_DP_LAUNCH(1);
    else if (waves == 4) PGA_DP_LAUNCH(4);
    else PGA_DP_LAUNCH(16);
#undef PGA_DP_LAUNCH
}
