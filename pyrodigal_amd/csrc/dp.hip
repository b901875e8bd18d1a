// Connection-scoring dynamic programme (Prodigal's dprog / score_connection) for gfx950.
//
// What is computed (ref: lib.pyx:1205-1237 `_score_connections`, _connection.h:94-408 the four
// split scorers, impl/generic.h:29-36 the six skip conditions): for every node i in position
// order, the best predecessor j in the window [lo_i, i):
//     score[i] = max(0, max_j (score[j] + w(j, i))),  ties -> largest j,  traceb[i] = that j.
//
// Mapping onto CDNA4 -- one 64-wide wavefront per (contig, model) chain:
//   * lane t owns target node i0+t of the current 64-node batch and keeps (best, traceb, ov_mark)
//     in registers; candidates j are visited in ascending order, wave-uniformly, so the per-lane
//     ">=" update reproduces the reference's sequential scan bit for bit (no cross-lane
//     reduction, no reassociation of floating-point adds);
//   * the source record of candidate j is wave-uniform (one 64-byte DpSrc, scalar-loaded), its
//     dynamic fields (score, ndx of its traceb) are read once per 64-source tile as a coalesced
//     vector load and broadcast with v_readlane;
//   * a __ballot over the tile builds the visit mask: gene-end sources that were never reached
//     (traceb == -1) can connect to nothing (ref: _connection.h:110-114) and are skipped for the
//     whole wave, and a second __ballot over the targets skips a source no lane can use
//     (the six conditions of impl/generic.h:29-36 folded into the per-kind predicates);
//   * the 63 candidates inside the batch itself are the other lanes: when the loop reaches
//     source i0+k, lane k has already seen every j < i0+k, so its registers hold final values.
//   * third-node indirections (star_ptr -> n3) never happen in the loop: dp_prepare folds
//     cs(n3)+igm(.,.) and n3.{ndx,stop_val} into the source/target records.
//
// Arithmetic: IEEE double, compiled with -ffp-contract=off, same operation order as the
// reference; (2 - d/60)*0.15*st_wt comes from a host-computed 61-entry table.

#include "pga_internal.h"
#include "dev_common.h"

#include <limits.h>

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
             NodeArrays nd, const ModelConst* __restrict__ models, DpSrc* __restrict__ src, DpTgt* __restrict__ tgt) {
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    g += node_begin;
    const int c = find_chain(chains, n_chains, g);
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
    s.cs = cs_c[i] + cs_s[i];
    s._pad2[0] = s._pad2[1] = 0.0;
    for (int k = 0; k < 3; k++) {
        s.x[k] = 0.0; t.n3ndx[k] = 0; t.n3stop[k] = 0;
        if (!stop) continue;
        const int p = sp[i * 3 + k];
        if (p < 0) continue;
        meta |= 1 << (4 + k);
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
    if ((kind == 2 || kind == 1) && ndx[lo] > my_stop) {
        int a = 0, b = lo;            // find last p in [0, lo) with ndx[p] <= my_stop
        while (a < b) { int m = (a + b) >> 1; if (ndx[m] <= my_stop) a = m + 1; else b = m; }
        lo = (a > 0 && ndx[a - 1] == my_stop) ? a - 1 : 0;
    }
    lo = lo < PGA_MAX_NODE_DIST ? 0 : lo - PGA_MAX_NODE_DIST;
    t.lo = lo; t._pad = 0;
    src[g] = s; tgt[g] = t;
}

struct Target {
    int kind, frame, ndx, stop_val, meta, lo, i;
    double cs, csd, x0, x1, x2;
    int n3n0, n3n1, n3n2, n3s0, n3s1, n3s2;
};

// Score candidate source j (wave-uniform fields s_*, sj, tbnj) against this lane's target.
// One call = one iteration of the loop in _connection.h:386-408 for all 64 lanes.
__device__ __forceinline__ void eval_source(const int j, const int s_ndx, const int s_stop, const int s_meta,
                                            const double s_cs, const double s_x0, const double s_x1, const double s_x2,
                                            const double sj, const int tbnj, const Target& T, const double negc,
                                            const double* s_igm, double& best, int& tb, int& ov) {
    const int sk = PGA_KIND(s_meta), sf = PGA_FRAME(s_meta);
    bool ok = (j >= T.lo) && (j < T.i);
    double w = 0.0; int mf = -1;
    if (sk == 0) {
        // 5'fwd -> 3'fwd: a gene (ref: _connection.h:166-174; skip condition 5: same frame only)
        ok = ok && T.kind == 1 && T.frame == sf && T.stop_val < s_ndx;
        w = s_cs;
    } else if (sk == 2) {
        // 5'rev -> 5'fwd (ref: :125-130) and 5'rev -> 3'rev (ref: :337-342)
        const bool a = T.kind == 0 && s_ndx < T.ndx;
        const bool b = T.kind == 3 && s_ndx < T.ndx - 2;
        ok = ok && (a || b);
        w = b ? igm_apart(T.ndx - s_ndx, negc, s_igm) : negc;
    } else if (sk == 3) {
        // 3'rev -> 5'rev: a gene (ref: :228-235; skip condition 6) and 3'rev -> 3'rev operon (ref: :345-356)
        const bool a = T.kind == 2 && T.frame == sf && s_stop > T.ndx;
        const bool b = T.kind == 3 && s_stop > T.ndx && PGA_SPVALID(T.meta, sf);
        ok = ok && (a || b);
        w = a ? T.cs : sel3(sf, T.x0, T.x1, T.x2);
    } else {
        // forward stop as source: connects to all four target kinds
        if (T.kind == 0) {            // 3'fwd -> 5'fwd intergenic (ref: :117-124)
            ok = ok && (s_ndx + 2 < T.ndx);
            w = igm_apart(T.ndx - s_ndx, negc, s_igm);
        } else if (T.kind == 1) {     // 3'fwd -> 3'fwd operon through j's overlapping start (ref: :177-188)
            ok = ok && T.stop_val < s_ndx && PGA_SPVALID(s_meta, T.frame);
            w = sel3(T.frame, s_x0, s_x1, s_x2);
        } else if (T.kind == 2) {     // 3'fwd -> 5'rev overlapping opposite 3' ends (ref: :238-254)
            const int ovlp = (s_ndx + 2) - (T.stop_val - 2) + 1;
            ok = ok && !(T.stop_val - 2 >= s_ndx + 2) && ovlp < PGA_MAX_OPP_OVLP
                    && (s_ndx - T.stop_val) < (T.ndx - s_ndx + 3)
                    && (s_ndx - T.stop_val) < (T.stop_val - 3 - tbnj);
            w = T.csd;
        } else {                      // 3'fwd -> 3'rev, possibly through one of i's overlapping starts (ref: :288-336)
            const int left = s_ndx + 2, right = T.ndx - 2;
            ok = ok && left < right;
            double maxval = 0.0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int n3s = sel3i(k, T.n3s0, T.n3s1, T.n3s2), n3n = sel3i(k, T.n3n0, T.n3n1, T.n3n2);
                const double cur = sel3(k, T.x0, T.x1, T.x2);
                const int ovlp = left - n3s + 3;
                const bool take = PGA_SPVALID(T.meta, k) && ovlp > 0 && ovlp < PGA_MAX_OPP_OVLP && ovlp < n3n - left
                                  && ovlp < n3s - tbnj - 2 && cur > maxval;
                if (take) { mf = k; maxval = cur; }
            }
            w = mf != -1 ? maxval : negc;
        }
    }
    const double val = sj + w;
    if (ok && val >= best) { best = val; tb = j; ov = mf; }
}

// One wavefront per chain.
__global__ void __launch_bounds__(64)
k_dp_wave(const ChainDesc* __restrict__ chains, const DpSrc* __restrict__ g_src, const DpTgt* __restrict__ g_tgt,
          const ModelConst* __restrict__ models, double* g_score, int32_t* g_traceb, int32_t* g_tbn, int8_t* g_ov,
          int32_t* __restrict__ max_index, double* __restrict__ max_score, int32_t* __restrict__ ipath) {
    __shared__ double s_igm[64];
    const ChainDesc cd = chains[blockIdx.x];
    const int lane = threadIdx.x;
    const int n = cd.n;
    const ModelConst* mc = &models[cd.model];
    s_igm[lane] = mc->igm[lane];
    __syncthreads();
    const double negc = mc->negc;
    const DpSrc* __restrict__ src = g_src + cd.off;
    const DpTgt* __restrict__ tgt = g_tgt + cd.off;
    double* score = g_score + cd.off; int32_t* traceb = g_traceb + cd.off;
    int32_t* tbn = g_tbn + cd.off; int8_t* ovm = g_ov + cd.off;

    double end_best = -1.0; int end_idx = -1, end_tb = -1;     // _find_max_index (ref: lib.pyx:1239-1251)

    for (int i0 = 0; i0 < n; i0 += 64) {
        Target T;
        T.i = i0 + lane;
        const bool act = T.i < n;
        {
            const int ii = act ? T.i : n - 1;
            const DpSrc me = src[ii]; const DpTgt mt = tgt[ii];
            T.kind = PGA_KIND(me.meta); T.frame = PGA_FRAME(me.meta); T.meta = me.meta;
            T.ndx = me.ndx; T.stop_val = me.stop_val; T.cs = me.cs; T.csd = me.cs + negc;
            T.x0 = me.x[0]; T.x1 = me.x[1]; T.x2 = me.x[2];
            T.n3n0 = mt.n3ndx[0]; T.n3n1 = mt.n3ndx[1]; T.n3n2 = mt.n3ndx[2];
            T.n3s0 = mt.n3stop[0]; T.n3s1 = mt.n3stop[1]; T.n3s2 = mt.n3stop[2];
            T.lo = act ? mt.lo : INT_MAX;
            if (!act) T.i = -1;       // j < T.i is never true: lane stays idle
        }
        double best = 0.0; int tb = -1, ov = -1;
        const int wlo = wave_min_i32(T.lo);

        // ---- candidates from earlier batches, one 64-source tile at a time
        for (int t0 = wlo & ~63; t0 < i0; t0 += 64) {
            const int sidx = t0 + lane;
            const double tsc = score[sidx];
            const int ttb = tbn[sidx];
            const int smeta = src[sidx].meta;
            const int sk = PGA_KIND(smeta);
            const bool dead = (sk == 1 || sk == 2) && ttb == -1;   // gene end never reached: connects to nothing
            unsigned long long visit = __ballot(sidx >= wlo && !dead);
            while (visit) {
                const int k = __builtin_ctzll(visit);
                visit &= visit - 1;
                const int j = t0 + k;
                const DpSrc s = src[j];
                const double sj = readlane_f64(tsc, k);
                const int tbnj = __builtin_amdgcn_readlane(ttb, k);
                eval_source(j, s.ndx, s.stop_val, s.meta, s.cs, s.x[0], s.x[1], s.x[2], sj, tbnj, T, negc, s_igm, best, tb, ov);
            }
        }
        // ---- candidates inside this batch: lane k is final once the loop reaches source i0+k
        const int kmax = min(63, n - 1 - i0);
        for (int k = 0; k < kmax; k++) {
            const int j = i0 + k;
            const DpSrc s = src[j];
            const int sk = PGA_KIND(s.meta);
            const int tbk = __builtin_amdgcn_readlane(tb, k);
            if ((sk == 1 || sk == 2) && tbk == -1) continue;
            const double sj = readlane_f64(best, k);
            int tbnj = 0;
            if (sk == 1) tbnj = src[tbk].ndx;
            eval_source(j, s.ndx, s.stop_val, s.meta, s.cs, s.x[0], s.x[1], s.x[2], sj, tbnj, T, negc, s_igm, best, tb, ov);
        }
        if (act) {
            score[T.i] = best; traceb[T.i] = tb; ovm[T.i] = (int8_t)ov;
            tbn[T.i] = tb < 0 ? -1 : src[tb].ndx;
            if ((T.kind == 1 || T.kind == 2) && best >= end_best) { end_best = best; end_idx = T.i; end_tb = tb; }
        }
    }
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

}  // namespace

void pga_launch_dp_prepare(const ChainDesc* d_chains, int n_chains, int64_t node_begin, int64_t total_nodes,
                           const NodeArrays& nodes, const ModelConst* d_models, DpBuffers buf, hipStream_t st) {
    if (total_nodes <= 0) return;
    const int threads = 256;
    const int64_t blocks = (total_nodes + threads - 1) / threads;
    hipLaunchKernelGGL(k_dp_prepare, dim3((unsigned)blocks), dim3(threads), 0, st,
                       d_chains, n_chains, node_begin, total_nodes, nodes, d_models, buf.src, buf.tgt);
}

void pga_launch_dp(const ChainDesc* d_chains, int n_chains, const ModelConst* d_models, DpBuffers buf,
                   int final, hipStream_t st) {
    (void)final;
    if (n_chains <= 0) return;
    hipLaunchKernelGGL(k_dp_wave, dim3(n_chains), dim3(64), 0, st,
                       d_chains, buf.src, buf.tgt, d_models, buf.score, buf.traceb, buf.tbn, buf.ov_mark,
                       buf.max_index, buf.max_score, buf.ipath);
}
