// Wave-batch connection scoring for launches with many chains (one wavefront per (contig, model) chain).
//
// Same recurrence as dp.hip (ref: lib.pyx:1205-1237, _connection.h:94-408, impl/generic.h:29-36); what changes is where a
// target's candidates come from.  Lane t of the wave owns target node i0 + t of the current 64-node batch.  Sources older than
// the batch never cost a per-lane search:
//   * far gene ends (F5 / R3 targets): every finished node leaves a = score + the constant intergenic term; the wave keeps, in
//     registers, the lexicographic maxima of `a` over the last 64 whole blocks ending at the batch (S1) and at the block before
//     (S2), and the inclusive prefix maxima inside the previous block; a window that starts inside a block reads that block's
//     suffix maximum, stored once per node when the block was finished.  A far range is then at most three pieces;
//   * forward stops: the best start / operon partner of the ORF met so far is a per-frame running maximum that restarts at every
//     forward stop of the frame (a static bit of each forward stop says whether it lies in the ORF of the next stop of a frame);
//   * reverse starts and reverse stops: only the LAST reverse stop of a frame can hold a later node in its ORF, so "own stop" and
//     "operon partner" are three uniform records;
//   * forward stops overlapping the 3' end of a reverse gene: a short chain through the forward stops (static links);
//   * the few sources within 3 * OPER_DIST bases before the batch are visited pair by pair like the sources inside the batch:
//     one wave-uniform source at a time (v_readlane), the serial critical path of a chain.
// Every scalar routine lives in dpw_core.h and also runs on the host in tests/dpw_model.cpp, which is compared with the plain
// restatement of the reference's loop on the CPU (tests/test_dpw_model.py); this file adds the wave mechanics.
//
// Records are SoA: topology arrays shared by every model of a contig (ndx, stop_val, kf, lo, q1, q2), per-chain cs and a
// 64-byte record of extras per STOP node; results score / traceb / ov_mark / position of the traceb node, plus the suffix
// maxima of chains longer than one window.

#include "pga_internal.h"
#include <atomic>
#include <type_traits>
#include "dev_common.h"
#include "dpw_core.h"
#include "dpw_walk_gfx950.inc"     // DPW_ASM_NEAR / DPW_ASM_WALK: the pair steps of k_dp_wave in gfx950 assembly (tools/gen_dpw_walk.py)

namespace {

__device__ __forceinline__ double rl_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ int rl_i32(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// contig of group node g (first-node offsets, n_contigs + 1 entries)
__device__ __forceinline__ int contig_of_node(const int32_t* __restrict__ cbase, int n_contigs, int g) {
    int lo = 0, hi = n_contigs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (cbase[mid] <= g) lo = mid; else hi = mid - 1; }
    return lo;
}

// One thread per topology node.  The expensive kind is the forward stop: it needs the next forward stop of every frame after it (q2 and
// the operon bits), some forty nodes ahead on average -- and every lane of a wave pays for that walk, whatever its own kind.  So the
// workgroup first turns the forward-stop flags of its 256 nodes and of the 192 behind them into three bit masks (one per frame) in
// LDS, and a forward stop finds its three successors with a few bit scans; only when a frame has none inside the masks and the
// contig goes on behind them does it walk (dpw_topo_node without a hint).  PGA_DPW_TOPO_WALK=1: always walk (cross-check).
constexpr int TOPO_AHEAD = 192, TOPO_WORDS = (256 + TOPO_AHEAD) / 64;
__global__ void __launch_bounds__(256)
k_dpw_topo(const int32_t* __restrict__ ndx, const int32_t* __restrict__ stopv, const uint8_t* __restrict__ type, const int8_t* __restrict__ strand,
           const int32_t* __restrict__ cbase, int n_contigs, int n_nodes, DpwTopoArrays ta, const int use_masks) {
    __shared__ int s_c0;
    __shared__ unsigned long long s_m[3][TOPO_WORDS];
    const int g0 = blockIdx.x * blockDim.x, g = g0 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0 && ta.scur != nullptr) { ta.scur[0] = 0u; ta.scur[1] = 0u; }      // the schedule's miss counter (k_dpw_sched runs behind this kernel)
    if (threadIdx.x == 0) s_c0 = contig_of_node(cbase, n_contigs, g0);     // one search per workgroup, then a short walk
    if (use_masks) {
        auto flags = [&](const int idx, const int word) {
            const bool in = idx < n_nodes;
            const bool f3 = in && type[in ? idx : 0] == 3 && strand[in ? idx : 0] == 1;
            const int fr = in ? ndx[idx] % 3 : 0;
#pragma unroll
            for (int f = 0; f < 3; f++) { const unsigned long long m = __ballot(f3 && fr == f); if (lane == 0) s_m[f][word] = m; }
        };
        flags(g, wave);
        if (wave < TOPO_AHEAD / 64) flags(g0 + 256 + threadIdx.x, 4 + wave);
    }
    __syncthreads();
    if (g >= n_nodes) return;
    int c = s_c0;
    while (c + 1 < n_contigs && cbase[c + 1] <= g) c++;
    const int b0 = cbase[c], n = cbase[c + 1] - b0;
    DpwF3Hint hint{0, 0};
    bool have = false;
    if (use_masks && type[g] == 3 && strand[g] == 1) {
        const int k = g - g0, e = b0 + n, my_ndx = ndx[g];
        const bool covered = g0 + 64 * TOPO_WORDS >= e;       // the masks reach the end of the contig
        int q2 = n; have = true;
#pragma unroll
        for (int f = 0; f < 3; f++) {
            int j = -1;
            for (int w = (k + 1) >> 6; w < TOPO_WORDS && j < 0; w++) {
                unsigned long long m = s_m[f][w];
                if (w == (k + 1) >> 6) m &= ~0ull << ((k + 1) & 63);
                if (m) j = g0 + 64 * w + __builtin_ctzll(m);
            }
            if (j < 0) { if (!covered) have = false; continue; }
            if (j >= e) continue;                              // the frame has no forward stop behind this one on the contig
            q2 = min(q2, j - b0);
            if (stopv[j] < my_ndx) hint.bits |= 1 << (4 + f);   // inside that stop's ORF: an operon candidate for it
        }
        hint.q2 = q2;
    }
    const DpwTopo t = dpw_topo_node(ndx + b0, stopv + b0, type + b0, strand + b0, n, g - b0, have ? &hint : nullptr);
    ta.kf[g] = t.kf; ta.lo[g] = t.lo; ta.q1[g] = t.q1; ta.q2[g] = t.q2;
}

// The same per contig with the contig's nodes in LDS (launches whose contigs all hold at most TOPO_LDS_NODES nodes: every launch
// of short contigs).  What a node's topology costs is searches among its neighbours -- the first node within 3 * OPER_DIST bases, the
// first forward stop that can overlap a reverse gene's 3' end (a binary search and a scan), the window's first node --, a dozen
// dependent loads per node; from global memory the kernel above spends two thirds of its wave time waiting for them
// (profiles/r04_c_config4_sq_counters.md).  One workgroup stages its contig's four node arrays (10 bytes per node) and runs the
// very same routine (dpw_topo_node) on the LDS copies.
// Round 5: each of the four kinds of node takes a path of its own through that routine, and 64 neighbouring nodes are a mix of all
// four, so a wavefront ran every path at a quarter of its lanes.  The workgroup now lists its contig's nodes BY KIND first (two more
// bytes of LDS per node) and walks the list: a wavefront's 64 nodes are of one kind except where two kinds meet.  The forward stops
// find the next forward stop of every frame in bit masks over the WHOLE contig (one word per 64 nodes and frame, built once).
constexpr int TOPO_LDS_NODES = 6144;
__global__ void __launch_bounds__(256)
k_dpw_topo_lds(const int32_t* __restrict__ ndx, const int32_t* __restrict__ stopv, const uint8_t* __restrict__ type, const int8_t* __restrict__ strand,
               const int32_t* __restrict__ cbase, DpwTopoArrays ta) {
    extern __shared__ int s_topo[];
    __shared__ unsigned long long s_m[3][TOPO_LDS_NODES / 64];
    __shared__ int s_cnt[4];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    if (c == 0 && tid == 0 && ta.scur != nullptr) { ta.scur[0] = 0u; ta.scur[1] = 0u; }      // the schedule's miss counter (k_dpw_sched runs behind this kernel)
    const int b0 = cbase[c], n = cbase[c + 1] - b0;
    if (n <= 0) return;
    int32_t* const l_ndx = s_topo; int32_t* const l_stop = s_topo + n;
    uint8_t* const l_type = (uint8_t*)(s_topo + 2 * n); int8_t* const l_strand = (int8_t*)(l_type + n);
    uint16_t* const l_perm = (uint16_t*)(s_topo + 2 * n + ((2 * n + 3) >> 2));      // behind the two byte arrays, 4-byte aligned
    if (tid < 4) s_cnt[tid] = 0;
    for (int i = tid; i < n; i += 256) { l_ndx[i] = ndx[b0 + i]; l_stop[i] = stopv[b0 + i]; l_type[i] = type[b0 + i]; l_strand[i] = strand[b0 + i]; }
    __syncthreads();
    // the forward stops of the contig by frame as bit masks, and how many nodes of each kind there are
    const int n_round = (n + 255) & ~255;
    for (int i = tid; i < n_round; i += 256) {
        const bool in = i < n;
        const bool rev = in && l_strand[in ? i : 0] != 1, stop = in && l_type[in ? i : 0] == 3;
        const int fr = in ? l_ndx[i] % 3 : 0;
        const bool f3 = in && stop && !rev;
#pragma unroll
        for (int f = 0; f < 3; f++) { const unsigned long long m = __ballot(f3 && fr == f); if (lane == 0) s_m[f][i >> 6] = m; }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned long long m = __ballot(in && ((rev ? 2 : 0) | (stop ? 1 : 0)) == k);
            if (lane == 0 && m) atomicAdd(&s_cnt[k], __popcll(m));
        }
    }
    __syncthreads();
    // list position of every node: its kind's range, in any order inside it (a node's result does not depend on the others')
    const int c0 = s_cnt[0], c1 = s_cnt[1], c2 = s_cnt[2];
    __syncthreads();
    if (tid == 0) { s_cnt[0] = 0; s_cnt[1] = c0; s_cnt[2] = c0 + c1; s_cnt[3] = c0 + c1 + c2; }
    __syncthreads();
    for (int i = tid; i < n_round; i += 256) {
        const bool in = i < n;
        const bool rev = in && l_strand[in ? i : 0] != 1, stop = in && l_type[in ? i : 0] == 3;
        const int kind = (rev ? 2 : 0) | (stop ? 1 : 0);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned long long m = __ballot(in && kind == k);
            if (!m) continue;
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_cnt[k], __popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            if (in && kind == k) l_perm[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)i;
        }
    }
    __syncthreads();
    const int n_words = (n + 63) >> 6;
    for (int p = tid; p < n; p += 256) {
        const int i = l_perm[p];
        DpwF3Hint hint{0, 0};
        bool have = false;
        if (l_type[i] == 3 && l_strand[i] == 1) {
            const int my_ndx = l_ndx[i];
            int q2 = n; have = true;
#pragma unroll
            for (int f = 0; f < 3; f++) {
                int j = -1;
                for (int w = (i + 1) >> 6; w < n_words && j < 0; w++) {
                    unsigned long long m = s_m[f][w];
                    if (w == (i + 1) >> 6) m &= ~0ull << ((i + 1) & 63);
                    if (m) j = 64 * w + __builtin_ctzll(m);
                }
                if (j < 0) continue;                                   // the frame has no forward stop behind this one on the contig
                q2 = min(q2, j);
                if (l_stop[j] < my_ndx) hint.bits |= 1 << (4 + f);      // inside that stop's ORF: an operon candidate for it
            }
            hint.q2 = q2;
        }
        const DpwTopo t = dpw_topo_node(l_ndx, l_stop, l_type, l_strand, n, i, have ? &hint : nullptr);
        const int g = b0 + i;
        ta.kf[g] = t.kf; ta.lo[g] = t.lo; ta.q1[g] = t.q1; ta.q2[g] = t.q2;
    }
}
__global__ void __launch_bounds__(256)
k_dpw_chain(const ChainDesc* __restrict__ chains, int n_chains, int64_t node_begin, int64_t total, NodeArrays nd, DpwTopoArrays ta,
            const ModelConst* __restrict__ models, double* __restrict__ cs, DpwExt* __restrict__ ext) {
    __shared__ int s_c0;
    const int64_t blk0 = node_begin + (int64_t)blockIdx.x * blockDim.x;
    const int64_t g = blk0 + threadIdx.x;
    const bool in_range = g < node_begin + total;
    const int c = find_chain_block(chains, n_chains, blk0, in_range ? g : blk0, &s_c0);
    if (!in_range) return;
    const int64_t off = chains[c].off, toff = chains[c].topo_off;
    const int i = (int)(g - off);
    cs[g] = nd.cscore[g] + nd.sscore[g];
    const int kind = DPW_KIND(ta.kf[toff + i]);
    if (!(kind & 1)) return;
    const ModelConst* mc = &models[chains[c].model];
    const DpwModel M{mc->st_wt, mc->negc, mc->igm};
    DpwExt e;
    dpw_chain_ext(nd.ndx + toff, nd.stop_val + toff, nd.strand + toff, ta.q2 + toff, nd.cscore + off, nd.sscore + off, nd.rscore + off,
                  nd.uscore + off, nd.star_ptr + off * 3, i, kind == 3, M, e);
    ext[g] = e;
}

// lexicographic (value, index) maximum: ties go to the larger index
__device__ __forceinline__ void lex_max(double& v, int& i, const double ov, const int oi) {
    const bool c = ov > v || (ov == v && oi > i);
    v = c ? ov : v; i = c ? oi : i;
}
// inclusive prefix (lower lanes first) of the lexicographic maximum
__device__ __forceinline__ void wave_prefix_lexmax(double& v, int& i, const int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double ov = __shfl_up(v, d, 64); const int oi = __shfl_up(i, d, 64);
        if (lane >= d) lex_max(v, i, ov, oi);
    }
}
__device__ __forceinline__ void wave_suffix_lexmax(double& v, int& i, const int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double ov = __shfl_down(v, d, 64); const int oi = __shfl_down(i, d, 64);
        if (lane + d < 64) lex_max(v, i, ov, oi);
    }
}
// Cross-lane moves through the data-parallel primitives of the vector ALU (DPP) instead of ds_bpermute: a lone wavefront waits out
// the full LDS round trip of every __shfl, six of them in a row per scan, and these scans sit on the serial path of a chain.
// CTRL: 0x111 .. 0x11f row_shr:1 .. 15, 0x138 wave_shr:1, 0x142 row_bcast15, 0x143 row_bcast31, 0x140 row_mirror, 0x141 row_half_mirror,
// quad_perm otherwise.  Lanes that receive nothing (shifted in from outside the row, or masked out by ROW_MASK) get `old`.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_f64(const double old, const double v) {
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_i32(const int old, const int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false); }
// (no NaNs, and a zero of either sign is never compared with the other here: v_max_f64 gives what `b > a ? b : a` gives, in one instruction)
__device__ __forceinline__ double fmax_nn(const double a, const double b) { return __builtin_fmax(a, b); }

// wave-wide maximum of v (never NaN), the same in every lane; the lane that holds it comes from a vote afterwards
__device__ __forceinline__ double wave_max_f64(double v) {
    const double NI = -__builtin_huge_val();
    v = fmax_nn(v, dpp_f64<0xb1>(NI, v));             // quad_perm [1, 0, 3, 2]
    v = fmax_nn(v, dpp_f64<0x4e>(NI, v));             // quad_perm [2, 3, 0, 1]
    v = fmax_nn(v, dpp_f64<0x141>(NI, v));            // row_half_mirror
    v = fmax_nn(v, dpp_f64<0x140>(NI, v));            // row_mirror: every lane holds its row's maximum
    v = fmax_nn(v, dpp_f64<0x142, 0xa>(NI, v));       // row_bcast15 into rows 1 and 3
    v = fmax_nn(v, dpp_f64<0x143, 0xc>(NI, v));       // row_bcast31 into rows 2 and 3: lane 63 holds the maximum
    return rl_f64(v, 63);
}
// inclusive prefix maximum (lower lanes first), values only
__device__ __forceinline__ double wave_prefix_max_f64(double v, const int lane) {
    const double NI = -__builtin_huge_val();
    v = fmax_nn(v, dpp_f64<0x111>(NI, v));            // row_shr:1, 2, 4, 8: the scan inside each row of sixteen lanes
    v = fmax_nn(v, dpp_f64<0x112>(NI, v));
    v = fmax_nn(v, dpp_f64<0x114>(NI, v));
    v = fmax_nn(v, dpp_f64<0x118>(NI, v));
    v = fmax_nn(v, dpp_f64<0x142, 0xa>(NI, v));       // rows 1 and 3 take the total of the row before
    v = fmax_nn(v, dpp_f64<0x143, 0xc>(NI, v));       // rows 2 and 3 take the total of rows 0 and 1
    return v;
}
// wave-wide minimum of an int, the same in every lane
__device__ __forceinline__ int wave_min_i32(int v) {
    const int BIG = 0x7fffffff;
    v = min(v, dpp_i32<0xb1>(BIG, v));
    v = min(v, dpp_i32<0x4e>(BIG, v));
    v = min(v, dpp_i32<0x141>(BIG, v));
    v = min(v, dpp_i32<0x140>(BIG, v));
    v = min(v, dpp_i32<0x142, 0xa>(BIG, v));
    v = min(v, dpp_i32<0x143, 0xc>(BIG, v));
    return rl_i32(v, 63);
}

struct WavePtrs {
    const int32_t* __restrict__ ndx; const int32_t* __restrict__ stopv; const uint8_t* __restrict__ kf;
    const int32_t* __restrict__ lo; const int32_t* __restrict__ q1; const int32_t* __restrict__ q2;
    const double* __restrict__ cs; const DpwExt* __restrict__ ext;
    const int32_t* __restrict__ srank;      // or nullptr: the extras of node i are ext[i]; else ext[srank[i]]
    double* score; int32_t* traceb; int32_t* tbn; int8_t* ov; double* sfxv; int32_t* sfxi;
};

__device__ __forceinline__ void load_ext(const DpwExt* __restrict__ e, DpwT& T) {
    // one 64-byte record: four 16-byte loads
    const int4* p = reinterpret_cast<const int4*>(e);
    const int4 a = p[0], b = p[1], c = p[2], d = p[3];
    T.x0 = __hiloint2double(a.y, a.x); T.x1 = __hiloint2double(a.w, a.z); T.x2 = __hiloint2double(b.y, b.x);
    T.dlo0 = b.z; T.dlo1 = b.w; T.dlo2 = c.x; T.dhi0 = c.y; T.dhi1 = c.z; T.dhi2 = c.w;
    T.cq0 = d.x; T.cq1 = d.y; T.cq2 = d.z; T.vm = d.w;
}

__device__ __forceinline__ void load_target_w(DpwT& T, int& kfb, const WavePtrs& P, const int i0, const int lane, const int n, const double negc) {
    const int i = i0 + lane;
    const bool act = i < n;
    const int ii = act ? i : n - 1;
    kfb = P.kf[ii];
    T.i = act ? i : -1;
    T.kind = act ? DPW_KIND(kfb) : -1; T.frame = DPW_FRAME(kfb);
    T.ndx = P.ndx[ii]; T.stop_val = P.stopv[ii]; T.lo = act ? P.lo[ii] : INT_MAX; T.q1 = P.q1[ii]; T.q2 = P.q2[ii];
    T.cs = P.cs[ii]; T.csd = T.cs + negc;
    T.vm = 0; T.x0 = T.x1 = T.x2 = 0.0;
    T.dlo0 = T.dlo1 = T.dlo2 = INT_MAX; T.dhi0 = T.dhi1 = T.dhi2 = INT_MIN; T.cq0 = T.cq1 = T.cq2 = DPW_NONE;
    const int er = P.srank != nullptr ? P.srank[ii] : ii;       // asked for with the other topology fields, not behind them
    if (act && (T.kind & 1)) load_ext(P.ext + er, T);
}

// a forward stop met through a chain of candidates, as a source (its operon terms are not needed towards reverse targets)
__device__ __forceinline__ DpwS load_f3_source(const WavePtrs& P, const int j, const int s_ndx) {
    DpwS S;
    S.j = j; S.kind = 1; S.frame = 0; S.ndx = s_ndx; S.stop_val = 0; S.vm = 0;
    S.tbn = P.tbn[j]; S.score = P.score[j]; S.cs = 0.0; S.x0 = S.x1 = S.x2 = 0.0;
    return S;
}

// Sources as the wave holds them: lane u of these registers is source u of a tile (a tile of finished nodes read back from
// memory for the near steps; the batch itself for the walk).
struct SrcRegs { int pack /* kind | frame << 2 | vm << 4 */, ndx, stop_val; double score, cs, x0, x1, x2; };

// Lane masks of the batch's targets that do not change from step to step.  They are wave-uniform 64-bit values, i.e. they
// live in scalar registers and a step combines them with scalar instructions; what depends on the source is one vector
// compare per condition (its result is a lane mask again).
typedef unsigned long long lanemask;
struct WaveMasks {
    lanemask act, gb, k2, k3;      // lanes with a node; gene begins (F5 or R3); reverse starts; reverse stops
    lanemask f3f0, f3f1, f3f2;     // forward stops of frame f
    lanemask r5f0, r5f1, r5f2;     // reverse starts of frame f
    lanemask r3v0, r3v1, r3v2;     // reverse stops with an overlapping start in frame f
};
__device__ __forceinline__ lanemask vote(const bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool in_mask(const lanemask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
__device__ __forceinline__ lanemask pick3m(const int f, const lanemask a, const lanemask b, const lanemask c) { return f == 0 ? a : (f == 1 ? b : c); }
__device__ __forceinline__ WaveMasks wave_masks(const DpwLT& T) {
    WaveMasks W;
    const bool act = T.i >= 0;
    const bool k1 = act && T.kind == 1, k2 = act && T.kind == 2, k3 = act && T.kind == 3;
    W.act = vote(act); W.gb = vote(act && (T.kind == 0 || T.kind == 3)); W.k2 = vote(k2); W.k3 = vote(k3);
    W.f3f0 = vote(k1 && T.frame == 0); W.f3f1 = vote(k1 && T.frame == 1); W.f3f2 = vote(k1 && T.frame == 2);
    W.r5f0 = vote(k2 && T.frame == 0); W.r5f1 = vote(k2 && T.frame == 1); W.r5f2 = vote(k2 && T.frame == 2);
    W.r3v0 = vote(k3 && (T.vm & 1)); W.r3v1 = vote(k3 && (T.vm & 2)); W.r3v2 = vote(k3 && (T.vm & 4));
    return W;
}

// One step: source u of R (chain index j) onto this lane's target; same rules as dpw_step (dpw_core.h), which the host model
// checks against the plain restatement of the reference's loop.  The source is wave-uniform, so its kind is a scalar branch and each kind reads (v_readlane)
// only the fields it needs; the step returns as soon as no lane can take the source.  `win`: lanes for which the source lies
// inside the window and before the lane's node.  key_r5: a reverse start at s_ndx precedes this gene begin when
// s_ndx < key_r5.  TBN() yields the position of the source's own traceb node, only asked for a forward stop.
template <class TBN>
__device__ __forceinline__ void wave_step(const SrcRegs& R, const int u, const int j, const lanemask win, const DpwLT& T, const WaveMasks& W,
                                          const int key_r5, DpwLane& L, const DpwModel& M, TBN tbn_of) {
    const int sp = rl_i32(R.pack, u);
    const int sk = sp & 3, sf = (sp >> 2) & 3;
    // every kind ends in the same place: `take` = the lanes that accept the source, with value `val` and tag `tag`
    lanemask take;
    double val;
    int tag = j;
    if (sk == 0) {
        // forward start: only the forward stop of its ORF (ref: _connection.h:166-174)
        lanemask ok = win & pick3m(sf, W.f3f0, W.f3f1, W.f3f2);
        if (!ok) return;
        ok &= vote(T.stop_val < rl_i32(R.ndx, u));
        if (!ok) return;
        val = rl_f64(R.score, u) + rl_f64(R.cs, u);
        take = ok & vote(val >= L.val);
    } else if (sk == 2) {
        // reverse start, a gene end: every later gene begin (ref: :125-130, 337-342)
        lanemask ok = win & W.gb;
        if (!ok) return;
        const int s_ndx = rl_i32(R.ndx, u);
        ok &= vote(s_ndx < key_r5);
        if (!ok) return;
        const double s_score = rl_f64(R.score, u);
        val = s_score + M.negc;
        const lanemask tab = ok & W.k3 & vote(T.ndx - s_ndx <= 3 * DPW_OPER_DIST);     // reverse stops nearby: the distance term
        if (tab) { if (in_mask(tab)) val = s_score + dpw_igm_apart(T.ndx - s_ndx, M.negc, M.igm); }
        take = ok & vote(val >= L.val);
    } else if (sk == 3) {
        // reverse stop: the reverse starts of its ORF; reverse stops inside its ORF, as an operon (ref: :228-235, 345-356)
        lanemask ok = win & (pick3m(sf, W.r5f0, W.r5f1, W.r5f2) | pick3m(sf, W.r3v0, W.r3v1, W.r3v2));
        if (!ok) return;
        ok &= vote(rl_i32(R.stop_val, u) > T.ndx);
        if (!ok) return;
        const double s_score = rl_f64(R.score, u);
        const bool r5t = in_mask(W.k2);
        // sf is uniform: three scalar branches instead of a register-indexed select
        // (the empty asm statements keep the branches apart: merged, they become a select over a private array in scratch)
        if (sf == 0) { val = s_score + (r5t ? T.cs : T.x0); asm volatile("; frame 0"); }
        else if (sf == 1) { val = s_score + (r5t ? T.cs : T.x1); asm volatile("; frame 1"); }
        else { val = s_score + (r5t ? T.cs : T.x2); asm volatile("; frame 2"); }
        take = ok & vote(val >= L.val);
    } else {
        // forward stop, a gene end: all four kinds (ref: :117-124, 177-188, 238-254, 288-336); dpw_step_f3 with `win` for
        // its window test
        const int s_ndx = rl_i32(R.ndx, u), s_vm = sp >> 4;
        const double s_score = rl_f64(R.score, u);
        bool ok = in_mask(win);
        double w; int ov1 = 0;
        if (T.kind == 0) {
            ok = ok && s_ndx + 2 < T.ndx;
            w = dpw_igm_apart(T.ndx - s_ndx, M.negc, M.igm);
        } else if (T.kind == 1) {
            ok = ok && T.stop_val < s_ndx && ((s_vm >> T.frame) & 1) != 0;
            w = dpw_sel3(T.frame, rl_f64(R.x0, u), rl_f64(R.x1, u), rl_f64(R.x2, u));
        } else {
            const int lhs = tbn_of() + s_ndx + 7;
            const bool c0 = (s_ndx > T.dlo0) & (s_ndx < T.dhi0) & (lhs < T.drhs0);
            if (T.kind == 2) { ok = ok && c0; w = T.csd; }
            else {
                ok = ok && s_ndx < T.okhi;
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
        val = s_score + w;
        tag = j | (ov1 << DPW_TAG_BITS);
        take = vote(ok && val >= L.val);
    }
    if (!take) return;
    if (in_mask(take)) { L.val = val; L.tag = tag; }
}


// ------------------------------------------------------------------------------------------------------------------------
// The step schedule of a group (dpw_core.h "Step schedule"): which lanes of its own 64-node batch, and of the batch behind it, a
// node reaches as the source of a pair step -- once per (contig, translation table) instead of once per step of every model's chain.
// Round 6: per NODE four 64-bit words (32 bytes: W0, W1 towards its own batch, N0, N1 towards the next one) that k_dp_wave's lanes
// load with two coalesced 16-byte reads; no lists, no slots, nothing that the walk has to fetch.  One workgroup per contig (and per
// run of 16 batches of a long one), a wavefront per four batches; batch g of the group owns the 64 records from 64 g on.  The wave of
// batch g writes the W words of its own nodes and the N words of the nodes of batch g - 1 (the near sources: from the earliest p_near
// of a gene begin of the batch on; they must all lie in the batch before -- else the batch is marked, counted in scur[1], and the
// launch falls back to k_dpw_dyn: never on sequence, forced in the tests).
// The lane masks are the tests of dpw_static_bits (dpw_core.h; the host model builds its words with that function and the two are
// compared through the kernels' results), taken apart by source kind so that a source costs a few vector compares.
__global__ void __launch_bounds__(256)
k_dpw_sched(const int32_t* __restrict__ cbase, const int32_t* __restrict__ bbase, const DpwTopoArrays ta, const int force_miss /* tests: every batch reports a miss */) {
    __shared__ int s_nd[4][64];                     // per wavefront: the positions of its batch's nodes (ascending with the lane)
    const int c = blockIdx.x, lane = threadIdx.x & 63;
    const int base = cbase[c], n = cbase[c + 1] - base;
    const int b = (blockIdx.y << 4) + (threadIdx.x >> 6) * 4;      // this wavefront's four batches: b .. b + 3
    const int32_t* __restrict__ ndx = ta.ndx + base; const int32_t* __restrict__ stopv = ta.stop_val + base;
    const uint8_t* __restrict__ kfp = ta.kf + base;
    const int bb0 = bbase[c];
    for (int bi = b; bi < b + 4 && (bi << 6) < n; bi++) {
        const int i0 = bi << 6, bg = bb0 + bi;
        const int i = i0 + lane;
        const bool act = i < n;
        const int ii = act ? i : n - 1;
        const int my_kf = kfp[ii], t_ndx = ndx[ii], t_stop = stopv[ii];
        const int t_lo = act ? ta.lo[base + ii] : INT_MAX;
        const int kind = act ? DPW_KIND(my_kf) : -1, frame = DPW_FRAME(my_kf);
        // the targets' kinds and frames as lane masks
        const lanemask k0 = vote(kind == 0), k1 = vote(kind == 1), k2 = vote(kind == 2), k3 = vote(kind == 3);
        const lanemask fr0 = vote(frame == 0), fr1 = vote(frame == 1), fr2 = vote(frame == 2);
        const lanemask gbm = k0 | k3;
        const int jm = wave_min_i32((kind == 0 || kind == 3) ? max(ta.q1[base + ii], t_lo) : i0);
        if (jm < i0 - 64 || force_miss) {
            if (lane == 0) { atomicAdd(&ta.scur[1], 1u); ta.shdr[bg] = DpwSchedHdr{DPW_SCHED_NONE, 0u, 0, 0}; }
            continue;
        }
        // per-lane constants of the position tests (dpw_st / dpw_static_bits)
        const int key_r5 = kind == 3 ? t_ndx - 2 : t_ndx;                     // a reverse start precedes this gene begin when s_ndx < key_r5
        int dlo0 = INT_MAX, dhi0 = INT_MIN;
        if (kind == 2) { dlo0 = t_stop - 4; dhi0 = min(t_stop + DPW_MAX_OPP_OVLP - 5, (t_ndx + t_stop + 4) >> 1); }
        // the words of one source (wave-uniform): r0 = every lane it reaches, by any relation (+ the forward starts a forward stop of the
        // batch pulls), r1 = the lanes whose distance term comes from the table.  `win`: the lanes whose window holds it and that lie behind it.
        auto words = [&](const int u, const int ukf, const int s_ndx, const int s_stop, const lanemask win, const bool in_batch, lanemask& r0, lanemask& r1) {
            const int sk = DPW_KIND(ukf), sf = DPW_FRAME(ukf);
            if (sk == 2) {
                r0 = win & gbm & vote(s_ndx < key_r5);
                r1 = r0 & k3 & vote(t_ndx - s_ndx <= 3 * DPW_OPER_DIST);
                // (a gene begin it does NOT reach -- a reverse stop one or two bases on, ref: _connection.h:337-342 -- goes into the second
                //  word, outside the first: the walk's shortcut for a reverse start without a second word, "every gene begin behind it", would
                //  take it; with the word the step reads its masks, and the first one keeps the lane out)
                r1 |= gbm & ~r0;
            } else if (sk == 3) {
                r0 = win & vote(s_stop > t_ndx) & ((k2 & pick3m(sf, fr0, fr1, fr2)) | k3);
                r1 = 0;
            } else {
                const lanemask m0 = win & k0 & vote(s_ndx + 2 < t_ndx);
                r1 = m0 & vote(t_ndx - s_ndx <= 3 * DPW_OPER_DIST);
                r0 = m0 | (win & k1 & vote(t_stop < s_ndx)) | (win & k2 & vote((s_ndx > dlo0) & (s_ndx < dhi0))) | (win & k3 & vote(s_ndx < t_ndx - 4));
                if (in_batch) r0 |= k0 & pick3m(sf, fr0, fr1, fr2) & ((1ull << u) - 1ull) & vote(t_ndx > s_stop);
            }
        };
        uint4* const rec = ta.sent + (size_t)bg * 128;                          // 64 records of two uint4
        // ---- the batch before as near sources: lane u = node i0 - 64 + u, from jm on
        if (bi > 0) {
            const int j = i0 - 64 + lane;
            const int s_kf = kfp[j], s_nd = ndx[j], s_sv = stopv[j];
            lanemask n0 = 0, n1 = 0;
            lanemask visit = vote(j >= jm && DPW_KIND(s_kf) != 0);
            while (visit) {
                const int u = __builtin_ctzll(visit);
                visit &= visit - 1;
                lanemask r0, r1;
                words(u, rl_i32(s_kf, u), rl_i32(s_nd, u), rl_i32(s_sv, u), vote(i0 - 64 + u >= t_lo), false, r0, r1);      // (t_lo of a lane without a node is INT_MAX)
                if (lane == u) { n0 = r0; n1 = r1; }
            }
            rec[2 * (lane - 64) + 1] = make_uint4((unsigned)n0, (unsigned)(n0 >> 32), (unsigned)n1, (unsigned)(n1 >> 32));
        }
        {
            // The batch's own nodes as sources.  Inside the batch the window test is "a later lane" (a window reaches back at least
            // MAX_NODE_DIST nodes), positions ascend with the lane, and every test of a REVERSE source is a threshold on the target's
            // position: its masks are lane ranges cut out of the kind masks.  So the reverse nodes are not visited one by one: every
            // lane finds the thresholds of its own node by binary search over the batch's positions (in LDS) and builds its own
            // words; only the forward stops (an eighth of the nodes, with masks that are no ranges) are visited.
            const lanemask own = k1 | k2 | k3, actm = k0 | own;
            int* const nd = s_nd[threadIdx.x >> 6];
            nd[lane] = act ? t_ndx : INT_MAX;
            __builtin_amdgcn_wave_barrier();
            auto rank = [&](const int x) {              // lanes whose position is <= x = the first lane whose position is > x
                int r = 0;
#pragma unroll
                for (int st = 32; st >= 1; st >>= 1) if (nd[r + st - 1] <= x) r += st;
                return r + (nd[r] <= x ? 1 : 0);
            };
            auto ge = [](const int r) -> lanemask { return r >= 64 ? 0ull : (~0ull << r); };
            auto lt = [](const int r) -> lanemask { return r >= 64 ? ~0ull : ((1ull << r) - 1ull); };
            lanemask w0 = 0, w1 = 0;                    // this lane's words as a source
            if (k2 | k3) {
                const int r1 = rank(kind == 2 ? t_ndx : t_stop - 1);
                if (kind == 3) {
                    const lanemask in_orf = (~1ull << lane) & lt(r1);          // later lanes whose position lies before the far end of the ORF
                    w0 = in_orf & ((k2 & pick3m(frame, fr0, fr1, fr2)) | k3);
                }
                if (k2) {
                    const int r2 = rank(t_ndx + 2), r3 = rank(t_ndx + 3 * DPW_OPER_DIST);
                    if (kind == 2) {
                        const lanemask far3 = k3 & ge(r2);
                        w0 = (k0 & ge(r1)) | far3;
                        w1 = far3 & lt(r3);
                        // A gene begin behind this node that it does not reach (a reverse stop one or two bases on, ref: _connection.h:337-342):
                        // the walk's shortcut for a reverse start WITHOUT a second word is "every gene begin behind it" (tools/gen_dpw_walk.py,
                        // "R5 (plain)").  Such a lane goes into the second word -- outside the first, which masks it out again on the path that
                        // reads the words.  (stress_variants seed 830022, contig 178: the one node in 10^8 of the sweeps where it decided a gene)
                        w1 |= gbm & (~1ull << lane) & ~w0;
                    }
                }
            }
            lanemask f3s = k1;
            while (f3s) {
                const int u = __builtin_ctzll(f3s);
                f3s &= f3s - 1;
                lanemask r0, r1;
                words(u, rl_i32(my_kf, u), rl_i32(t_ndx, u), rl_i32(t_stop, u), actm & (~1ull << u), true, r0, r1);
                if (lane == u) { w0 = r0; w1 = r1; }
            }
            __builtin_amdgcn_wave_barrier();            // (the next batch of this wave rewrites nd)
            rec[2 * lane] = make_uint4((unsigned)w0, (unsigned)(w0 >> 32), (unsigned)w1, (unsigned)(w1 >> 32));
        }
        {
            // the header: which lanes hold a stop node, and the rank of the first of them among the stop nodes of the contig -- a chain's
            // wavefront finds the extras of its stop nodes (dense by rank) from these two without reading the rank of every node first
            const lanemask stops = k1 | k3;
            const int r = (ta.srank != nullptr && stops) ? ta.srank[base + i0 + __builtin_ctzll(stops)] : 0;
            if (lane == 0) ta.shdr[bg] = DpwSchedHdr{0u, (uint32_t)r, (int32_t)(uint32_t)stops, (int32_t)(uint32_t)(stops >> 32)};
        }
    }
}

// OCC: wavefronts per SIMD the register budget is cut for (PGA_DPW_OCC).  k_dpw_dyn wants 103 VGPRs: 4 spills nothing, 5
// spills 16 bytes per lane, 6 spills 64 bytes (round 2, 12 500-contig launches: 4.53 / 4.06 / 3.98 ms at 1.55 / 1.9 / 3.0 x the algorithmic
// HBM bytes).  k_dp_wave, 6 250-contig launches -- round 5: 4: no scratch, 1.34 ms; 5: 12 bytes of scratch per lane, 1.19 ms; 6: 80 VGPRs and
// 72 bytes, 1.22 ms.  Round 6 (block structures, the chain's best gene end and the previous batch's x[] in LDS: 94 VGPRs wanted): 4: 1.085,
// 5: 1.103, 6: 80 VGPRs and 24 bytes of scratch, 1.057 ms -- the default.
template <int OCC>
__global__ void __launch_bounds__(64, OCC)
k_dpw_dyn(const ChainDesc* __restrict__ chains, DpwGroupPtrs groups, const double* __restrict__ g_cs, const DpwExt* __restrict__ g_ext,
          const ModelConst* __restrict__ models, DpBuffers buf, double* __restrict__ g_sfxv, int32_t* __restrict__ g_sfxi,
          const int32_t* __restrict__ order /* or nullptr: workgroup b walks chain order[b] (longest chains first) */) {
    __shared__ double s_igm[64];
    const int chain = order != nullptr ? order[blockIdx.x] : (int)blockIdx.x;
    if (chain < 0) return;                          // a filler: the per-XCD queues of the start order are not equally long
    const ChainDesc cd = chains[chain];
    const int lane = threadIdx.x;
    const int n = cd.n;
    const ModelConst* mc = &models[cd.model];
    s_igm[lane] = mc->igm[lane];
    __syncthreads();
    const double NEG_INF = -__builtin_huge_val();
    const DpwModel M{mc->st_wt, mc->negc, s_igm};
    WavePtrs P;
    {
        const DpwTopoArrays& ta = groups.g[cd.group];
        P.ndx = ta.ndx + cd.topo_off; P.stopv = ta.stop_val + cd.topo_off; P.kf = ta.kf + cd.topo_off;
        P.lo = ta.lo + cd.topo_off; P.q1 = ta.q1 + cd.topo_off; P.q2 = ta.q2 + cd.topo_off;
        P.cs = g_cs + cd.off;
        P.srank = ta.srank != nullptr ? ta.srank + cd.topo_off : nullptr;
        P.ext = g_ext + (ta.srank != nullptr ? cd.soff : cd.off);
        P.score = buf.score + cd.off; P.traceb = buf.traceb + cd.off; P.tbn = buf.tbn + cd.off; P.ov = buf.ov_mark + cd.off;
        P.sfxv = g_sfxv + cd.off; P.sfxi = g_sfxi + cd.off;
    }
    // a launch ends when its longest chain does: long chains issue first, the short ones fill their stalls
    if (n >= 2048) __builtin_amdgcn_s_setprio(3); else if (n >= 1536) __builtin_amdgcn_s_setprio(2); else if (n >= 1024) __builtin_amdgcn_s_setprio(1);
    const bool long_chain = n > 2 * DPW_MAX_NODE_DIST;         // only then can a window start past node 0
    double end_best = -1.0; int end_idx = -1, end_tb = -1;
    // block structures of `a` (see the head of the file): lane q of S1 = blocks [b-1-q, b-1], of S2 = blocks [b-2-q, b-2]
    double s1v = NEG_INF, s2v = NEG_INF, ppv = NEG_INF; int s1i = -1, s2i = -1, ppi = -1;
    // uniform carries
    double rv0 = NEG_INF, rv1 = NEG_INF, rv2 = NEG_INF; int ri0 = -1, ri1 = -1, ri2 = -1, rn0 = -1, rn1 = -1, rn2 = -1;
    int l3i0 = -1, l3i1 = -1, l3i2 = -1, l3s0 = 0, l3s1 = 0, l3s2 = 0, l3n0 = 0, l3n1 = 0, l3n2 = 0;
    double l3v0 = 0.0, l3v1 = 0.0, l3v2 = 0.0;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));      // lanes before this one

    const int nb = (n + 63) >> 6;
    // optional phase timing of chain 0 (PGA_DP_PROFILE through the scorer-level call): cycles per batch phase
    const bool prof = buf.prof != nullptr && (blockIdx.x & 63) == 0;      // one chain alone, or every 64th of a launch
    unsigned long long tp = prof ? __builtin_readcyclecounter() : 0;
    auto mark = [&](const int slot) {
        if (!prof) return;
        const unsigned long long now = __builtin_readcyclecounter();
        if (lane == 0) atomicAdd(&buf.prof[slot], now - tp);
        tp = now;
    };
    for (int b = 0; b < nb; b++) {
        const int i0 = b << 6;
        DpwT T; int kfb;
        load_target_w(T, kfb, P, i0, lane, n, M.negc);
        const DpwLT LT = dpw_lean(T);
        const WaveMasks W = wave_masks(LT);
        const int key_r5 = T.kind == 3 ? T.ndx - 2 : T.ndx;
        const bool act = T.i >= 0;
        mark(0);
        DpwLane L{0.0, -1};
        int tbn_pre = -1;                   // position of the traceb node while it is older than the batch
        // a candidate older than the batch, in any order: the lexicographic (value, index) rule spelled out
        auto take = [&](const bool ok, const double val, const int j, const int ov1, const int s_ndx) {
            const int cur = dpw_tag_index(L.tag);
            if (ok && (val > L.val || (val == L.val && j > cur))) { L.val = val; L.tag = j | (ov1 << DPW_TAG_BITS); tbn_pre = s_ndx; }
        };

        // ---- (2) near steps first (ascending sources onto an empty state: ">=" is the whole tie rule): from the earliest
        //      p_near of a gene begin up to the batch, one wave-uniform source at a time
        {
            int jm = (act && (T.kind == 0 || T.kind == 3)) ? max(T.q1, T.lo) : i0;
            jm = wave_min_i32(jm);
            for (int t0 = jm; t0 < i0; t0 += 64) {
                const int j = t0 + lane;
                const bool in = j < i0;
                const int jj = in ? j : i0 - 1;
                const int s_kf = P.kf[jj];
                const int sk = DPW_KIND(s_kf);
                SrcRegs R;
                R.ndx = P.ndx[jj]; R.stop_val = P.stopv[jj]; R.score = P.score[jj]; R.cs = P.cs[jj];
                const int s_tbn = P.tbn[jj];
                const int er = P.srank != nullptr ? P.srank[jj] : jj;
                int vm = 0; R.x0 = R.x1 = R.x2 = 0.0;
                if (sk == 1) { const DpwExt* e = P.ext + er; vm = e->vm; R.x0 = e->x[0]; R.x1 = e->x[1]; R.x2 = e->x[2]; }
                R.pack = sk | (DPW_FRAME(s_kf) << 2) | (vm << 4);
                const bool dead = (sk == 1 || sk == 2) && s_tbn == -1;
                // (forward starts before the batch take no step: all they can offer goes to the forward stop of their ORF, and the
                //  per-frame running maximum of (3) carries exactly that)
                unsigned long long visit = __ballot(in && !dead && sk != 0);
                while (visit) {
                    const int u = __builtin_ctzll(visit);
                    visit &= visit - 1;
                    const int js = t0 + u;
                    wave_step(R, u, js, W.act & vote(js >= T.lo), LT, W, key_r5, L, M, [&]() { return rl_i32(s_tbn, u); });
                }
            }
            if (L.tag >= 0) tbn_pre = P.ndx[dpw_tag_index(L.tag)];
        }
        mark(1);
        // ---- (1) gene begins: far gene ends, `a` over [lo, min(p_near, i0))
        {
            const bool gb = act && (T.kind == 0 || T.kind == 3);
            const int lo = T.lo, hi = min(T.q1, i0);
            const bool want = gb && hi > lo;
            const int rb = hi >> 6, part = hi & 63, Bl = lo >> 6, lpart = lo & 63;
            const int x = lpart ? Bl + 1 : Bl;                 // first whole block
            bool generic = want && !(rb == b || (rb == b - 1 && !(Bl >= rb && part > 0 && lpart != 0)));
            const int q = rb - 1 - x;                          // whole blocks [x, rb): lane q of S1 (rb == b) or S2 (rb == b - 1)
            const bool whole = want && !generic && x < rb;
            if (whole && q >= 64) generic = true;
            const int qs = whole ? (q & 63) : 0;
            const double w1v = __shfl(s1v, qs, 64), w2v = __shfl(s2v, qs, 64);
            const int w1i = __shfl(s1i, qs, 64), w2i = __shfl(s2i, qs, 64);
            const int ps = (part - 1) & 63;
            const double pv = __shfl(ppv, ps, 64); const int pi = __shfl(ppi, ps, 64);
            double rv = NEG_INF; int ri = -1;
            if (want && !generic) {
                if (whole) { if (rb == b) lex_max(rv, ri, w1v, w1i); else lex_max(rv, ri, w2v, w2i); }
                if (rb == b - 1 && part > 0 && (Bl < rb || lpart == 0)) lex_max(rv, ri, pv, pi);
                if (lpart != 0 && Bl < rb) lex_max(rv, ri, P.sfxv[lo], P.sfxi[lo]);
            }
            if (__any(generic)) {
                // a window narrower than its blocks, or a near zone deeper than one block: never on real sequence; pair by pair
                if (generic) {
                    rv = NEG_INF; ri = -1;
                    for (int j = lo; j < hi; j++) {
                        const int k = DPW_KIND(P.kf[j]);
                        if ((k == 1 || k == 2) && P.traceb[j] != -1) lex_max(rv, ri, P.score[j] + M.negc, j);
                    }
                }
            }
            if (ri >= 0) take(true, rv, ri, 0, P.ndx[ri]);
        }
        mark(2);
        // ---- (3) forward stops: the running maximum of their frame, for the first forward stop of the frame in the batch
        {
            const bool f3 = act && T.kind == 1;
            const unsigned long long m0 = __ballot(f3 && T.frame == 0), m1 = __ballot(f3 && T.frame == 1), m2 = __ballot(f3 && T.frame == 2);
            const unsigned long long mine = T.frame == 0 ? m0 : (T.frame == 1 ? m1 : m2);
            const double cv = dpw_sel3(T.frame, rv0, rv1, rv2);
            const int ci = dpw_sel3i(T.frame, ri0, ri1, ri2), cn = dpw_sel3i(T.frame, rn0, rn1, rn2);
            take(f3 && (mine & below) == 0ull && ci >= 0, cv, ci, 0, cn);
        }
        // ---- (4) reverse nodes: the last reverse stop of a frame before the batch (own stop of a reverse start; operon partner)
        if (act && T.kind == 2) {
            const int j = dpw_sel3i(T.frame, l3i0, l3i1, l3i2), ss = dpw_sel3i(T.frame, l3s0, l3s1, l3s2);
            take(j >= 0 && j >= T.lo && ss > T.ndx, dpw_sel3(T.frame, l3v0, l3v1, l3v2) + T.cs, j, 0, dpw_sel3i(T.frame, l3n0, l3n1, l3n2));
        } else if (act && T.kind == 3) {
            take((T.vm & 1) && l3i0 >= 0 && l3i0 >= T.lo && l3s0 > T.ndx, l3v0 + T.x0, l3i0, 0, l3n0);
            take((T.vm & 2) && l3i1 >= 0 && l3i1 >= T.lo && l3s1 > T.ndx, l3v1 + T.x1, l3i1, 0, l3n1);
            take((T.vm & 4) && l3i2 >= 0 && l3i2 >= T.lo && l3s2 > T.ndx, l3v2 + T.x2, l3i2, 0, l3n2);
        }
        mark(3);
        // ---- (5) reverse nodes: forward stops that overlap the 3' end of the gene, through the chain of forward stops
        {
            const bool r5 = act && T.kind == 2, r3 = act && T.kind == 3;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                // a reverse start has one chain (q == 0), a reverse stop one per overlapping start
                int j = DPW_NONE, bound = 0;
                if (r5 && q == 0) { j = T.q2; bound = T.stop_val + DPW_MAX_OPP_OVLP - 5; }
                // (dlo = n3s - 5; an overlapping start worth nothing has no interval and is never taken: no chain to walk)
                if (r3 && dpw_sel3i(q, T.dlo0, T.dlo1, T.dlo2) != INT_MAX) { j = dpw_sel3i(q, T.cq0, T.cq1, T.cq2); bound = dpw_sel3i(q, T.dlo0, T.dlo1, T.dlo2) + DPW_MAX_OPP_OVLP; }
                while (__any(j < i0)) {
                    if (j < i0) {
                        const int s_ndx = P.ndx[j];
                        if (s_ndx >= bound) j = DPW_NONE;
                        else {
                            const DpwS S = load_f3_source(P, j, s_ndx);
                            bool ok; double w; int mf;
                            dpw_pair(S, T, M, ok, w, mf);
                            take(ok, S.score + w, j, mf + 1, s_ndx);
                            j = P.q2[j];
                        }
                    }
                }
            }
        }
        mark(4);
        // ---- (6) the walk: lane k is final when the walk reaches source i0 + k.  Forward starts are not walked: a forward start
        //      only ever offers itself to the forward stop of its own ORF (ref: _connection.h:166-174), so that stop PULLS the
        //      starts of its ORF that sit before it in the batch when the walk reaches it -- the starts are final by then -- instead
        //      of every start taking a step of the whole wavefront (there are four starts to a stop).  (value, index) decides, ties
        //      to the larger index, as everywhere between classes of candidates.
        {
            const int kmax = min(63, n - 1 - i0);
            const bool f5 = act && T.kind == 0;
            const lanemask f5f0 = vote(f5 && T.frame == 0), f5f1 = vote(f5 && T.frame == 1), f5f2 = vote(f5 && T.frame == 2);
            lanemask todo = W.act & ~(f5f0 | f5f1 | f5f2) & ((1ull << kmax) - 1ull);
            SrcRegs R;
            R.pack = T.kind | (T.frame << 2) | (T.vm << 4); R.ndx = T.ndx; R.stop_val = T.stop_val; R.cs = T.cs; R.x0 = T.x0; R.x1 = T.x1; R.x2 = T.x2;
            // the forward starts of lane k's ORF before it in the batch, onto lane k (a forward stop of frame fk); returns its tag
            auto pull_starts = [&](const int k, const int fk, int tagk) {
                lanemask cand = pick3m(fk, f5f0, f5f1, f5f2) & ((1ull << k) - 1ull);
                if (!cand) return tagk;
                cand &= vote(T.ndx > rl_i32(T.stop_val, k));
                if (!cand) return tagk;
                double bv = rl_f64(L.val, k);
                int bi = tagk < 0 ? -1 : (tagk & DPW_TAG_MASK), bt = tagk;
                bool changed = false;
                const double offer = L.val + T.cs;                  // what each lane would offer as a forward start (one vector add)
                while (cand) {
                    const int c = __builtin_ctzll(cand);
                    cand &= cand - 1ull;
                    const double v = rl_f64(offer, c);
                    if (v > bv || (v == bv && i0 + c > bi)) { bv = v; bi = i0 + c; bt = i0 + c; changed = true; }
                }
                if (changed && lane == k) { L.val = bv; L.tag = bt; }
                return bt;
            };
            while (todo) {
                const int k = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                int tagk = rl_i32(L.tag, k);
                const int pk = rl_i32(R.pack, k);
                if ((pk & 3) == 1) tagk = pull_starts(k, (pk >> 2) & 3, tagk);
                if (tagk < 0 && ((W.gb >> k) & 1ull) == 0ull) continue;  // a gene end that was never reached connects to nothing
                R.score = L.val;                                         // lane k's value is final now
                // inside the batch the window test is "a later lane": the window reaches back at least 500 nodes
                wave_step(R, k, i0 + k, W.act & (~0ull << (k + 1)), LT, W, key_r5, L, M, [&]() {
                    const int tbk = tagk & DPW_TAG_MASK;
                    return tbk >= i0 ? rl_i32(T.ndx, tbk - i0) : rl_i32(tbn_pre, k);
                });
            }
            // the last lane is nobody's source, but a forward stop there still has its starts to take
            {
                const int pk = rl_i32(R.pack, kmax);
                if ((pk & 3) == 1) pull_starts(kmax, (pk >> 2) & 3, rl_i32(L.tag, kmax));
            }
        }
        mark(5);
        // ---- (7) the batch is final: results, block structures, carries
        DpwBest B;
        B.val = L.val; B.tb = dpw_tag_index(L.tag); B.ov = dpw_tag_ov(L.tag);
        {
            const int src = B.tb >= i0 ? B.tb - i0 : 0;
            const int nd_in = __shfl(T.ndx, src, 64);
            B.tbn = B.tb < 0 ? -1 : (B.tb >= i0 ? nd_in : tbn_pre);
        }
        if (act) {
            P.score[T.i] = B.val; P.traceb[T.i] = B.tb; P.tbn[T.i] = B.tbn; P.ov[T.i] = (int8_t)B.ov;
            if ((T.kind == 1 || T.kind == 2) && B.val >= end_best) { end_best = B.val; end_idx = T.i; end_tb = B.tb; }
        }
        const DpwOut O = dpw_outputs(T, kfb, B, M.negc);
        {
            // inclusive prefix maxima of `a` inside the block: values by a scan, the index from a vote -- lane r is a record
            // when a[r] equals its own prefix maximum, and the prefix maximum at t sits at the LAST record at or before t
            // (ties to the larger index, as the ascending ">=" of the reference)
            const bool has = O.a > NEG_INF;
            const double pv = wave_prefix_max_f64(O.a, lane);
            const lanemask rec = vote(has && O.a == pv) & (below | (1ull << lane));
            const int pi = rec ? i0 + 63 - __builtin_clzll(rec) : -1;
            ppv = pv; ppi = pi;
            const double bmv = rl_f64(pv, 63); const int bmi = rl_i32(pi, 63);
            s2v = s1v; s2i = s1i;
            double nv = dpp_f64<0x138>(NEG_INF, s1v); int ni = dpp_i32<0x138>(-1, s1i);      // wave_shr:1, lane 0 takes the empty entry
            lex_max(nv, ni, bmv, bmi);
            s1v = nv; s1i = ni;
            if (long_chain) {
                double av = O.a; int ai = has ? i0 + lane : -1;
                wave_suffix_lexmax(av, ai, lane);
                if (act) { P.sfxv[T.i] = av; P.sfxi[T.i] = ai; }
            }
        }
        {
            const bool f3 = act && T.kind == 1, r3n = act && T.kind == 3;
#pragma unroll
            for (int f = 0; f < 3; f++) {
                // forward stops of the frame restart the running maximum; what follows them in the batch joins it
                const lanemask mf = vote(f3 && T.frame == f);
                const int u = mf ? 63 - __builtin_clzll(mf) : -1;
                double v = f == 0 ? O.v0 : (f == 1 ? O.v1 : O.v2);
                if (lane <= u) v = NEG_INF;
                const lanemask some = vote(v > NEG_INF);
                double& rv = f == 0 ? rv0 : (f == 1 ? rv1 : rv2);
                int& ri = f == 0 ? ri0 : (f == 1 ? ri1 : ri2);
                int& rn = f == 0 ? rn0 : (f == 1 ? rn1 : rn2);
                if (some) {
                    const double m = wave_max_f64(v);
                    const lanemask at = some & vote(v == m);               // a later node wins a tie
                    const int wl = 63 - __builtin_clzll(at);
                    const double mv = rl_f64(v, wl);                       // the same value, in scalar registers
                    if (u >= 0 || mv >= rv) { rv = mv; ri = i0 + wl; rn = rl_i32(T.ndx, wl); }
                } else if (u >= 0) { rv = NEG_INF; ri = -1; rn = -1; }
                const lanemask mr = vote(r3n && T.frame == f);
                if (mr) {
                    const int w = 63 - __builtin_clzll(mr);
                    const int li = i0 + w, ls = rl_i32(T.stop_val, w), ln = rl_i32(T.ndx, w); const double lv = rl_f64(B.val, w);
                    if (f == 0) { l3i0 = li; l3s0 = ls; l3n0 = ln; l3v0 = lv; } else if (f == 1) { l3i1 = li; l3s1 = ls; l3n1 = ln; l3v1 = lv; }
                    else { l3i2 = li; l3s2 = ls; l3n2 = ln; l3v2 = lv; }
                }
            }
        }
        mark(6);
        if (prof && lane == 0) atomicAdd(&buf.prof[7], 1ull);
    }
    // highest score among gene ends, ties to the largest index (ref: lib.pyx:1239-1251, 1311)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ob = __shfl_xor(end_best, m, 64);
        const int oi = __shfl_xor(end_idx, m, 64), ot = __shfl_xor(end_tb, m, 64);
        if (ob > end_best || (ob == end_best && oi > end_idx)) { end_best = ob; end_idx = oi; end_tb = ot; }
    }
    if (lane == 0) {
        buf.max_index[chain] = end_idx; buf.max_score[chain] = end_idx >= 0 ? end_best : 0.0;
        buf.ipath[chain] = (end_idx >= 0 && end_tb != -1) ? end_idx : -1;
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// k_dp_wave: the same batch structure as k_dpw_dyn above, its pair steps -- the near steps (2) and the in-batch walk (6) --
// driven by the group's step schedule and written in gfx950 assembly (dpw_walk_gfx950.inc, generated by tools/gen_dpw_walk.py,
// which also states the steps; k_dpw_dyn's wave_step is the same logic in C++).  An entry arrives through scalar loads and
// carries the lanes the source can reach; a step is the source's value (v_readlane), one add, one compare under the entry's mask.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) u32x4 k_uint4;

// Round 6: the wave-uniform carries of a chain live in LDS (a workgroup is one wavefront), not in scalar registers.  The assembly of
// the pair steps takes most of the scalar register file, so the compiler kept the 27 registers of these carries -- and the pointers
// next to them -- in lanes of a spill VGPR: a v_writelane / v_readlane each per batch, 350 of the 2 240 vector instructions of the
// loop.  From LDS a lane reads the record of ITS frame with one ds_read_b128 (it was three v_readlane and a chain of selects), and
// the lane that sets a record writes it itself.
struct alignas(16) DpwCarR { double v; int i, n; };          // frame f since its last forward stop: best offer to the next one, its index, its position
struct alignas(16) DpwCarL { double v; int i, s, n, pad; };  // the last reverse stop of frame f: its score, index, stop_val, position
// wave-wide maximum into an LDS double: every lane of EXEC offers its value (the compiler's own atomics become a loop of
// v_readlane over the lanes when the address is uniform)
__device__ __forceinline__ void lds_fmax_f64(double* p, const double v) {
    asm volatile("ds_max_f64 %0, %1" :: "v"((unsigned)(uintptr_t)p), "v"(v) : "memory");
}
// v_max_f64 without the canonicalising v_max_f64 x, x, x the compiler puts in front of fmax's operands (never a NaN here)
__device__ __forceinline__ double vmax_f64(const double a, const double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double wave_prefix_max_f64_lean(double v) {
    const double NI = -__builtin_huge_val();
    v = vmax_f64(v, dpp_f64<0x111>(NI, v));
    v = vmax_f64(v, dpp_f64<0x112>(NI, v));
    v = vmax_f64(v, dpp_f64<0x114>(NI, v));
    v = vmax_f64(v, dpp_f64<0x118>(NI, v));
    v = vmax_f64(v, dpp_f64<0x142, 0xa>(NI, v));
    v = vmax_f64(v, dpp_f64<0x143, 0xc>(NI, v));
    return v;
}

// SUB (round 6): the chains are SUB-CHAINS of a few long chains (dp.hip, "segments walked side by side"): sub-chain k is the nodes from
// ChainDesc::rebase (a multiple of 64) on of the chain whose topology, schedule, extras and cs it reads in place (cs from rec_off), walked
// from the empty state; its results go to its own region at `off`, in sub-chain indices (k_seg_gather shifts them back), its chain-level
// results to slot order[k].  Node indices the topology holds (window start, p_near, the first forward stop of a candidate chain) are
// shifted down on load; what lies before the sub-chain does not exist for it.  The walk of a segment is a SPECULATION that is verified
// node by node afterwards, so a candidate chain that begins before the sub-chain is simply left out.
template <int OCC, bool SUB = false>
__global__ void __launch_bounds__(64, OCC)
k_dp_wave(const ChainDesc* __restrict__ chains, DpwGroupPtrs groups, const double* __restrict__ g_cs, const DpwExt* __restrict__ g_ext,
          const ModelConst* __restrict__ models, DpBuffers buf, double* __restrict__ g_sfxv, int32_t* __restrict__ g_sfxi,
          const int32_t* __restrict__ order /* or nullptr: workgroup b walks chain order[b] (longest chains first); SUB: the result slot of sub-chain b */) {
    __shared__ double s_igm[64];
    __shared__ DpwCarR s_cr[3];                     // the carries of the chain (see DpwCarR)
    __shared__ DpwCarL s_cl[3];
    // Round 6, the register diet (101 -> under 80 vector registers: six wavefronts per SIMD instead of five).  Per-lane state that a
    // batch touches once or twice lives in LDS, 4.5 KB per wavefront: the block structures of the far gene ends (S1 / S2 as two buffers
    // that swap roles -- the new S1 is written over the old S2 --, the prefix maxima of the last block), what the forward stops of the
    // batch before offer an operon partner (x[3]), and the chain's best gene end.
    __shared__ double s_blkv[2][64]; __shared__ int s_blki[2][64];      // [p]: S1, [p ^ 1]: S2 (p flips per batch)
    __shared__ double s_ppv[64]; __shared__ int s_ppi[64];
    __shared__ double s_px[64][3];
    __shared__ double s_endv; __shared__ int s_endi[2];                 // best gene end: value; index, traceb
    const int chain = SUB ? (int)blockIdx.x : (order != nullptr ? order[blockIdx.x] : (int)blockIdx.x);
    if (chain < 0) return;                          // a filler: the per-XCD queues of the start order are not equally long
    const ChainDesc cd = chains[chain];
    const int lane = threadIdx.x;
    const int n = cd.n;
    const int rb = SUB ? cd.rebase : 0;             // the chain's node this sub-chain begins at
    auto down = [&](const int v) { return SUB ? (v == DPW_NONE || v < rb ? DPW_NONE : v - rb) : v; };      // a node index of the topology, or DPW_NONE
    const ModelConst* mc = &models[cd.model];
    s_igm[lane] = mc->igm[lane];
    const double NEG_INF = -__builtin_huge_val();
    if (lane < 3) { s_cr[lane] = DpwCarR{NEG_INF, -1, -1}; s_cl[lane] = DpwCarL{0.0, -1, 0, 0, 0}; }
    s_blkv[0][lane] = s_blkv[1][lane] = s_ppv[lane] = NEG_INF; s_blki[0][lane] = s_blki[1][lane] = s_ppi[lane] = -1;
    if (lane == 0) { s_endv = -1.0; s_endi[0] = -1; s_endi[1] = -1; }
    __syncthreads();
    const DpwModel M{mc->st_wt, mc->negc, s_igm};
    WavePtrs P;
    k_uint4* s_hdr; const uint4* g_words;
    {
        const DpwTopoArrays& ta = groups.g[cd.group];
        const int64_t to = cd.topo_off + rb;
        P.ndx = ta.ndx + to; P.stopv = ta.stop_val + to; P.kf = ta.kf + to;
        P.lo = ta.lo + to; P.q1 = ta.q1 + to; P.q2 = ta.q2 + to;
        P.cs = g_cs + ((SUB && cd.rec_off >= 0) ? cd.rec_off : cd.off);
        P.srank = ta.srank != nullptr ? ta.srank + to : nullptr;
        P.ext = g_ext + (ta.srank != nullptr ? cd.soff : cd.off);           // (SUB: by rank from the CHAIN's first stop; the schedule headers hold ranks within the contig)
        P.score = buf.score + cd.off; P.traceb = buf.traceb + cd.off; P.tbn = buf.tbn + cd.off; P.ov = buf.ov_mark + cd.off;
        P.sfxv = g_sfxv + cd.off; P.sfxi = g_sfxi + cd.off;
        const int b0 = cd.sched_b0 + (rb >> 6);
        s_hdr = (k_uint4*)(ta.shdr) + b0;
        g_words = ta.sent + (size_t)b0 * (2 * DPW_SCHED_STRIDE) + 2 * lane;       // this lane's record of batch 0: {W0, W1}, {N0, N1}
    }
    // a launch ends when its longest chain does: long chains issue first, the short ones fill their stalls
    if (n >= 2048) __builtin_amdgcn_s_setprio(3); else if (n >= 1536) __builtin_amdgcn_s_setprio(2); else if (n >= 1024) __builtin_amdgcn_s_setprio(1);
    const bool long_chain = n > 2 * DPW_MAX_NODE_DIST;         // only then can a window start past node 0
    // lanes of a mask before this one, counted (v_mbcnt): what `mask & below` was for
    auto before = [&](const lanemask m) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); };
    // the batch before, as the source of this batch's near steps: lane l held node i0 - 64 + l then, and keeps what that node offers
    // (the x[] of a forward stop, which only an operon step asks for, in LDS: s_px)
    double p_ns = NEG_INF; int p_tbn = -1, p_kinfo = 0x80, p_ndx = 0;

    const int nb = (n + 63) >> 6;
    const bool prof = buf.prof != nullptr && (blockIdx.x & 63) == 0;
    unsigned long long tp = prof ? __builtin_readcyclecounter() : 0;
    auto mark = [&](const int slot) {
        if (!prof) return;
        const unsigned long long now = __builtin_readcyclecounter();
        if (lane == 0) atomicAdd(&buf.prof[slot], now - tp);
        tp = now;
    };
    for (int b = 0; b < nb; b++) {
        const int i0 = b << 6;
        const u32x4 hdr = s_hdr[b];                  // {0 or DPW_SCHED_NONE, rank of the batch's first stop node, lanes that hold a stop node (lo, hi)}
        // the batch's words of the step schedule (dpw_core.h): this lane's node as a source towards its own batch, and the node this lane
        // held one batch ago towards this batch
        uint4 nw = make_uint4(0u, 0u, 0u, 0u);
        if (b > 0) nw = g_words[(size_t)(b - 1) * (2 * DPW_SCHED_STRIDE) + 1];
        DpwT T; int kfb;
        {
            // load_target_w with the extras' index from the header: rank of the first stop + the stop lanes before this one (the loads of the
            // extras then go out with the topology loads, not behind the rank's)
            const int i = i0 + lane;
            const bool in = i < n;
            const int ii = in ? i : n - 1;
            const lanemask stops = ((lanemask)hdr.w << 32) | hdr.z;
            const int er = P.srank != nullptr ? (int)hdr.y + before(stops) : ii;
            kfb = P.kf[ii];
            T.i = in ? i : -1;
            T.ndx = P.ndx[ii]; T.stop_val = P.stopv[ii]; T.lo = in ? P.lo[ii] : INT_MAX; T.q1 = P.q1[ii]; T.q2 = P.q2[ii];
            T.cs = P.cs[ii];
            T.vm = 0; T.x0 = T.x1 = T.x2 = 0.0;
            T.dlo0 = T.dlo1 = T.dlo2 = INT_MAX; T.dhi0 = T.dhi1 = T.dhi2 = INT_MIN; T.cq0 = T.cq1 = T.cq2 = DPW_NONE;
            if (in && ((stops >> lane) & 1ull)) load_ext(P.ext + er, T);
            if (SUB) {
                if (in) T.lo = max(T.lo - rb, 0);
                T.q1 = max(T.q1 - rb, 0); T.q2 = down(T.q2); T.cq0 = down(T.cq0); T.cq1 = down(T.cq1); T.cq2 = down(T.cq2);
            }
            if (hdr.x == DPW_SCHED_NONE) {            // the batch's near sources reach past the batch before: the host repeats the launch with k_dpw_dyn
                // (SUB: no repeat -- the rest of the sub-chain claims nothing, the verification rejects it and the chain is repaired or walked serially)
                if (SUB) for (int k = i0 + lane; k < n; k += 64) P.traceb[k] = -1;
                return;
            }
            T.kind = in ? DPW_KIND(kfb) : -1; T.frame = DPW_FRAME(kfb);
            T.csd = 0.0;
        }
        const DpwLT LT = dpw_lean(T);
        const bool act = T.i >= 0;
        // (a forward start keeps cs in the x[] of its frame: what it offers the forward stop of its ORF is score + x[frame], as a forward
        //  stop's offer to an operon partner of frame f is score + x[f] -- one form for the carries of (7); no step reads a start's x[])
        if (T.kind == 0) { if (T.frame == 0) T.x0 = T.cs; else if (T.frame == 1) T.x1 = T.cs; else T.x2 = T.cs; }
        // what the assembly's lane masks come from: kind | frame << 2 | vm << 8, 0x80 for a lane without a node (bits 12 .. 14, for the
        // carries of (7): the node lies in the ORF of the next forward stop of frame f)
        const int kinfo = act ? (T.kind | (T.frame << 2) | (T.vm << 8) | ((kfb >> 4) << 12)) : 0x80;
        mark(0);
        DpwLane L{0.0, -1};
        int tbn_pre = -1;                   // position of the traceb node while it is older than the batch
        auto take = [&](const bool ok, const double val, const int j, const int ov1, const int s_ndx) {
            const int cur = dpw_tag_index(L.tag);
            if (ok && (val > L.val || (val == L.val && j > cur))) { L.val = val; L.tag = j | (ov1 << DPW_TAG_BITS); tbn_pre = s_ndx; }
        };
        // operands of the assembly blocks (names fixed by tools/gen_dpw_walk.py)
        double& a_lv = L.val; int& a_lt = L.tag;
        const double a_x0 = T.x0, a_x1 = T.x1, a_x2 = T.x2, a_cs = T.cs, a_negc = M.negc;
        const int a_ndx = T.ndx, a_i0 = i0, a_kinfo = kinfo;
        const int a_drhs0 = LT.drhs0, a_drhs1 = LT.drhs1, a_drhs2 = LT.drhs2, a_dlo0 = LT.dlo0, a_dlo1 = LT.dlo1, a_dlo2 = LT.dlo2,
                  a_dhi0 = LT.dhi0, a_dhi1 = LT.dhi1, a_dhi2 = LT.dhi2;
        const unsigned a_igmb = (unsigned)(uintptr_t)s_igm;
        // ---- (2) near steps first (ascending sources onto an empty state: ">=" is the whole tie rule): the nodes of the batch before
        //      that the schedule lists for this batch, their values from the registers this wave left them in
        {
            const unsigned long long a_w0 = ((unsigned long long)nw.y << 32) | nw.x, a_w1 = ((unsigned long long)nw.w << 32) | nw.z;
            const double a_ns = p_ns;
            const int a_nb = p_tbn, a_pk = p_kinfo, a_pndx = p_ndx;
            const unsigned a_pxb = (unsigned)(uintptr_t)&s_px[0][0];
            DPW_ASM_NEAR();
            // the position of the traceb node: a node of the batch before, i.e. a lane of p_ndx
            const int src = L.tag >= 0 ? (dpw_tag_index(L.tag) & 63) : 0;
            const int nd_prev = __shfl(p_ndx, src, 64);
            if (L.tag >= 0) tbn_pre = nd_prev;
        }
        // (asked for here, not with the other loads of the batch: the words of the near steps and these share the assembly's registers)
        const uint4 ww = g_words[(size_t)b * (2 * DPW_SCHED_STRIDE)];
        mark(1);
        // ---- (1) gene begins: far gene ends, `a` over [lo, min(p_near, i0))
        int far_ri = -1, far_nd = 0;
        {
            const bool gb = act && (T.kind == 0 || T.kind == 3);
            const int lo = T.lo, hi = min(T.q1, i0);
            const bool want = gb && hi > lo;
            const int rb = hi >> 6, part = hi & 63, Bl = lo >> 6, lpart = lo & 63;
            const int x = lpart ? Bl + 1 : Bl;                 // first whole block
            bool generic = want && !(rb == b || (rb == b - 1 && !(Bl >= rb && part > 0 && lpart != 0)));
            const int q = rb - 1 - x;                          // whole blocks [x, rb): lane q of S1 (rb == b) or S2 (rb == b - 1)
            const bool whole = want && !generic && x < rb;
            if (whole && q >= 64) generic = true;
            const int qs = whole ? (q & 63) : 0;
            const int pb = b & 1;                              // S1 sits in buffer pb, S2 in the other one
            const int wb = rb == b ? pb : pb ^ 1;
            const double wv = s_blkv[wb][qs]; const int wi = s_blki[wb][qs];
            const int ps = (part - 1) & 63;
            const double pv = s_ppv[ps]; const int pi = s_ppi[ps];
            double rv = NEG_INF; int ri = -1;
            if (want && !generic) {
                if (whole) lex_max(rv, ri, wv, wi);
                if (rb == b - 1 && part > 0 && (Bl < rb || lpart == 0)) lex_max(rv, ri, pv, pi);
                if (lpart != 0 && Bl < rb) lex_max(rv, ri, P.sfxv[lo], P.sfxi[lo]);
            }
            if (__any(generic)) {
                if (generic) {
                    rv = NEG_INF; ri = -1;
                    for (int j = lo; j < hi; j++) {
                        const int k = DPW_KIND(P.kf[j]);
                        if ((k == 1 || k == 2) && P.traceb[j] != -1) lex_max(rv, ri, P.score[j] + M.negc, j);
                    }
                }
            }
            // (the position of that node -- a gather from memory -- is asked for here and used behind (5), if the node is still the lane's
            //  traceb then: nothing waits for it in between)
            far_ri = ri;
            if (ri >= 0) { far_nd = P.ndx[ri]; take(true, rv, ri, 0, -2); }
        }
        mark(2);
        // ---- (3) forward stops: the running maximum of their frame, for the first forward stop of the frame in the batch
        const bool f3 = act && T.kind == 1;
        const lanemask f3f0 = vote(f3 && T.frame == 0), f3f1 = vote(f3 && T.frame == 1), f3f2 = vote(f3 && T.frame == 2);
        {
            const lanemask mine = T.frame == 0 ? f3f0 : (T.frame == 1 ? f3f1 : f3f2);
            const DpwCarR r = s_cr[T.frame];                 // every lane the record of its own frame
            take(f3 && before(mine) == 0 && r.i >= 0, r.v, r.i, 0, r.n);
        }
        // ---- (4) reverse nodes: the last reverse stop of a frame before the batch (own stop of a reverse start; operon partner)
        if (act && T.kind == 2) {
            const DpwCarL l = s_cl[T.frame];
            take(l.i >= 0 && l.i >= T.lo && l.s > T.ndx, l.v + T.cs, l.i, 0, l.n);
        } else if (act && T.kind == 3) {
            const DpwCarL l0 = s_cl[0], l1 = s_cl[1], l2 = s_cl[2];
            take((T.vm & 1) && l0.i >= 0 && l0.i >= T.lo && l0.s > T.ndx, l0.v + T.x0, l0.i, 0, l0.n);
            take((T.vm & 2) && l1.i >= 0 && l1.i >= T.lo && l1.s > T.ndx, l1.v + T.x1, l1.i, 0, l1.n);
            take((T.vm & 4) && l2.i >= 0 && l2.i >= T.lo && l2.s > T.ndx, l2.v + T.x2, l2.i, 0, l2.n);
        }
        mark(3);
        // ---- (5) reverse nodes: forward stops that overlap the 3' end of the gene, through the chain of forward stops.  Round 6: a
        //      candidate is priced through the chain's OWN overlapping start only, by the interval tests of dpw_lean (a pair that is
        //      admissible through another overlapping start lies on that start's chain as well, and the plain connection is what the far gene
        //      ends / the near steps hold already); a lane walks its chains one after the other in ONE loop of the wavefront.
        {
            const bool r5 = act && T.kind == 2, r3 = act && T.kind == 3;
            int q = 0, j = DPW_NONE, dlo = INT_MAX, dhi = INT_MIN, drhs = INT_MIN;
            double xq = 0.0;
            // the chain of candidate q, or the next one that exists
            auto open_chain = [&]() {
                j = DPW_NONE;
                if (r5) { if (q == 0) { j = T.q2; dlo = LT.dlo0; dhi = LT.dhi0; drhs = LT.drhs0; xq = T.cs + M.negc; } q = 3; }
                else if (r3) {
                    while (q < 3 && j == DPW_NONE) {
                        const int lo_q = dpw_sel3i(q, LT.dlo0, LT.dlo1, LT.dlo2);
                        if (lo_q != INT_MAX) {          // (an overlapping start worth nothing is never taken: dpw_lean leaves its interval empty)
                            j = dpw_sel3i(q, T.cq0, T.cq1, T.cq2); dlo = lo_q; dhi = dpw_sel3i(q, LT.dhi0, LT.dhi1, LT.dhi2);
                            drhs = dpw_sel3i(q, LT.drhs0, LT.drhs1, LT.drhs2); xq = dpw_sel3(q, T.x0, T.x1, T.x2);
                        }
                        q++;
                    }
                } else q = 3;
            };
            open_chain();
            while (__any(j < i0 || q < 3)) {
                if (j < i0) {
                    // (the four loads of a hop go out together: a hop is one round trip, whether the chain ends here or not)
                    const int s_ndx = P.ndx[j], tbj = P.tbn[j], nj = down(P.q2[j]);
                    const double sj = P.score[j];
                    // a reverse start's chain ends at stop_val + MAX_OPP_OVLP - 5 = dlo0 + MAX_OPP_OVLP - 1, a reverse stop's at n3s + MAX_OPP_OVLP - 5 = dlo + MAX_OPP_OVLP
                    if (s_ndx >= dlo + DPW_MAX_OPP_OVLP - (r5 ? 1 : 0)) j = DPW_NONE;
                    else {
                        const bool ok = (j >= T.lo) & (tbj != -1) & (s_ndx > dlo) & (s_ndx < dhi) & (tbj + s_ndx + 7 < drhs);
                        take(ok, sj + xq, j, r5 ? 0 : q, s_ndx);          // (q was advanced when the chain was opened: ov_mark + 1)
                        j = nj;
                    }
                }
                if (j >= i0 && q < 3) open_chain();
            }
        }
        if (far_ri >= 0 && L.tag >= 0 && dpw_tag_index(L.tag) == far_ri) tbn_pre = far_nd;
        mark(4);
        // ---- (6) the walk, from the schedule: lane k is final when the walk reaches its entry.  A forward stop first pulls the
        //      forward starts of its ORF that sit before it in the batch (the entry's `pull` lanes; they are final by then).
        {
            const int a_tbnpre = tbn_pre;
            const unsigned long long a_w0 = ((unsigned long long)ww.y << 32) | ww.x, a_w1 = ((unsigned long long)ww.w << 32) | ww.z;
            double a_sv = L.tag >= 0 ? L.val : NEG_INF;      // what a lane offers as a gene-end source: -inf while it was never reached
            DPW_ASM_WALK();
        }
        mark(5);
        // ---- (7) the batch is final: results, block structures, carries
        // (the lane number as the compiler cannot see through: the LDS addresses it makes of `lane` are loop-invariant, it computed them once in
        //  front of the loop and -- out of registers at six wavefronts per SIMD -- kept them in scratch: five reloads per batch, 0.5 GB of
        //  HBM reads per launch; one shift-or each, made here, instead)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const bool alive = L.tag >= 0;
        const int tb = alive ? (L.tag & DPW_TAG_MASK) : -1;
        {
            const int src = tb >= i0 ? tb - i0 : 0;
            const int nd_in = __shfl(T.ndx, src, 64);
            const int tbn = !alive ? -1 : (tb >= i0 ? nd_in : tbn_pre);
            // what this node offers the next batch as a near source (a gene end that was never reached: -inf, which no lane takes)
            p_ns = (!alive && (T.kind == 1 || T.kind == 2)) ? NEG_INF : L.val; p_tbn = tbn; p_kinfo = kinfo; p_ndx = T.ndx;
            if (act && T.kind == 1) { s_px[ln][0] = T.x0; s_px[ln][1] = T.x1; s_px[ln][2] = T.x2; }
            if (act) {
                P.score[T.i] = L.val; P.traceb[T.i] = tb; P.tbn[T.i] = tbn; P.ov[T.i] = (int8_t)dpw_tag_ov(L.tag);
            }
            // the chain's best gene end so far (ref: lib.pyx:1239-1251: the highest score, ties to the largest index): one ds_max_f64 of the
            // batch's gene ends; the last lane that holds the maximum -- if any does: an earlier batch's may be larger -- writes its index
            const lanemask ge = vote(act && (T.kind == 1 || T.kind == 2));
            if (ge) {
                if (in_mask(ge)) lds_fmax_f64(&s_endv, L.val);
                const double m = s_endv;
                const lanemask at = ge & vote(L.val == m);
                if (at) { const int wl = 63 - __builtin_clzll(at); if (lane == wl) { s_endv = L.val; s_endi[0] = T.i; s_endi[1] = tb; } }
            }
        }
        {
            // `a`: what a reached gene end offers the far gene begins behind it (dpw_outputs).  Inclusive prefix maxima inside the
            // block: values by a scan, the index from a vote -- lane r is a record when a[r] equals its own prefix maximum, and the
            // prefix maximum at t sits at the LAST record at or before t (ties to the larger index, as the ascending ">=")
            const bool has = act && alive && (T.kind == 1 || T.kind == 2);
            const double av = has ? L.val + M.negc : NEG_INF;
            const double pv = wave_prefix_max_f64_lean(av);
            const lanemask upto = (2ull << lane) - 1ull;        // the lanes up to this one
            const lanemask rec = vote(has && av == pv) & upto;
            const int pi = rec ? i0 + 63 - __builtin_clzll(rec) : -1;
            s_ppv[ln] = pv; s_ppi[ln] = pi;
            const double bmv = rl_f64(pv, 63); const int bmi = rl_i32(pi, 63);
            // S2 := S1, S1 := S1 moved up a lane with the block's maximum merged in: the new S1 goes where S2 was, the buffers swap roles
            const int pb = b & 1;
            double nv = ln > 0 ? s_blkv[pb][(ln - 1) & 63] : NEG_INF; int ni = ln > 0 ? s_blki[pb][(ln - 1) & 63] : -1;
            lex_max(nv, ni, bmv, bmi);
            s_blkv[pb ^ 1][ln] = nv; s_blki[pb ^ 1][ln] = ni;
            if (long_chain) {
                double sv = av; int si = has ? i0 + lane : -1;
                wave_suffix_lexmax(sv, si, lane);
                if (act) { P.sfxv[T.i] = sv; P.sfxi[T.i] = si; }
            }
        }
        {
            // The carries (LDS).  What a finished lane offers the next forward stop of frame f: a forward start of that frame its score +
            // cs (x[frame] of a forward start holds cs since the load); a reached forward stop inside that stop's ORF its score + x[f]
            // where it has an overlapping start of frame f (dpw_outputs v0 .. v2).  A forward stop of the frame restarts the running
            // maximum; what follows it in the batch joins it; ties go to the later node.  The maximum itself is one ds_max_f64 over the
            // offering lanes (it was six rounds of cross-lane moves per frame), the lane that holds it then writes the record.
            const int vout = !act ? 0 : (T.kind == 0 ? (1 << T.frame) : ((T.kind == 1 && alive) ? (T.vm & (kinfo >> 12)) : 0));
            const bool r3n = act && T.kind == 3;
#pragma unroll
            for (int f = 0; f < 3; f++) {
                const lanemask mf = f == 0 ? f3f0 : (f == 1 ? f3f1 : f3f2);
                lanemask cand = vote((vout >> f) & 1);
                if (mf) {
                    const int u = 63 - __builtin_clzll(mf);
                    cand &= ~((2ull << u) - 1ull);                     // the lanes behind the frame's last forward stop
                    if (lane == 0) s_cr[f] = DpwCarR{NEG_INF, -1, -1};
                }
                if (cand) {
                    const double v = L.val + (f == 0 ? T.x0 : (f == 1 ? T.x1 : T.x2));
                    if (in_mask(cand)) lds_fmax_f64(&s_cr[f].v, v);
                    const double m = s_cr[f].v;
                    const lanemask at = cand & vote(v == m);           // (empty: an earlier batch holds the maximum, its record stays)
                    if (at) { const int wl = 63 - __builtin_clzll(at); if (lane == wl) s_cr[f] = DpwCarR{v, i0 + lane, T.ndx}; }
                }
                const lanemask mr = vote(r3n && T.frame == f);
                if (mr) { const int w = 63 - __builtin_clzll(mr); if (lane == w) s_cl[f] = DpwCarL{L.val, i0 + lane, T.stop_val, T.ndx, 0}; }
            }
        }
        mark(6);
        if (prof && lane == 0) atomicAdd(&buf.prof[7], 1ull);
    }
    // highest score among gene ends, ties to the largest index (ref: lib.pyx:1239-1251, 1311): kept in LDS as the batches went by
    const double end_best = s_endv; const int end_idx = s_endi[0], end_tb = s_endi[1];
    if (lane == 0) {
        const int slot = SUB && order != nullptr ? order[chain] : chain;
        buf.max_index[slot] = end_idx; buf.max_score[slot] = end_idx >= 0 ? end_best : 0.0;
        buf.ipath[slot] = (end_idx >= 0 && end_tb != -1) ? end_idx : -1;
    }
}

}  // namespace

void pga_launch_dpw_topo(const DpwTopoArrays& ta, const uint8_t* type, const int8_t* strand, const int32_t* d_cbase, int n_contigs, int n_nodes,
                         hipStream_t st, int max_contig_nodes) {
    if (n_nodes <= 0) return;
    const int walk = getenv("PGA_DPW_TOPO_WALK") ? atoi(getenv("PGA_DPW_TOPO_WALK")) : 0;
    // contigs that fit: a workgroup per contig on LDS copies of its node arrays (PGA_DPW_TOPO_LDS=0: the per-node kernel on global memory)
    const bool lds = max_contig_nodes > 0 && max_contig_nodes <= TOPO_LDS_NODES && !walk && !(getenv("PGA_DPW_TOPO_LDS") && atoi(getenv("PGA_DPW_TOPO_LDS")) == 0);
    if (lds) {
        // (up to 12 * TOPO_LDS_NODES + 32 bytes of dynamic LDS next to 2.3 KB of static: past the 64 KB a kernel may use unasked)
        static std::atomic<bool> attr_set[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        dev &= 63;
        if (!attr_set[dev].load()) {
            (void)hipFuncSetAttribute((const void*)k_dpw_topo_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 12 * TOPO_LDS_NODES + 32);
            attr_set[dev].store(true);
        }
        hipLaunchKernelGGL(k_dpw_topo_lds, dim3((unsigned)n_contigs), dim3(256), (size_t)12 * max_contig_nodes + 32, st, ta.ndx, ta.stop_val, type, strand, d_cbase, ta);
        return;
    }
    hipLaunchKernelGGL(k_dpw_topo, dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, st, ta.ndx, ta.stop_val, type, strand, d_cbase,
                       n_contigs, n_nodes, ta, walk ? 0 : 1);
}

void pga_launch_dpw_chain(const ChainDesc* d_chains, int n_chains, int64_t node_begin, int64_t total_nodes, const NodeArrays& nodes,
                          const DpwTopoArrays& ta, const ModelConst* d_models, const DpwBuffers& wb, hipStream_t st) {
    if (total_nodes <= 0) return;
    hipLaunchKernelGGL(k_dpw_chain, dim3((unsigned)((total_nodes + 255) / 256)), dim3(256), 0, st, d_chains, n_chains, node_begin, total_nodes,
                       nodes, ta, d_models, wb.cs, wb.ext);
}

// the sub-chains of a segmented launch (dp.hip) by the scheduled wave-batch kernel: one wavefront each, results to slot[k]
void pga_launch_dp_wave_sub(const ChainDesc* d_subs, int n_subs, const DpwGroupPtrs& groups, const ModelConst* d_models, DpBuffers buf,
                            const DpwBuffers& wb, hipStream_t st, const int32_t* d_slot) {
    if (n_subs <= 0) return;
    hipLaunchKernelGGL((k_dp_wave<6, true>), dim3((unsigned)n_subs), dim3(64), 0, st, d_subs, groups, (const double*)wb.cs, (const DpwExt*)wb.ext,
                       d_models, buf, wb.sfxv, wb.sfxi, d_slot);
}

bool pga_dpw_use_sched() {
    const char* e = getenv("PGA_DPW_SCHED");        // 0: every launch by k_dpw_dyn (cross-check)
    return e ? (atoi(e) != 0) : true;
}

void pga_launch_dpw_sched(const DpwTopoArrays& ta, const int32_t* d_cbase, const int32_t* d_bbase, int n_contigs, int max_batches, hipStream_t st) {
    // (ta.scur is cleared by the topology kernel, which always runs in front of this one)
    if (n_contigs <= 0 || max_batches <= 0) return;
    const char* fm = getenv("PGA_DPW_SCHED_MISS");
    hipLaunchKernelGGL(k_dpw_sched, dim3((unsigned)n_contigs, (unsigned)((max_batches + 15) / 16)), dim3(256), 0, st, d_cbase, d_bbase, ta, (fm && atoi(fm)) ? 1 : 0);
}

void pga_launch_dp_wave(const ChainDesc* d_chains, int n_chains, const DpwGroupPtrs& groups, const ModelConst* d_models, DpBuffers buf,
                        const DpwBuffers& wb, hipStream_t st, const int32_t* d_order, int n_blocks, bool scheduled) {
    if (n_chains <= 0) return;
    if (n_blocks <= 0 || d_order == nullptr) n_blocks = n_chains;
    static int occ = 0;
    if (!occ) { const char* e = getenv("PGA_DPW_OCC"); occ = e ? atoi(e) : 6; if (occ < 4 || occ > 6) occ = 6; }
#define DPW_LAUNCH(K) hipLaunchKernelGGL(K, dim3((unsigned)n_blocks), dim3(64), 0, st, d_chains, groups, (const double*)wb.cs, (const DpwExt*)wb.ext, \
                                         d_models, buf, wb.sfxv, wb.sfxi, d_order)
    if (scheduled) { if (occ == 5) DPW_LAUNCH(k_dp_wave<5>); else if (occ == 6) DPW_LAUNCH(k_dp_wave<6>); else DPW_LAUNCH(k_dp_wave<4>); }
    else { if (occ == 5) DPW_LAUNCH(k_dpw_dyn<5>); else if (occ == 6) DPW_LAUNCH(k_dpw_dyn<6>); else DPW_LAUNCH(k_dpw_dyn<4>); }
#undef DPW_LAUNCH
}
