// Lane-per-chain connection scoring for launches with very many chains: one LANE walks one (contig, model) chain node by node,
// 64 chains to a wavefront, all lanes of a wave at the same node index.
//
// Same recurrence as dp.hip / dp_wave.hip (ref: lib.pyx:1205-1237, _connection.h:94-408, impl/generic.h:29-36).  dp_wave.hip
// gives a chain a whole wavefront and spends about 32 vector instructions per node-pass, most of them on steps that change no
// lane (a lone source against 64 targets): it is bound by the length of its dependent instruction chains, at 8 % of the HBM
// roofline.  Here every class of candidates of a node comes from a running structure of ITS chain that costs O(1) per node
// (dpl_core.h: running maxima of the far gene ends, two small rings of the near ones, per-frame carries, static candidate
// chains), so a vector instruction does useful work in every lane: a few hundred instructions per 64 node-passes.  What it
// needs is chains -- a wavefront is only full with 64 of them, the chip with tens of thousands -- which is what a metagenome
// batch has (config 4: 4.5 chains per contig); launches with few chains keep the one-wave-per-chain kernel.
//
// Memory.  Lanes of a wave read and write at the same node index, so their records are interleaved: wave w owns
// 64 x steps(w) records, node t of lane l at base(w) + 64 t + l -- every load and store of a step is one contiguous 1 KB per
// wave.  k_dpl_pack builds the input records from the per-chain arrays (chains are dealt to lanes longest first, so the lanes
// of a wave run out of nodes together); the results stay interleaved: only the winning chain of a contig is ever read again
// (k_gather_winners, finder.hip), and k_dpl_unpack serves callers that want the plain arrays.
//   input  A  {ndx, stop_val, q2, kf}             16 B     B  {cs (f64), lo, index of the stop's extras record}       16 B
//   output    {score (f64), tag = traceb | (ov_mark + 1) << 28 or -1, position of the traceb node}     16 B
// The 64-byte extras of a stop node (dpw_core.h DpwExt) are read from the per-chain array, one step ahead of their use.
// The rings and the candidate lists live in LDS, [slot][lane] of 16 bytes: conflict-free whatever slot each lane is at.

#include "pga_internal.h"
#include "dev_common.h"
#include "dpl_core.h"

#include <algorithm>
#include <numeric>

namespace {

// (2 - d / 60) * 0.15 for d = 0 .. 60, folded at compile time with the host's double arithmetic: times st_wt it is ModelConst::igm[d]
// bit for bit (same operations in the same order, ref: _connection.h:73-75), without a per-model table in memory
struct DplT2 { double v[64]; };
constexpr DplT2 dpl_make_t2() {
    DplT2 t{};
    for (int d = 0; d <= DPW_OPER_DIST; d++) t.v[d] = (2.0 - ((double)d / DPW_OPER_DIST)) * 0.15;
    return t;
}
__constant__ DplT2 c_dpl_t2 = dpl_make_t2();

struct LaneX {
    int4 (*r5)[64]; int4 (*f3)[64]; int (*f3t)[64]; int4 (*cand)[64]; int (*candt)[64];
    const int4* A; const int4* O;      // this lane's column of the interleaved records: node j at [64 j]
    const double* t2; double st_wt;
    int lane;
    __device__ __forceinline__ DplEnt r5_get(const int s) const { const int4 v = r5[s][lane]; return DplEnt{__hiloint2double(v.y, v.x), v.z, v.w}; }
    __device__ __forceinline__ void r5_put(const int s, const DplEnt& e) { r5[s][lane] = make_int4(__double2loint(e.score), __double2hiint(e.score), e.ndx, e.idx); }
    __device__ __forceinline__ DplEnt f3_get(const int s) const { const int4 v = f3[s][lane]; return DplEnt{__hiloint2double(v.y, v.x), v.z, v.w}; }
    __device__ __forceinline__ void f3_put(const int s, const DplEnt& e) { f3[s][lane] = make_int4(__double2loint(e.score), __double2hiint(e.score), e.ndx, e.idx); }
    __device__ __forceinline__ int f3t_get(const int s) const { return f3t[s][lane]; }
    __device__ __forceinline__ void f3t_put(const int s, const int v) { f3t[s][lane] = v; }
    __device__ __forceinline__ DplCand cand_get(const int f, const int k) const {
        const int4 v = cand[f * DPL_CAND + k][lane];
        return DplCand{__hiloint2double(v.y, v.x), v.z, v.w, candt[f * DPL_CAND + k][lane]};
    }
    __device__ __forceinline__ void cand_put(const int f, const int k, const DplCand& c) {
        cand[f * DPL_CAND + k][lane] = make_int4(__double2loint(c.score), __double2hiint(c.score), c.ndx, c.idx);
        candt[f * DPL_CAND + k][lane] = c.tbn;
    }
    __device__ __forceinline__ DplFin fin(const int j) const {
        const int4 a = A[(int64_t)j * 64], o = O[(int64_t)j * 64];
        return DplFin{a.x, a.z, a.w, o.w, dpw_tag_index(o.z), __hiloint2double(o.y, o.x)};
    }
    __device__ __forceinline__ double igm(const int d) const { return t2[d] * st_wt; }
    __device__ __forceinline__ void note(int) const {}
};

// ---- input records -------------------------------------------------------------------------------------------------------
// grid (tiles of 16 node indices, waves); a thread owns four consecutive nodes of one lane's chain: 16-byte loads from the
// per-chain arrays, 16-byte stores that are contiguous over the sixteen lanes of a quarter workgroup
__global__ void __launch_bounds__(256)
k_dpl_pack(const DplDev L, const ChainDesc* __restrict__ chains, const DpwGroupPtrs groups, const double* __restrict__ g_cs, const int wave0) {
    const int w = wave0 + blockIdx.y, t0 = blockIdx.x * 16;
    if (t0 >= L.wave_steps[w]) return;
    const int c = threadIdx.x & 63, part = threadIdx.x >> 6;       // lane of the wave; which four nodes of the tile
    const int chain = L.lane_chain[(int64_t)w * 64 + c];
    const int64_t rec0 = L.wave_base[w] + c;
    const int i0 = t0 + 4 * part;
    int nd[4] = {0, 0, 0, 0}, sv[4] = {0, 0, 0, 0}, q2[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0}, kf[4] = {0, 0, 0, 0}, sr[4] = {0, 1, 2, 3};
    double cs[4] = {0.0, 0.0, 0.0, 0.0};
    int n = 0;
    if (chain >= 0) {
        const ChainDesc cd = chains[chain];
        n = cd.n;
        const DpwTopoArrays& ta = groups.g[cd.group];
        if (i0 + 3 < n) {
            int4 v;
            __builtin_memcpy(&v, ta.ndx + cd.topo_off + i0, 16); nd[0] = v.x; nd[1] = v.y; nd[2] = v.z; nd[3] = v.w;
            __builtin_memcpy(&v, ta.stop_val + cd.topo_off + i0, 16); sv[0] = v.x; sv[1] = v.y; sv[2] = v.z; sv[3] = v.w;
            __builtin_memcpy(&v, ta.q2 + cd.topo_off + i0, 16); q2[0] = v.x; q2[1] = v.y; q2[2] = v.z; q2[3] = v.w;
            __builtin_memcpy(&v, ta.lo + cd.topo_off + i0, 16); lo[0] = v.x; lo[1] = v.y; lo[2] = v.z; lo[3] = v.w;
            unsigned k4; __builtin_memcpy(&k4, ta.kf + cd.topo_off + i0, 4);
            kf[0] = k4 & 255; kf[1] = (k4 >> 8) & 255; kf[2] = (k4 >> 16) & 255; kf[3] = k4 >> 24;
            double2 d; __builtin_memcpy(&d, g_cs + cd.off + i0, 16); cs[0] = d.x; cs[1] = d.y;
            __builtin_memcpy(&d, g_cs + cd.off + i0 + 2, 16); cs[2] = d.x; cs[3] = d.y;
            if (ta.srank != nullptr) { __builtin_memcpy(&v, ta.srank + cd.topo_off + i0, 16); sr[0] = v.x; sr[1] = v.y; sr[2] = v.z; sr[3] = v.w; }
            else { sr[0] = i0; sr[1] = i0 + 1; sr[2] = i0 + 2; sr[3] = i0 + 3; }
        } else {
            for (int k = 0; k < 4; k++) {
                const int i = i0 + k;
                if (i >= n) break;
                nd[k] = ta.ndx[cd.topo_off + i]; sv[k] = ta.stop_val[cd.topo_off + i]; q2[k] = ta.q2[cd.topo_off + i]; lo[k] = ta.lo[cd.topo_off + i];
                kf[k] = ta.kf[cd.topo_off + i]; cs[k] = g_cs[cd.off + i];
                sr[k] = ta.srank != nullptr ? ta.srank[cd.topo_off + i] : i;
            }
        }
    }
    const int steps = L.wave_steps[w];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = i0 + k;
        if (i >= steps) break;
        L.inA[rec0 + (int64_t)i * 64] = make_int4(nd[k], sv[k], q2[k], kf[k]);
        L.inB[rec0 + (int64_t)i * 64] = make_int4(__double2loint(cs[k]), __double2hiint(cs[k]), lo[k], sr[k]);
    }
}

// ---- the walk ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_dp_lane(const DplDev L, const ChainDesc* __restrict__ chains, const DpwExt* __restrict__ g_ext, const ModelConst* __restrict__ models,
          DpBuffers buf) {
    __shared__ int4 s_r5[DPL_R5_RING][64];
    __shared__ int4 s_f3[DPL_F3_RING][64];
    __shared__ int s_f3t[DPL_F3_RING][64];
    __shared__ int4 s_cand[3 * DPL_CAND][64];
    __shared__ int s_candt[3 * DPL_CAND][64];
    __shared__ double s_t2[64];
    const int w = blockIdx.x, lane = threadIdx.x;
    const int chain = L.lane_chain[(int64_t)w * 64 + lane];
    const bool have = chain >= 0;
    ChainDesc cd{};
    if (have) cd = chains[chain];
    const int n = have ? cd.n : 0;
    const int steps = L.wave_steps[w];
    const int64_t rec0 = L.wave_base[w] + lane;
    const int4* __restrict__ A = L.inA + rec0;
    const int4* __restrict__ Bp = L.inB + rec0;
    int4* O = L.out + rec0;
    const DpwExt* ext = g_ext + (L.dense_ext ? cd.soff : cd.off);      // record of a stop node: ext[B.w] (its node index, or its rank among the chain's stops)
    const ModelConst* mc = &models[have ? cd.model : 0];
    const DpwModel M{mc->st_wt, mc->negc, mc->igm};
    s_t2[lane] = c_dpl_t2.v[lane];
    __syncthreads();
    LaneX x{s_r5, s_f3, s_f3t, s_cand, s_candt, A, O, s_t2, mc->st_wt, lane};
    DplState S;
    dpl_init(S);

    // software pipeline: the records of node t + 2 are asked for while node t is walked; the extras of a stop node one step ahead
    // (its topology byte is in the record that arrived a step earlier)
    const int4 zero = make_int4(0, 0, 0, 0);
    int4 a0 = steps > 0 ? A[0] : zero, b0 = steps > 0 ? Bp[0] : zero;
    int4 a1 = steps > 1 ? A[64] : zero, b1 = steps > 1 ? Bp[64] : zero;
    int4 e0 = zero, e1 = zero, e2 = zero, e3 = zero;
    if (n > 0 && (DPW_KIND(a0.w) & 1)) { const int4* p = reinterpret_cast<const int4*>(ext + b0.w); e0 = p[0]; e1 = p[1]; e2 = p[2]; e3 = p[3]; }
    for (int t = 0; t < steps; t++) {
        const int4 a = a0, b = b0;
        const int4 x0 = e0, x1 = e1, x2 = e2, x3 = e3;
        a0 = a1; b0 = b1;
        if (t + 2 < steps) { a1 = A[(int64_t)(t + 2) * 64]; b1 = Bp[(int64_t)(t + 2) * 64]; }
        if (t + 1 < n && (DPW_KIND(a0.w) & 1)) {
            const int4* p = reinterpret_cast<const int4*>(ext + b0.w);
            e0 = p[0]; e1 = p[1]; e2 = p[2]; e3 = p[3];
        }
        if (t < n) {
            DpwT T;
            const int kfb = a.w;
            T.i = t; T.kind = DPW_KIND(kfb); T.frame = DPW_FRAME(kfb); T.ndx = a.x; T.stop_val = a.y; T.q2 = a.z; T.q1 = 0;
            T.cs = __hiloint2double(b.y, b.x); T.csd = T.cs + M.negc; T.lo = b.z;
            T.vm = 0; T.x0 = T.x1 = T.x2 = 0.0;
            T.n3n0 = T.n3n1 = T.n3n2 = T.n3s0 = T.n3s1 = T.n3s2 = 0; T.cq0 = T.cq1 = T.cq2 = DPW_NONE;
            if (T.kind & 1) {
                T.x0 = __hiloint2double(x0.y, x0.x); T.x1 = __hiloint2double(x0.w, x0.z); T.x2 = __hiloint2double(x1.y, x1.x);
                T.n3n0 = x1.z; T.n3n1 = x1.w; T.n3n2 = x2.x; T.n3s0 = x2.y; T.n3s1 = x2.z; T.n3s2 = x2.w;
                T.cq0 = x3.x; T.cq1 = x3.y; T.cq2 = x3.z; T.vm = x3.w;
            }
            DpwBest B;
            dpl_step(S, T, kfb, M, x, B);
            const int tag = B.tb < 0 ? -1 : (B.tb | ((B.ov + 1) << DPW_TAG_BITS));
            O[(int64_t)t * 64] = make_int4(__double2loint(B.val), __double2hiint(B.val), tag, B.tbn);
        }
    }
    // highest score among gene ends, ties to the largest index (ref: lib.pyx:1239-1251, 1311)
    if (have) {
        buf.max_index[chain] = S.end_idx; buf.max_score[chain] = S.end_idx >= 0 ? S.end_best : 0.0;
        buf.ipath[chain] = (S.end_idx >= 0 && S.end_tb != -1) ? S.end_idx : -1;
    }
}

// the plain result arrays of every chain from the interleaved records (callers that read whole chains back)
__global__ void __launch_bounds__(256)
k_dpl_unpack(const DplDev L, const ChainDesc* __restrict__ chains, int n_chains, const int64_t* __restrict__ chain_rec, int64_t node_begin,
             int64_t total, DpBuffers buf) {
    __shared__ int s_c0;
    const int64_t blk0 = node_begin + (int64_t)blockIdx.x * blockDim.x;
    const int64_t g = blk0 + threadIdx.x;
    const bool in_range = g < node_begin + total;
    const int c = find_chain_block(chains, n_chains, blk0, in_range ? g : blk0, &s_c0);
    if (!in_range) return;
    const int i = (int)(g - chains[c].off);
    const int4 o = L.out[chain_rec[c] + (int64_t)i * 64];
    buf.score[g] = __hiloint2double(o.y, o.x); buf.traceb[g] = dpw_tag_index(o.z); buf.ov_mark[g] = (int8_t)dpw_tag_ov(o.z); buf.tbn[g] = o.w;
}

}  // namespace

void pga_dpl_plan(const ChainDesc* h, int n_chains, DplPlan& plan) {
    plan = DplPlan();
    if (n_chains <= 0) return;
    // chains are dealt to lanes longest first: the lanes of a wave run out of nodes together, and the waves that walk longest
    // start first (same length: launch order, which keeps the models of a contig side by side)
    std::vector<int32_t> order((size_t)n_chains);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](const int32_t a, const int32_t b) { return h[a].n > h[b].n; });
    const int n_waves = (n_chains + 63) / 64;
    plan.n_waves = n_waves;
    plan.lane_chain.assign((size_t)n_waves * 64, -1);
    plan.wave_base.assign((size_t)n_waves + 1, 0);
    plan.wave_steps.assign((size_t)n_waves, 0);
    plan.chain_rec.assign((size_t)n_chains, 0);
    int64_t at = 0;
    for (int w = 0; w < n_waves; w++) {
        int steps = 0;
        for (int l = 0; l < 64 && w * 64 + l < n_chains; l++) {
            const int c = order[(size_t)w * 64 + l];
            plan.lane_chain[(size_t)w * 64 + l] = c;
            plan.chain_rec[(size_t)c] = at + l;
            steps = std::max(steps, h[c].n);
        }
        plan.wave_base[(size_t)w] = at; plan.wave_steps[(size_t)w] = steps;
        plan.max_steps = std::max(plan.max_steps, steps);
        at += (int64_t)64 * steps;
    }
    plan.wave_base[(size_t)n_waves] = at;
    plan.records = at;
}

void pga_launch_dp_lane(const ChainDesc* d_chains, const DpwGroupPtrs& groups, const ModelConst* d_models, DpBuffers buf, const DpwBuffers& wb,
                        const DplDev& L0, hipStream_t st) {
    if (L0.n_waves <= 0 || L0.max_steps <= 0) return;
    DplDev L = L0;
    L.dense_ext = groups.g[0].srank != nullptr;
    // grid.y holds at most 65 535 workgroups (4.19 M chains): more waves than that are packed in slices
    for (int w0 = 0; w0 < L.n_waves; w0 += 65535)
        hipLaunchKernelGGL(k_dpl_pack, dim3((unsigned)((L.max_steps + 15) / 16), (unsigned)std::min(65535, L.n_waves - w0)), dim3(256), 0, st, L, d_chains,
                           groups, (const double*)wb.cs, w0);
    hipLaunchKernelGGL(k_dp_lane, dim3((unsigned)L.n_waves), dim3(64), 0, st, L, d_chains, (const DpwExt*)wb.ext, d_models, buf);
    // (the callers check hipGetLastError() once behind the launches of a device call; a launch that was refused shows up there)
}

void pga_launch_dpl_unpack(const ChainDesc* d_chains, int n_chains, const int64_t* d_chain_rec, int64_t node_begin, int64_t total, const DplDev& L,
                           DpBuffers buf, hipStream_t st) {
    if (total <= 0) return;
    hipLaunchKernelGGL(k_dpl_unpack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, L, d_chains, n_chains, d_chain_rec, node_begin, total, buf);
}
