// Contig-per-wavefront connection scoring for launches with many contigs: one wavefront walks one contig node by node, its
// lanes are the models scored on that contig (the chains of one translation-table group; dpc_core.h).
//
// Same recurrence as dp.hip / dp_wave.hip (ref: lib.pyx:1205-1237, _connection.h:94-408, impl/generic.h:29-36).  dp_wave.hip gives
// every (contig, model) chain a wavefront of its own and pays the topology work -- kinds, frames, windows, lane masks, the walk
// over the sources of a batch -- once per model: 87 wave instructions per node-pass, both issue pipes 70 % busy.  Here the
// topology work is done once per CONTIG, in scalar registers (the models of a contig share it), a node only runs the code of its
// own kind, every class of candidates is O(1) per node (dpc_core.h), and a launch is thousands of short waves.  A lane is a model:
// a vector instruction does useful work in as many lanes as the contig has models (four or five in a metagenome batch) -- what it
// costs is its issue slot, whatever the number of lanes.
//
// Memory.  Topology (shared by the models): 64 nodes per batch, lane = node, asked for one batch ahead; a step reads its node's
// fields with v_readlane.  Per model: cs of eight nodes x LW models per load (lane = (node, model)), staged through LDS one group
// ahead; the 64-byte extras of a stop node one stop ahead.  Results wait in the LDS history (which the near gene ends are read
// from) and leave in whole lines, 32 nodes x 2 models per store instruction.

#include "pga_internal.h"
#include "dev_common.h"
#include "dpc_core.h"

#include <algorithm>
#include <numeric>

namespace {

#define DPC_FLUSH 16        // results leave the history sixteen nodes at a time (half of it: the other half is still being read)

// (2 - d / 60) * 0.15 for d = 0 .. 60, folded at compile time with the host's double arithmetic: times st_wt it is ModelConst::igm[d]
// bit for bit (same operations in the same order, ref: _connection.h:73-75)
struct DpcT2 { double v[64]; };
constexpr DpcT2 dpc_make_t2() {
    DpcT2 t{};
    for (int d = 0; d <= DPW_OPER_DIST; d++) t.v[d] = (2.0 - ((double)d / DPW_OPER_DIST)) * 0.15;
    return t;
}
__constant__ DpcT2 c_dpc_t2 = dpc_make_t2();

typedef unsigned long long lanemask;
__device__ __forceinline__ int rfl(const int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int rl(const int v, const int lane) { return __builtin_amdgcn_readlane(v, lane); }
// a uniform value into one lane of a VGPR (v_writelane_b32 wants its lane select in m0 next to a scalar value: the select costs the same)
__device__ __forceinline__ int wl(const int val, const int lane, const int old) { return (int)threadIdx.x == lane ? val : old; }
// LDS written by one lane and read by another of the SAME wavefront: the LDS executes a wave's instructions in order, so all that is
// needed is that the compiler keeps them in order (a workgroup barrier would also wait for every outstanding global load -- the
// prefetches -- at each call)
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Sixteen bytes per active lane from memory straight into LDS (lane l's land at lds_off + 16 l), behind the compiler's back: told about
// an LDS DMA in flight, it waits for every outstanding load in front of each LDS read that follows.  The waits are written by hand
// (s_waitcnt vmcnt(N): at most N vector memory operations outstanding; they complete in the order they were issued).
__device__ __forceinline__ void lds_dma16(const void* gptr, const unsigned lds_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(lds_off) : "memory");
}
template <class T> __device__ __forceinline__ unsigned lds_offset(T* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) void*)p; }
// at most 4 k vector memory operations outstanding (k >= 4: 16)
__device__ __forceinline__ void wait_vm4(const int k) {
    if (k >= 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (k == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (k == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (k == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ lanemask bits_range(const int lo, const int hi) { return (~0ull >> (64 - hi)) & (~0ull << lo); }     // 0 <= lo < hi <= 64

typedef int v4i __attribute__((ext_vector_type(4)));
typedef const v4i __attribute__((address_space(4)))* prog_ptr;      // the compiled records: read with scalar loads

template <int LW>
struct CmX {
    int4 (*hist_)[LW];          // [DPC_HIST][LW]: {score as a source (lo, hi), position of the traceb node (forward stops), tag}
    int4 (*cand_)[LW];          // [3 * DPC_CAND][LW]: {score as a source (lo, hi), position of the traceb node, -}
    int4 (*carry_)[LW];         // [3][LW]: forward carry of each frame {v (lo, hi), i, n}
    double (*l3v_)[LW];         // [3][LW]: score of the last reverse stop of each frame
    prog_ptr prog;              // the contig's records
    int cur;                    // the node being walked
    int cidx, cndx;             // candidate lists: lane 8 f + k holds entry k of reverse frame f (uniform values parked in a VGPR)
    double st_wt;
    int ml;                     // lane % LW: this lane's model
    bool writer;                // lane < LW
    // memory (slow path; the per-lane addresses are formed there: no registers held for them)
    const int4* g_tp; const int32_t* g_srank; const double* g_cs; const DpwExt* g_ext; const double* g_score; const int32_t* g_tb; int64_t off;

    __device__ __forceinline__ DpcHist hist(const int j) const {
        const int4 v = hist_[j & (DPC_HIST - 1)][ml];
        return DpcHist{__hiloint2double(v.y, v.x), v.z};
    }
    __device__ __forceinline__ int ndx_of(const int j) const { return prog[4 * j].y; }
    __device__ __forceinline__ void hist_put(const int i, const double sv, const int tbn, const int tag) {
        if (writer) hist_[i & (DPC_HIST - 1)][ml] = make_int4(__double2loint(sv), __double2hiint(sv), tbn, tag);
    }
    __device__ __forceinline__ DpcCarry carry(const int f) const { const int4 v = carry_[f][ml]; return DpcCarry{__hiloint2double(v.y, v.x), v.z, v.w}; }
    __device__ __forceinline__ void set_carry(const int f, const DpcCarry& c) { if (writer) carry_[f][ml] = make_int4(__double2loint(c.v), __double2hiint(c.v), c.i, c.n); }
    __device__ __forceinline__ double l3v(const int f) const { return l3v_[f][ml]; }
    __device__ __forceinline__ void set_l3v(const int f, const double v) { if (writer) l3v_[f][ml] = v; }
    // any finished node as a source, from memory (slow path)
    __device__ __forceinline__ DpwS src(const int j) const {
        DpwS s;
        const int4 t0 = g_tp[2 * j], t1 = g_tp[2 * j + 1];
        s.j = j; s.kind = DPW_KIND(rfl(t1.y)); s.frame = DPW_FRAME(rfl(t1.y)); s.ndx = rfl(t0.x); s.stop_val = rfl(t0.y);
        int tb;
        if (j > cur - DPC_HIST) {
            const int4 v = hist_[j & (DPC_HIST - 1)][ml];
            tb = dpw_tag_index(v.w); s.score = v.w < 0 ? 0.0 : __hiloint2double(v.y, v.x);
        } else {
            tb = g_tb[off + j]; s.score = g_score[off + j];
        }
        s.tbn = tb == -1 ? -1 : g_tp[2 * tb].x;
        s.cs = g_cs[j];
        s.vm = 0; s.x0 = s.x1 = s.x2 = 0.0;
        if (s.kind == 1) {
            const DpwExt* e = g_ext + (g_srank != nullptr ? rfl(g_srank[j]) : j);
            s.vm = e->vm; s.x0 = e->x[0]; s.x1 = e->x[1]; s.x2 = e->x[2];
        }
        return s;
    }
    __device__ __forceinline__ void cand_put(const int f, const int k, const int idx, const int ndx, const double sv, const int tbn) {
        cidx = wl(idx, 8 * f + k, cidx);
        cndx = wl(ndx, 8 * f + k, cndx);
        if (writer) cand_[f * DPC_CAND + k][ml] = make_int4(__double2loint(sv), __double2hiint(sv), tbn, 0);
    }
    __device__ __forceinline__ int cand_idx(const int f, const int k) const { return rl(cidx, 8 * f + k); }
    __device__ __forceinline__ int cand_ndx(const int f, const int k) const { return rl(cndx, 8 * f + k); }
    __device__ __forceinline__ DpcHist cand_val(const int f, const int k) const {
        const int4 v = cand_[f * DPC_CAND + k][ml];
        return DpcHist{__hiloint2double(v.y, v.x), v.z};
    }
    __device__ __forceinline__ double igm(const int d) const { return c_dpc_t2.v[d] * st_wt; }      // (the assembly block reads the table's LDS copy)
    __device__ __forceinline__ bool any(const bool p) const { return __ballot(p) != 0ull; }
};

__device__ __forceinline__ DpcProg load_prog(prog_ptr prog, const int i) {
    DpcProg P;
    const v4i a = prog[4 * i], b = prog[4 * i + 1], c = prog[4 * i + 2], d = prog[4 * i + 3];
    P.w[0] = a.x; P.w[1] = a.y; P.w[2] = a.z; P.w[3] = a.w; P.w[4] = b.x; P.w[5] = b.y; P.w[6] = b.z; P.w[7] = b.w;
    P.w[8] = c.x; P.w[9] = c.y; P.w[10] = c.z; P.w[11] = c.w; P.w[12] = d.x; P.w[13] = d.y; P.w[14] = d.z; P.w[15] = d.w;
    return P;
}

// The topology of every contig of a group, compiled (dpc_compile_node): one wavefront per contig, lane = node, 64 nodes a pass; the
// last reverse stop of each frame before a node comes from ballots over the pass and a carry from the passes before.
__global__ void __launch_bounds__(64)
k_dpc_compile(const int32_t* __restrict__ cbase, DpwTopoArrays ta) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int b0 = cbase[c], n = cbase[c + 1] - b0;
    const int32_t* ndx = ta.ndx + b0; const int32_t* stopv = ta.stop_val + b0; const uint8_t* kf = ta.kf + b0;
    const int32_t* lo = ta.lo + b0; const int32_t* q1 = ta.q1 + b0; const int32_t* q2 = ta.q2 + b0;
    int carry0 = -1, carry1 = -1, carry2 = -1;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool act = i < n;
        const int k = act ? kf[i] : 0;
        const bool r3 = act && DPW_KIND(k) == 3;
        const lanemask m0 = __ballot(r3 && DPW_FRAME(k) == 0), m1 = __ballot(r3 && DPW_FRAME(k) == 1), m2 = __ballot(r3 && DPW_FRAME(k) == 2);
        const lanemask below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        const int l0 = (m0 & below) ? i0 + 63 - __builtin_clzll(m0 & below) : carry0;
        const int l1 = (m1 & below) ? i0 + 63 - __builtin_clzll(m1 & below) : carry1;
        const int l2 = (m2 & below) ? i0 + 63 - __builtin_clzll(m2 & below) : carry2;
        if (act) {
            DpcProg P;
            dpc_compile_node(ndx, stopv, kf, lo, q1, q2, i, l0, l1, l2, P);
            int4* out = ta.prog + 4 * ((int64_t)b0 + i);
            out[0] = make_int4(P.w[0], P.w[1], P.w[2], P.w[3]); out[1] = make_int4(P.w[4], P.w[5], P.w[6], P.w[7]);
            out[2] = make_int4(P.w[8], P.w[9], P.w[10], P.w[11]); out[3] = make_int4(P.w[12], P.w[13], P.w[14], P.w[15]);
        }
        if (m0) carry0 = i0 + 63 - __builtin_clzll(m0);
        if (m1) carry1 = i0 + 63 - __builtin_clzll(m1);
        if (m2) carry2 = i0 + 63 - __builtin_clzll(m2);
    }
}

// everything the walk keeps in LDS (the assembly block addresses it by offsets from the struct's start)
template <int LW>
struct DpcLds {
    int4 hist[DPC_HIST][LW];        // {score as a source (lo, hi), position of the traceb node (forward stops), tag}: the last DPC_HIST nodes
    int4 cand[3 * DPC_CAND][LW];    // {score as a source (lo, hi), position of the traceb node, -}: the candidate list of each reverse frame
    int4 carry[3][LW];              // forward carry of each frame {v (lo, hi), i, n}
    double l3v[3][LW];              // score of the last reverse stop of each frame
    double cs[2][64];               // cs of 64 / LW nodes x LW models, two groups: [group parity][(node pair, model)][node of the pair]
    int4 ext[4][4][LW];             // the extras of the next stop nodes: [stop rank & 3][16-byte part][lane]
    double t2[64];                  // (2 - d / 60) * 0.15, d = 0 .. 60
};

template <int LW, bool ASM>
__global__ void __launch_bounds__(64)
k_dp_contig(const int2* __restrict__ waves, const ChainDesc* __restrict__ chains, DpwGroupPtrs groups, const double* __restrict__ g_cs,
            const DpwExt* __restrict__ g_ext, const ModelConst* __restrict__ models, DpBuffers buf) {
    static_assert(LW == 8, "cs staging: 32 lanes x 16 bytes = 8 nodes x 8 models");
    static_assert(DPC_HIST == 2 * DPC_FLUSH && 64 % DPC_FLUSH == 0 && DPC_REACH == DPC_HIST, "the history holds a flush unit of unflushed results and one being read");
    constexpr int NPG = 64 / LW;                    // nodes per load of cs
    constexpr int MPF = 64 / DPC_FLUSH;             // models per store instruction of a flush
    __shared__ DpcLds<LW> L;
    static_assert(offsetof(DpcLds<LW>, hist) == 0 && offsetof(DpcLds<LW>, cand) == 4096 && offsetof(DpcLds<LW>, carry) == 6400 && offsetof(DpcLds<LW>, l3v) == 6784 &&
                  offsetof(DpcLds<LW>, cs) == 6976 && offsetof(DpcLds<LW>, ext) == 8000 && offsetof(DpcLds<LW>, t2) == 10048 && DPC_CAND == 6,
                  "tools/gen_dpc_walk.py knows this layout");
    auto& s_hist = L.hist; auto& s_cs = L.cs; auto& s_ext = L.ext;
    const int lane = threadIdx.x, ml = lane % LW;
    const int2 wv = waves[blockIdx.x];              // first chain, number of chains (<= LW)
    const int first = wv.x, count = wv.y;
    const ChainDesc cd = chains[first + min(ml, count - 1)];
    const int n = rfl(cd.n);
    if (n <= 0) {
        if (lane < count) { buf.max_index[first + lane] = -1; buf.max_score[first + lane] = 0.0; buf.ipath[first + lane] = -1; }
        return;
    }
    const int grp = rfl(cd.group);
    const int64_t toff = ((int64_t)rfl((int)(cd.topo_off >> 32)) << 32) | (uint32_t)rfl((int)cd.topo_off);
    const int4* __restrict__ tp = groups.g[grp].tp + 2 * toff;
    const prog_ptr prog = (prog_ptr)(const void*)(groups.g[grp].prog + 4 * toff);
    const bool dense = groups.g[grp].srank != nullptr;
    const ModelConst* mc = &models[cd.model];
    const double st_wt = mc->st_wt, negc = mc->negc;
    const DpwModel M{st_wt, negc, mc->igm};          // the table in memory is only read by the slow routine
    const double* __restrict__ my_cs = g_cs + cd.off;
    const DpwExt* __restrict__ my_ext = g_ext + (dense ? cd.soff : cd.off);
    const int off_lo = (int)cd.off, off_hi = (int)(cd.off >> 32);

    CmX<LW> x;
    x.hist_ = L.hist; x.cand_ = L.cand; x.carry_ = L.carry; x.l3v_ = L.l3v; x.prog = prog;
    L.t2[lane] = c_dpc_t2.v[lane];
    x.cur = 0; x.cidx = 0; x.cndx = 0;
    x.st_wt = st_wt; x.ml = ml; x.writer = lane < LW;
    x.g_tp = tp; x.g_srank = dense ? groups.g[grp].srank + toff : nullptr; x.g_cs = my_cs; x.g_ext = my_ext;
    x.g_score = buf.score; x.g_tb = buf.traceb; x.off = cd.off;

    // what this lane has asked memory for so far has to be here before the walk starts: a wait for it that the compiler leaves inside
    // the loop would also wait for the loads the loop issues behind its back (lds_dma16)
    {
        int a0 = __double2loint(st_wt), a1 = __double2hiint(st_wt), a2 = __double2loint(negc), a3 = __double2hiint(negc), a4 = off_lo, a5 = off_hi;
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    DpcRegs R;
    dpc_init(R, x);
    if (n >= 1024) __builtin_amdgcn_s_setprio(2); else if (n >= 768) __builtin_amdgcn_s_setprio(1);      // a launch ends when its longest contig does

    // cs = cscore + sscore of this lane's model, 64 / LW nodes per group, asked for one group ahead, straight into LDS: lane
    // (node pair p, model m) brings the two nodes of its pair (16 bytes; the array has two doubles of slack behind its last chain)
    auto ask_cs = [&](const int g) {
        const int k = g * NPG + 2 * (lane / LW);
        if (lane < 32 && k < n) lds_dma16(my_cs + k, lds_offset(&s_cs[g & 1][0]));
    };
    ask_cs(0);
    int cs_stop = 0;                                // stop nodes walked when the group on its way was asked for
    // The extras of the stop nodes (64 bytes per stop and model): dense records by stop rank, asked for three stops ahead, straight
    // into LDS (no registers held across the nodes in between, nothing to wait for until the stop is reached)
    int nstop = 0;                                  // stop nodes walked so far
    auto ask_ext = [&](const int rec, const int slot) {
        if (lane < LW) {
            const char* p = reinterpret_cast<const char*>(my_ext + rec);
#pragma unroll
            for (int part = 0; part < 4; part++) lds_dma16(p + 16 * part, lds_offset(&s_ext[slot][part][0]));
        }
    };
    // the record of the stop being walked: parts 0, 1 and 3 for a forward stop (x[], vm), all four for a reverse stop
    auto read_ext = [&](DpcExt& E, const int slot, const bool all) {
        // the two records asked for after this one -- four loads each -- may still be on their way
        if (dense) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int4 a = s_ext[slot][0][ml], bq = s_ext[slot][1][ml], d = s_ext[slot][3][ml];
        E.x[0] = __hiloint2double(a.y, a.x); E.x[1] = __hiloint2double(a.w, a.z); E.x[2] = __hiloint2double(bq.y, bq.x); E.vm = d.w;
        E.n3n[0] = bq.z; E.n3n[1] = bq.w; E.cq[0] = d.x; E.cq[1] = d.y; E.cq[2] = d.z;
        if (all) { const int4 c = s_ext[slot][2][ml]; E.n3n[2] = c.x; E.n3s[0] = c.y; E.n3s[1] = c.z; E.n3s[2] = c.w; }
        else { E.n3n[2] = 0; E.n3s[0] = E.n3s[1] = E.n3s[2] = 0; }
    };
    if (dense) { ask_ext(0, 0); ask_ext(1, 1); ask_ext(2, 2); }
    // results of nodes [from, to) (at most DPC_FLUSH): DPC_FLUSH nodes x MPF models per round, whole lines per store
    auto flush = [&](const int from, const int to) {
        wave_lds_order();
        const int r = lane % DPC_FLUSH, h = lane / DPC_FLUSH;
        const int node = from + r;
        for (int mm = 0; mm < count; mm += MPF) {
            const int m = min(mm + h, count - 1);
            const int ol = __shfl(off_lo, m, 64), oh = __shfl(off_hi, m, 64);
            if (mm + h < count && node < to) {
                const int4 v = s_hist[node & (DPC_HIST - 1)][m];
                const int64_t g = (((int64_t)oh << 32) | (uint32_t)ol) + node;
                buf.score[g] = v.w < 0 ? 0.0 : __hiloint2double(v.y, v.x);
                buf.traceb[g] = dpw_tag_index(v.w);
                buf.ov_mark[g] = (int8_t)dpw_tag_ov(v.w);
            }
        }
    };

    // one node with the C++ routines (dpc_core.h): any node when the assembly block is not used, the nodes it stops in front of
    // otherwise
    auto node_cpp = [&](const int i) {
        const DpcProg P = load_prog(prog, i);
        x.cur = i;
        const int kind = dpc_prog_kind(P), f = dpc_prog_frame(P);
        // A node the fast routines do not cover (rare) goes through the reference's loop over its whole window; every other node
        // through the branch of its kind, from its loads to what it leaves in the history: nothing but the loop-carried state
        // crosses from one branch to the next.  (dpc_need_slow_* are uniform by construction; the readfirstlane tells the compiler.)
        const int slot = nstop & 3;
        DpcExt E;
        bool slow;
        if (kind == 0) slow = dpc_need_slow_begin(R, P, x);
        else if (kind == 2) slow = (P.w[0] & DPC_F_SLOW) != 0;
        else if (kind == 1) { if (!dense) ask_ext(i, slot); read_ext(E, slot, false); slow = false; }
        else {
            if (!dense) ask_ext(i, slot);
            read_ext(E, slot, true);
            slow = dpc_need_slow_begin(R, P, x) || dpc_need_slow_r3(P, i, E, x);
        }
        if (kind & 1) { nstop++; if (dense) ask_ext(nstop + 2, (nstop + 2) & 3); }     // (three records of slack behind the last chain's)
        if (__builtin_expect(rfl((int)slow), 0)) {
            DpcNode N;
            N.i = i; N.kind = kind; N.frame = f; N.kfb = P.w[0] & 255; N.ndx = P.w[1]; N.stop_val = P.w[2]; N.lo = P.w[3];
            N.q1 = rfl(tp[2 * i].w); N.q2 = rfl(tp[2 * i + 1].x);
            const double cs = s_cs[(i / NPG) & 1][(((i % NPG) >> 1) * LW + ml) * 2 + (i & 1)];
            DpcOut B{0.0, -1, -1, 0.0, -1};
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // results written out by this wave are read back
            dpc_cand_slow(R, N, cs, E, M, x, B);
            {
                // what the slow routine read from memory is waited for HERE (an empty statement that uses the values): left to the point
                // where the branches meet, the wait would sit on every node's path and catch the loads in flight behind the compiler's back
                int v0 = __double2loint(B.val), v1 = __double2hiint(B.val), v2 = __double2loint(R.r5_all.v), v3 = __double2hiint(R.r5_all.v),
                    v4 = __double2loint(R.r5_far.v), v5 = __double2hiint(R.r5_far.v), v6 = __double2loint(R.f3_far.v), v7 = __double2hiint(R.f3_far.v);
                asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(B.tb), "+v"(B.ov), "+v"(R.r5_all.i),
                             "+v"(R.r5_far.i), "+v"(R.f3_far.i));
                B.val = __hiloint2double(v1, v0); R.r5_all.v = __hiloint2double(v3, v2); R.r5_far.v = __hiloint2double(v5, v4); R.f3_far.v = __hiloint2double(v7, v6);
            }
            if (kind == 0) dpc_finish_f5(P, i, cs, x.carry(f), x, B);
            else if (kind == 2) dpc_finish_r5(R, i, negc, B);
            else dpc_finish_r3(P, i, x, B);
            x.hist_put(i, B.sv, -1, B.tb < 0 ? -1 : (B.tb | ((B.ov + 1) << DPW_TAG_BITS)));
        } else if (kind == 0) {
            const double cs = s_cs[(i / NPG) & 1][(((i % NPG) >> 1) * LW + ml) * 2 + (i & 1)];
            const DpcCarry c = x.carry(f);
            DpcOut B{0.0, -1, -1, 0.0, -1};
            dpc_cand_f5(R, P, i, M, x, B);
            dpc_finish_f5(P, i, cs, c, x, B);
            x.hist_put(i, B.val, -1, B.tb);
        } else if (kind == 2) {
            const double cs = s_cs[(i / NPG) & 1][(((i % NPG) >> 1) * LW + ml) * 2 + (i & 1)];
            DpcOut B{0.0, -1, -1, 0.0, -1};
            dpc_cand_r5(P, cs, negc, x, B);
            dpc_finish_r5(R, i, negc, B);
            x.hist_put(i, B.sv, -1, B.tb);
        } else if (kind == 1) {
            DpcOut B{0.0, -1, -1, 0.0, -1};
            dpc_cand_f3(x.carry(f), B);
            dpc_finish_f3(R, P, i, E, x, B);
            x.hist_put(i, B.sv, B.tbn, B.tb);
        } else {
            DpcOut B{0.0, -1, -1, 0.0, -1};
            dpc_cand_r3(R, P, i, E, M, x, B);
            dpc_finish_r3(P, i, x, B);
            x.hist_put(i, B.val, -1, B.tb < 0 ? -1 : (B.tb | ((B.ov + 1) << DPW_TAG_BITS)));
        }
    };
    // what happens between runs of nodes: results leave every DPC_FLUSH nodes, the next group of cs is asked for every 64 / LW
    auto service = [&](const int i) {
        if ((i % DPC_FLUSH) == 0 && i > 0) flush(i - DPC_FLUSH, i);
        // this group's cs has arrived (asked for a group ago), the next group's is asked for
        // (every stop node walked since has asked for a record of extras: four loads each, all of them later than this group's)
        wait_vm4(dense ? nstop - cs_stop : 0);
        ask_cs(i / NPG + 1);
        cs_stop = nstop;
    };
    if (ASM) {
        // runs of 64 / LW nodes through the assembly block (tools/gen_dpc_walk.py); it stops in front of a node its routines do not cover
        const unsigned long long progbits = (unsigned long long)(const void*)(groups.g[grp].prog + 4 * toff);
        const unsigned long long extbits = (unsigned long long)my_ext;
        int w_i = 0, w_nstop = 0;
        int w_tid = lane, w_negc_lo = __double2loint(negc), w_negc_hi = __double2hiint(negc), w_stwt_lo = __double2loint(st_wt), w_stwt_hi = __double2hiint(st_wt);
        int w_extp_lo = (int)extbits, w_extp_hi = (int)(extbits >> 32);
        const int w_prog_lo = rfl((int)progbits), w_prog_hi = rfl((int)(progbits >> 32)), w_base = (int)lds_offset(&L);
        while (w_i < n) {
            if ((w_i % NPG) == 0) service(w_i);
            const int w_end = min(n, (w_i | (NPG - 1)) + 1);
            int w_r5a_lo = __double2loint(R.r5_all.v), w_r5a_hi = __double2hiint(R.r5_all.v), w_r5a_i = R.r5_all.i;
            int w_r5f_lo = __double2loint(R.r5_far.v), w_r5f_hi = __double2hiint(R.r5_far.v), w_r5f_i = R.r5_far.i;
            int w_f3f_lo = __double2loint(R.f3_far.v), w_f3f_hi = __double2hiint(R.f3_far.v), w_f3f_i = R.f3_far.i;
            int w_end_lo = __double2loint(R.end_best), w_end_hi = __double2hiint(R.end_best), w_end_i = R.end_idx, w_end_tb = R.end_tb;
            int w_cidx = x.cidx, w_cndx = x.cndx;
            w_nstop = nstop;
#include "dpc_walk_gfx950.inc"
            R.r5_all.v = __hiloint2double(w_r5a_hi, w_r5a_lo); R.r5_all.i = w_r5a_i;
            R.r5_far.v = __hiloint2double(w_r5f_hi, w_r5f_lo); R.r5_far.i = w_r5f_i;
            R.f3_far.v = __hiloint2double(w_f3f_hi, w_f3f_lo); R.f3_far.i = w_f3f_i;
            R.end_best = __hiloint2double(w_end_hi, w_end_lo); R.end_idx = w_end_i; R.end_tb = w_end_tb;
            x.cidx = w_cidx; x.cndx = w_cndx;
            nstop = w_nstop;
            if (w_i < w_end) {
                // a node for the slow routine (or for the fast ones, in C++, when the block's own test was the cautious one)
                node_cpp(w_i);
                w_i++;
                if (w_i < n && (w_i % NPG) == 0) { /* its group's service runs at the top of the loop */ }
            }
        }
    } else {
        for (int i = 0; i < n; i++) {
            if ((i % NPG) == 0) service(i);
            node_cpp(i);
        }
    }
    flush((n - 1) & ~(DPC_FLUSH - 1), n);
    // highest score among gene ends, ties to the largest index (ref: lib.pyx:1239-1251, 1311)
    if (lane < count) {
        const int chain = first + lane;
        buf.max_index[chain] = R.end_idx; buf.max_score[chain] = R.end_idx >= 0 ? R.end_best : 0.0;
        buf.ipath[chain] = (R.end_idx >= 0 && R.end_tb != -1) ? R.end_idx : -1;
    }
}

}  // namespace

// Waves of a launch: the chains of one (group, contig) run side by side in one wavefront, DPC_LW at most (a longer run is cut);
// chains come in (group, contig, model) order.  Longest contigs first: a launch ends when its longest wave does.
#define DPC_LW 8
void pga_dpc_plan(const ChainDesc* h, int n_chains, std::vector<int2>& waves) {
    waves.clear();
    for (int k = 0; k < n_chains;) {
        int e = k + 1;
        while (e < n_chains && e - k < DPC_LW && h[e].contig == h[k].contig && h[e].group == h[k].group && h[e].topo_off == h[k].topo_off && h[e].n == h[k].n) e++;
        waves.push_back(make_int2(k, e - k));
        k = e;
    }
    std::stable_sort(waves.begin(), waves.end(), [&](const int2& a, const int2& b) { return h[a.x].n > h[b.x].n; });
}

void pga_launch_dpc_compile(const DpwTopoArrays& ta, const int32_t* d_cbase, int n_contigs, hipStream_t st) {
    if (n_contigs <= 0 || ta.prog == nullptr) return;
    hipLaunchKernelGGL(k_dpc_compile, dim3((unsigned)n_contigs), dim3(64), 0, st, d_cbase, ta);
}

void pga_launch_dp_contig(const int2* d_waves, int n_waves, const ChainDesc* d_chains, const DpwGroupPtrs& groups, const ModelConst* d_models,
                          DpBuffers buf, const DpwBuffers& wb, hipStream_t st) {
    if (n_waves <= 0) return;
    // the assembly walk needs the extras dense by stop rank (the finder's layout); PGA_DPC_ASM=0: the C++ routines everywhere
    static int use_asm = -1;
    if (use_asm < 0) { const char* e = getenv("PGA_DPC_ASM"); use_asm = e ? atoi(e) != 0 : 1; }
    if (use_asm && groups.g[0].srank != nullptr)
        hipLaunchKernelGGL((k_dp_contig<DPC_LW, true>), dim3((unsigned)n_waves), dim3(64), 0, st, d_waves, d_chains, groups, (const double*)wb.cs,
                           (const DpwExt*)wb.ext, d_models, buf);
    else
        hipLaunchKernelGGL((k_dp_contig<DPC_LW, false>), dim3((unsigned)n_waves), dim3(64), 0, st, d_waves, d_chains, groups, (const double*)wb.cs,
                           (const DpwExt*)wb.ext, d_models, buf);
}
