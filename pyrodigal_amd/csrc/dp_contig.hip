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

template <int LW>
struct CmX {
    int4 (*hist_)[LW];          // [DPC_HIST][LW]: {score as a source (lo, hi), position of the traceb node (forward stops), tag}
    int4 (*cand_)[LW];          // [3 * DPC_CAND][LW]: {score as a source (lo, hi), position of the traceb node, -}
    int4 (*carry_)[LW];         // [3][LW]: forward carry of each frame {v (lo, hi), i, n}
    double (*l3v_)[LW];         // [3][LW]: score of the last reverse stop of each frame
    int ndx_c, ndx_p;           // positions of the current / previous batch, lane = node
    lanemask f3_c, r5_c, f3_p, r5_p;       // forward stops / reverse starts of the current / previous batch
    int i0, cur;                // first node of the current batch; the node being walked
    int cidx, cndx;             // candidate lists: lane 8 f + k holds entry k of reverse frame f (uniform values parked in a VGPR)
    int l3i_, l3s_, l3n_;       // last reverse stop of frame f -- index, stop_val, position -- in lane f (uniform values parked in VGPRs)
    double st_wt;
    int ml;                     // lane % LW: this lane's model
    bool writer;                // lane < LW
    // memory (rare paths; the per-lane addresses are formed there: no registers held for them)
    const int4* g_tp; const int32_t* g_srank; const double* g_cs; const DpwExt* g_ext; const double* g_score; const int32_t* g_tb; int64_t off;

    __device__ __forceinline__ int reach() const { return i0 >= 64 ? i0 - 64 : 0; }
    template <class F> __device__ __forceinline__ void seg(const int a, const int b, const int base, const lanemask km, const int ndxv, F f) {
        const int lo_l = max(a, base) - base, hi_l = min(b, base + 64) - base;
        if (hi_l > lo_l) {
            lanemask m = km & bits_range(lo_l, hi_l);
            while (m) {
                const int u = __builtin_ctzll(m);
                m &= m - 1ull;
                f(base + u, rl(ndxv, u));
            }
        }
    }
    // [a, b) from reach() on: in the current batch or the one before
    template <class F> __device__ __forceinline__ void for_near(const int a, const int b, const int kind, F f) {
        if (a < i0) seg(a, b, i0 - 64, kind == DPC_K_F3 ? f3_p : r5_p, ndx_p, f);
        seg(a, b, i0, kind == DPC_K_F3 ? f3_c : r5_c, ndx_c, f);
    }
    // a gene end of the last DPC_HIST nodes
    __device__ __forceinline__ DpcHist hist(const int j) const {
        const int4 v = hist_[j & (DPC_HIST - 1)][ml];
        return DpcHist{__hiloint2double(v.y, v.x), v.z};
    }
    // ... or an older one, whose results have been written out (by other lanes of this wave)
    __device__ __forceinline__ DpcHist hist_deep(const int j) const {
        if (j > cur - DPC_HIST) return hist(j);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int tb = g_tb[off + j];
        return DpcHist{tb == -1 ? -__builtin_huge_val() : g_score[off + j], tb == -1 ? -1 : g_tp[2 * tb].x};
    }
    __device__ __forceinline__ int ndx_of(const int j) const { return j >= i0 ? rl(ndx_c, j - i0) : rl(ndx_p, j - i0 + 64); }
    __device__ __forceinline__ void hist_put(const int i, const double sv, const int tbn, const int tag) {
        if (writer) hist_[i & (DPC_HIST - 1)][ml] = make_int4(__double2loint(sv), __double2hiint(sv), tbn, tag);
    }
    __device__ __forceinline__ DpcCarry carry(const int f) const { const int4 v = carry_[f][ml]; return DpcCarry{__hiloint2double(v.y, v.x), v.z, v.w}; }
    __device__ __forceinline__ void set_carry(const int f, const DpcCarry& c) { if (writer) carry_[f][ml] = make_int4(__double2loint(c.v), __double2hiint(c.v), c.i, c.n); }
    __device__ __forceinline__ int l3i(const int f) const { return rl(l3i_, f); }
    __device__ __forceinline__ int l3s(const int f) const { return rl(l3s_, f); }
    __device__ __forceinline__ int l3n(const int f) const { return rl(l3n_, f); }
    __device__ __forceinline__ void set_l3(const int f, const int i, const int s, const int n) { l3i_ = wl(i, f, l3i_); l3s_ = wl(s, f, l3s_); l3n_ = wl(n, f, l3n_); }
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
    __device__ __forceinline__ double igm(const int d) const { return c_dpc_t2.v[d] * st_wt; }
    __device__ __forceinline__ bool any(const bool p) const { return __ballot(p) != 0ull; }
};

// The same accessor for a gene begin whose unfolded range reaches back beyond the history (a dense stretch: rare): its own copy of
// the fast routines, so that the loads from memory -- and the waits for them -- stay out of the common path.
template <int LW>
struct CmXDeep : CmX<LW> {
    __device__ __forceinline__ DpcHist hist(const int j) const { return this->hist_deep(j); }
};

// packed topology of 64 nodes, lane = node: a = {ndx, stop_val, lo, q1}, b = {q2, kf}
struct TopoRegs { int4 a; int2 b; };
__device__ __forceinline__ TopoRegs load_topo(const int4* __restrict__ tp, const int i0, const int lane, const int n) {
    TopoRegs r;
    const int ii = min(i0 + lane, n - 1);
    r.a = tp[2 * ii];
    const int4 b = tp[2 * ii + 1];
    r.b = make_int2(b.x, b.y);
    return r;
}

template <int LW>
__global__ void __launch_bounds__(64)
k_dp_contig(const int2* __restrict__ waves, const ChainDesc* __restrict__ chains, DpwGroupPtrs groups, const double* __restrict__ g_cs,
            const DpwExt* __restrict__ g_ext, const ModelConst* __restrict__ models, DpBuffers buf) {
    static_assert(LW == 8, "cs staging: 32 lanes x 16 bytes = 8 nodes x 8 models");
    static_assert(DPC_HIST == 2 * DPC_FLUSH && 64 % DPC_FLUSH == 0, "the history holds a flush unit of unflushed results and one being read");
    constexpr int NPG = 64 / LW;                    // nodes per load of cs
    constexpr int MPF = 64 / DPC_FLUSH;             // models per store instruction of a flush
    __shared__ int4 s_hist[DPC_HIST][LW];
    __shared__ int4 s_cand[3 * DPC_CAND][LW];
    __shared__ int4 s_carry[3][LW];
    __shared__ double s_l3v[3][LW];
    __shared__ double s_cs[2][64];                  // cs of 64 / LW nodes x LW models, two groups: [group parity][(node pair, model)][node of the pair]
    __shared__ int4 s_ext[4][4][LW];                // the extras of the next stop nodes: [stop rank & 3][16-byte part][lane]
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
    const bool dense = groups.g[grp].srank != nullptr;
    const ModelConst* mc = &models[cd.model];
    const double st_wt = mc->st_wt, negc = mc->negc;
    const DpwModel M{st_wt, negc, mc->igm};          // the table in memory is only read by the slow routine
    const double* __restrict__ my_cs = g_cs + cd.off;
    const DpwExt* __restrict__ my_ext = g_ext + (dense ? cd.soff : cd.off);
    const int off_lo = (int)cd.off, off_hi = (int)(cd.off >> 32);

    CmXDeep<LW> xd;                                // (the deep variant only overrides hist)
    CmX<LW>& x = xd;
    x.hist_ = s_hist; x.cand_ = s_cand; x.carry_ = s_carry; x.l3v_ = s_l3v;
    x.ndx_c = 0; x.ndx_p = 0; x.f3_c = x.r5_c = x.f3_p = x.r5_p = 0ull; x.i0 = 0; x.cur = 0; x.cidx = 0; x.cndx = 0; x.l3i_ = -1; x.l3s_ = 0; x.l3n_ = 0;
    x.st_wt = st_wt; x.ml = ml; x.writer = lane < LW;
    x.g_tp = tp; x.g_srank = dense ? groups.g[grp].srank + toff : nullptr; x.g_cs = my_cs; x.g_ext = my_ext;
    x.g_score = buf.score; x.g_tb = buf.traceb; x.off = cd.off;

    DpcRegs R; DpcUni U;
    dpc_init(R, U, x);
    if (n >= 1024) __builtin_amdgcn_s_setprio(2); else if (n >= 768) __builtin_amdgcn_s_setprio(1);      // a launch ends when its longest contig does

    TopoRegs cur{}, nxt = load_topo(tp, 0, lane, n);
    // cs = cscore + sscore of this lane's model, 64 / LW nodes per group, asked for one group ahead, straight into LDS: lane
    // (node pair p, model m) brings the two nodes of its pair (16 bytes; the array has two doubles of slack behind its last chain)
    auto ask_cs = [&](const int g) {
        const int k = g * NPG + 2 * (lane / LW);
        if (lane < 32 && k < n) lds_dma16(my_cs + k, lds_offset(&s_cs[g & 1][0]));
    };
    ask_cs(0);
    int cs_stop = 0;                                // stop nodes walked when the group on its way was asked for
    // The extras of the stop nodes (64 bytes per stop and model): dense records by stop rank, asked for three stops ahead, straight
    // into LDS (global_load_lds: no registers held across the nodes in between, nothing to wait for until the stop is reached)
    int nstop = 0;                                  // stop nodes walked so far
#ifdef DPC_PROFILE
    unsigned long long pc_wait = 0, pc_cswait = 0; int pc_slow = 0, pc_deep = 0, pc_slowsrc = 0;
#endif
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
#ifdef DPC_PROFILE
        const unsigned long long w0 = __builtin_readcyclecounter();
#endif
        if (dense) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef DPC_PROFILE
        pc_wait += __builtin_readcyclecounter() - w0;
#endif
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

#ifdef DPC_PROFILE
    unsigned long long pc_cyc[5] = {0, 0, 0, 0, 0}; int pc_cnt[5] = {0, 0, 0, 0, 0};
#endif
    for (int i = 0; i < n; i++) {
#ifdef DPC_PROFILE
        const unsigned long long pc_t0 = __builtin_readcyclecounter();
#endif
        const int t = i & 63;
        if (t == 0) {
            // this batch's topology arrives, the next one is asked for
            x.ndx_p = x.ndx_c; x.f3_p = x.f3_c; x.r5_p = x.r5_c;
            cur = nxt;
            if (i + 64 < n) nxt = load_topo(tp, i + 64, lane, n);
            const bool act = i + lane < n;
            const int k = DPW_KIND(cur.b.y);
            x.f3_c = __ballot(act && k == 1); x.r5_c = __ballot(act && k == 2);
            x.ndx_c = cur.a.x; x.i0 = i;
        }
        if ((i % DPC_FLUSH) == 0 && i > 0) flush(i - DPC_FLUSH, i);
        if ((i % NPG) == 0) {
            // this group's cs has arrived (asked for a group ago), the next group's is asked for
            // (every stop node walked since has asked for a record of extras: four loads each, all of them later than this group's)
#ifdef DPC_PROFILE
            const unsigned long long w0 = __builtin_readcyclecounter();
#endif
            wait_vm4(dense ? nstop - cs_stop : 0);
#ifdef DPC_PROFILE
            pc_cswait += __builtin_readcyclecounter() - w0;
#endif
            ask_cs(i / NPG + 1);
            cs_stop = nstop;
        }
        DpcNode N;
        N.i = i; N.kfb = rl(cur.b.y, t); N.kind = DPW_KIND(N.kfb); N.frame = DPW_FRAME(N.kfb);
        N.ndx = rl(cur.a.x, t); N.stop_val = 0; N.lo = rl(cur.a.z, t); N.q1 = 0; N.q2 = 0;
        x.cur = i;
#ifdef DPC_PROFILE
        const unsigned long long pc_t1 = __builtin_readcyclecounter();
        pc_cyc[4] += pc_t1 - pc_t0; pc_cnt[4]++;
#endif
        // A node the fast routines do not cover (rare) goes through the reference's loop over its whole window; every other node
        // through the branch of its kind, from its loads to what it leaves in the history: nothing but the loop-carried state
        // crosses from one branch to the next.  (dpc_need_slow_* are uniform by construction; the readfirstlane tells the compiler.)
        const int slot = nstop & 3;
        DpcExt E;
        bool slow;
        if (N.kind == 0) { N.q1 = rl(cur.a.w, t); slow = dpc_need_slow_begin(R, U, N, x); }
        else if (N.kind == 2) { N.stop_val = rl(cur.a.y, t); N.q2 = rl(cur.b.x, t); slow = dpc_need_slow_r5(U, N, x); }
        else if (N.kind == 1) { if (!dense) ask_ext(i, slot); read_ext(E, slot, false); slow = false; }
        else {
            N.stop_val = rl(cur.a.y, t); N.q1 = rl(cur.a.w, t);
            if (!dense) ask_ext(i, slot);
            read_ext(E, slot, true);
            slow = dpc_need_slow_begin(R, U, N, x) || dpc_need_slow_r3(U, N, E, x);
        }
        if (N.kind & 1) { nstop++; if (dense) ask_ext(nstop + 2, (nstop + 2) & 3); }     // (three records of slack behind the last chain's)
        if (__builtin_expect(rfl((int)slow), 0)) {
#ifdef DPC_PROFILE
            pc_slow++; pc_slowsrc += i - N.lo;
#endif
            N.stop_val = rl(cur.a.y, t); N.q1 = rl(cur.a.w, t); N.q2 = rl(cur.b.x, t);
            const double cs = s_cs[(i / NPG) & 1][(((i % NPG) >> 1) * LW + ml) * 2 + (i & 1)];
            DpcOut B{0.0, -1, -1, 0.0, -1};
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // results written out by this wave are read back
            dpc_cand_slow(R, U, N, cs, E, M, x, B);
            if (N.kind == 0) dpc_finish_f5(N, cs, x.carry(N.frame), x, B);
            else if (N.kind == 2) dpc_finish_r5(R, N, negc, B);
            else dpc_finish_r3(U, N, x, B);
            x.hist_put(i, B.sv, -1, B.tb < 0 ? -1 : (B.tb | ((B.ov + 1) << DPW_TAG_BITS)));
        } else if (__builtin_expect((N.kind == 0 || N.kind == 3) && U.fp <= i - DPC_HIST, 0)) {
#ifdef DPC_PROFILE
            pc_deep++;
#endif
            // a gene begin whose unfolded range reaches back beyond the history: the same routines on the accessor that reads older
            // gene ends back from memory (a dense stretch of the contig: rare)
            const double cs = s_cs[(i / NPG) & 1][(((i % NPG) >> 1) * LW + ml) * 2 + (i & 1)];
            DpcOut B{0.0, -1, -1, 0.0, -1};
            if (N.kind == 0) { const DpcCarry c = x.carry(N.frame); dpc_cand_f5(R, U, N, M, xd, B); dpc_finish_f5(N, cs, c, x, B); }
            else { dpc_cand_r3(R, U, N, E, M, xd, B); dpc_finish_r3(U, N, x, B); }
            x.hist_put(i, B.val, -1, B.tb < 0 ? -1 : (B.tb | ((B.ov + 1) << DPW_TAG_BITS)));
        } else if (N.kind == 0) {
            const double cs = s_cs[(i / NPG) & 1][(((i % NPG) >> 1) * LW + ml) * 2 + (i & 1)];
            const DpcCarry c = x.carry(N.frame);
            DpcOut B{0.0, -1, -1, 0.0, -1};
            dpc_cand_f5(R, U, N, M, x, B);
            dpc_finish_f5(N, cs, c, x, B);
            x.hist_put(i, B.val, -1, B.tb);
        } else if (N.kind == 2) {
            const double cs = s_cs[(i / NPG) & 1][(((i % NPG) >> 1) * LW + ml) * 2 + (i & 1)];
            DpcOut B{0.0, -1, -1, 0.0, -1};
            dpc_cand_r5(U, N, cs, negc, x, B);
            dpc_finish_r5(R, N, negc, B);
            x.hist_put(i, B.sv, -1, B.tb);
        } else if (N.kind == 1) {
            DpcOut B{0.0, -1, -1, 0.0, -1};
            dpc_cand_f3(x.carry(N.frame), B);
            dpc_finish_f3(R, U, N, E, x, B);
            x.hist_put(i, B.sv, B.tbn, B.tb);
        } else {
            DpcOut B{0.0, -1, -1, 0.0, -1};
            dpc_cand_r3(R, U, N, E, M, x, B);
            dpc_finish_r3(U, N, x, B);
            x.hist_put(i, B.val, -1, B.tb < 0 ? -1 : (B.tb | ((B.ov + 1) << DPW_TAG_BITS)));
        }
#ifdef DPC_PROFILE
        pc_cyc[N.kind] += __builtin_readcyclecounter() - pc_t1; pc_cnt[N.kind]++;
#endif
    }
#ifdef DPC_PROFILE
    if (blockIdx.x == 100 && lane == 0)
        printf("[dpc profile] ext wait per stop %.0f, cs wait per group %.0f\n", (double)pc_wait / (pc_cnt[1] + pc_cnt[3] + 1), (double)pc_cswait / (n / 8 + 1));
    if ((pc_slow || pc_deep) && lane == 0 && (blockIdx.x % 64) == 0) printf("[dpc profile] wave %d n=%d: slow %d (sources %d) deep %d\n", (int)blockIdx.x, n, pc_slow, pc_slowsrc, pc_deep);
    if (blockIdx.x == 100 && lane == 0)
        printf("[dpc profile] wave 100: n=%d models=%d | cycles per node: head %.0f | F5 %.0f (x%d) F3 %.0f (x%d) R5 %.0f (x%d) R3 %.0f (x%d)\n", n, count,
               (double)pc_cyc[4] / pc_cnt[4], (double)pc_cyc[0] / (pc_cnt[0] ? pc_cnt[0] : 1), pc_cnt[0], (double)pc_cyc[1] / (pc_cnt[1] ? pc_cnt[1] : 1), pc_cnt[1],
               (double)pc_cyc[2] / (pc_cnt[2] ? pc_cnt[2] : 1), pc_cnt[2], (double)pc_cyc[3] / (pc_cnt[3] ? pc_cnt[3] : 1), pc_cnt[3]);
#endif
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

void pga_launch_dp_contig(const int2* d_waves, int n_waves, const ChainDesc* d_chains, const DpwGroupPtrs& groups, const ModelConst* d_models,
                          DpBuffers buf, const DpwBuffers& wb, hipStream_t st) {
    if (n_waves <= 0) return;
    hipLaunchKernelGGL(k_dp_contig<DPC_LW>, dim3((unsigned)n_waves), dim3(64), 0, st, d_waves, d_chains, groups, (const double*)wb.cs,
                       (const DpwExt*)wb.ext, d_models, buf);
}
