// Device helpers shared by dp.hip and pipeline.hip.
#pragma once
#include "pga_internal.h"

// General _intergenic_mod_same (ref: _connection.h:52-78); a = n1, b = n2, both on strand a_strand.
// igm_tab[d] = (2 - d/60) * 0.15 * st_wt for d = 0..60 (host-computed, ModelConst::igm).
__device__ __forceinline__ double igm_same_dev(int a_ndx, int a_strand, double a_r, double a_u,
                                               int b_ndx, double b_r, double b_u, double st_wt, const double* igm_tab) {
    const int dist = abs(a_ndx - b_ndx);
    const bool ovl = a_ndx + 2 * a_strand >= b_ndx;
    double r = 0.0;
    if (a_ndx + 2 == b_ndx || a_ndx == b_ndx + 1) {
        if (a_strand == 1) { if (b_r < 0) r -= b_r; if (b_u < 0) r -= b_u; }
        else               { if (a_r < 0) r -= a_r; if (a_u < 0) r -= a_u; }
    }
    if (dist > 3 * PGA_OPER_DIST) r -= 0.15 * st_wt;
    else if ((dist <= PGA_OPER_DIST && !ovl) || dist * 4 < PGA_OPER_DIST) r += igm_tab[dist];
    return r;
}

// chain containing global chain-node index g (chains sorted by off)
__device__ __forceinline__ int find_chain(const ChainDesc* __restrict__ chains, int n_chains, int64_t g) {
    int lo = 0, hi = n_chains - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (chains[mid].off <= g) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// The last index in [0, n) whose key is <= x (keys ascend, key(0) <= x), found by ONE WAVEFRONT together: every lane probes one of 64
// evenly spaced positions of what is left of the range, a ballot narrows it 64-fold -- three rounds of one load each for 262 144
// entries, where the binary search of a lone thread is eighteen dependent loads in front of everything the workgroup does (round 6:
// k_ovl_stops, a workgroup of 256 short threads, spent a third of its life there).  All 64 lanes of the wavefront must call it.
template <class Key>
__device__ __forceinline__ int wave_search_le(const Key key, const int n, const int64_t x) {
    const int lane = threadIdx.x & 63;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int step = (hi - lo + 63) >> 6;
        const int idx = min(lo + (lane + 1) * step, hi);           // (lane 63 probes hi)
        const int cnt = __popcll(__ballot((int64_t)key(idx) <= x));      // a prefix of the lanes
        const int nlo = cnt ? min(lo + cnt * step, hi) : lo;
        if (cnt < 64) hi = min(lo + (cnt + 1) * step, hi) - 1;
        lo = nlo;
    }
    return lo;
}
// ... once per workgroup (its first wavefront), the result in *s_slot for everybody (barrier inside: every thread must call it)
template <class Key>
__device__ __forceinline__ int block_search_le(const Key key, const int n, const int64_t x, int* s_slot) {
    if (threadIdx.x < 64) { const int r = wave_search_le(key, n, x); if (threadIdx.x == 0) *s_slot = r; }
    __syncthreads();
    return *s_slot;
}

// The same for every thread of a workgroup whose first element is block_first: one search per workgroup (its first wavefront), then
// a short forward walk per thread (a workgroup rarely spans more than two chains) -- a search per thread is sixteen dependent
// loads that nothing hides.  Every thread of the workgroup must call it (barrier inside).
__device__ __forceinline__ int find_chain_block(const ChainDesc* __restrict__ chains, int n_chains, int64_t block_first, int64_t g, int* s_slot) {
    int c = block_search_le([&](const int k) { return chains[k].off; }, n_chains, block_first, s_slot);
    while (c + 1 < n_chains && chains[c + 1].off <= g) c++;
    return c;
}
