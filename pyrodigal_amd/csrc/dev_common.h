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

// The same for every thread of a workgroup whose first element is block_first: one binary search per workgroup (thread 0), then
// a short forward walk per thread (a workgroup rarely spans more than two chains) -- a search per thread is sixteen dependent
// loads that nothing hides.  Every thread of the workgroup must call it (barrier inside).
__device__ __forceinline__ int find_chain_block(const ChainDesc* __restrict__ chains, int n_chains, int64_t block_first, int64_t g, int* s_slot) {
    if (threadIdx.x == 0) *s_slot = find_chain(chains, n_chains, block_first);
    __syncthreads();
    int c = *s_slot;
    while (c + 1 < n_chains && chains[c + 1].off <= g) c++;
    return c;
}
