// Parallel traceback tail (included by finder.hip): the path of every contig, its untangling, the bad-gene elimination
// and the gene list, without a pointer chase.
//
// ref: lib.pyx:1253-1311 (_disentangle_overlaps, _max_forward_pointers, tracef), Prodigal dprog.c eliminate_bad_genes
// (call sites lib.pyx:5308, 5369), lib.pyx:3231-3270 (Genes._extract).
//
// The reference follows traceb from the best gene end back to the head of the path, one node after the other, three
// times (two untangling passes that splice nodes into the path, one pass that sets tracef), then walks the path twice
// more.  One device thread per contig does the same at one memory round trip per step (k_tail_path): fine for a batch
// of small contigs side by side, hopeless for a genome (18 k steps).  Here:
//   * the nodes ON the path are found by pointer doubling: in round k every node's pointer reaches its 2^k-th traceb
//     ancestor and every marked node marks that ancestor, starting from the best gene end -- every ancestor of a marked node
//     is itself on the path, so the order inside a round does not matter; log2(longest chain) rounds;
//   * traceb always points to a smaller index, so the path in walk order is simply the marked nodes in decreasing index
//     order: a prefix sum over the marks gives every node its position;
//   * both untangling passes only look at one path edge (p, traceb[p]) at a time and splice 0, 1 or 2 nodes into it; the
//     nodes they splice in never satisfy a splice condition themselves (the start of an overlapping gene is no stop; the
//     reverse stop spliced in by the first pass has its ov_mark reset; no second-pass rule fires on the new edges), so
//     the final path is the original one with per-edge insertions, counted before the prefix sum;
//   * elimination and the gene list are per-path-position work with neighbours (k_tp_elim_*, k_tp_extract: the
//     reference's "last begin / end / start node / stop node seen" state becomes four running maxima of positions).
// The start tweaks (k_tail_tweak, k_tail_tweak_fixup) and k_emit_genes follow as for the one-thread-per-contig tail.

struct TpSeg { int64_t off; int32_t n; int32_t contig; };     // winning chains in the gathered arrays, ascending `off`

struct TpWork {
    const TpSeg* seg; int n_seg;
    int64_t nw;                 // nodes of all winning chains
    int levels;                 // 2^levels >= longest chain
    int32_t* up;                // [2][nw] global index of the 2^k-th ancestor in round k (a head points to itself), ping-pong
    uint8_t* mark;              // node is on the path
    int32_t* slots;             // path positions the node takes: itself + what the untangling splices in after it
    int32_t* ins;               // [nw][2] the spliced-in nodes (chain index), -1 = none
    int32_t* excl;              // [nw + 1] exclusive prefix sum of slots
    int32_t* bsum;              // per 256 nodes
    int32_t* cnt;               // [n_seg] path length
};

__device__ inline int tp_seg_of(const TpSeg* __restrict__ seg, int n_seg, int64_t g) {
    int lo = 0, hi = n_seg - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (seg[mid].off <= g) lo = mid; else hi = mid - 1; }
    return lo;
}

__global__ void __launch_bounds__(256)
k_tp_init(TpWork w, const TailDesc* __restrict__ td, OutArrays o) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= w.nw) return;
    const TpSeg s = w.seg[tp_seg_of(w.seg, w.n_seg, g)];
    const int i = (int)(g - s.off);
    const int tb = o.traceb[g];
    w.up[g] = tb >= 0 ? (int32_t)(s.off + tb) : (int32_t)g;
    w.mark[g] = i == td[s.contig].mx ? 1 : 0;
}

// one round of pointer doubling: every marked node marks its 2^k-th ancestor, every node's pointer doubles its reach.
// After round k the marks cover the first 2^(k+1) nodes of the path (a node marked during the round may already pass the
// mark on: whatever it reaches is an ancestor of the best gene end, hence on the path).
__global__ void __launch_bounds__(256)
k_tp_jump(TpWork w, const int32_t* __restrict__ up_in, int32_t* __restrict__ up_out) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= w.nw) return;
    const int a = up_in[g];
    if (w.mark[g]) w.mark[a] = 1;
    up_out[g] = up_in[a];
}
// Eight hops per launch (round 6): pointers that reach R nodes become pointers that reach 8 R, a marked node marks its ancestors at R, 2 R,
// .. 7 R -- after round k the marks cover the first 8^(k+1) nodes of the path, six launches for a chain of 262 144 nodes where the
// doubling form takes eighteen (a launch of this kind is 3 us of work and 4 us of gap to the next: the genome-sized calls pay per launch).
__global__ void __launch_bounds__(256)
k_tp_jump8(TpWork w, const int32_t* __restrict__ up_in, int32_t* __restrict__ up_out) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= w.nw) return;
    const bool m = w.mark[g] != 0;
    int a = up_in[g];
#pragma unroll
    for (int h = 0; h < 7; h++) {
        if (m) w.mark[a] = 1;
        a = up_in[a];
    }
    up_out[g] = a;
}

// what the two untangling passes splice in after path node p (edge p -> nx = traceb[p]); ref: lib.pyx:1253-1295
__global__ void __launch_bounds__(256)
k_tp_slots(TpWork w, const TailDesc* __restrict__ td, OutArrays o) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= w.nw) return;
    int slots = 0, i0 = -1, i1 = -1;
    if (w.mark[g]) {
        slots = 1;
        const TpSeg s = w.seg[tp_seg_of(w.seg, w.n_seg, g)];
        const TailDesc d = td[s.contig];
        NodeView v = node_view(d, o, nullptr, nullptr);
        const int p = (int)(g - s.off), nx = v.traceb[p];
        if (nx != -1) {
            const int sp = v.strand[p], sn = v.strand[nx], ov = v.ov_mark[p], np = v.ndx[p], nn = v.ndx[nx];
            const bool stp = is_stop_n(v, p), stn = is_stop_n(v, nx);
            if ((sp == -1) & stp & (sn == 1) & stn & (ov != -1) & (np > nn)) {
                i0 = v.star_ptr[3 * p + ov];
                i1 = walk_down_to(v, i0, v.stop_val[i0]);
                slots = 3;
            } else {
                const bool p_rb = sp == -1 && !stp, p_fs = sp == 1 && stp, p_rs = sp == -1 && stp;
                const bool n_fs = sn == 1 && stn, n_rs = sn == -1 && stn;
                if (p_rb && n_fs) i0 = walk_down_to(v, p, v.stop_val[p]);
                if (p_fs && n_fs) i0 = v.star_ptr[3 * nx + np % 3];
                if (p_rs && n_rs) i0 = v.star_ptr[3 * p + nn % 3];
                if (i0 != -1) slots = 2;
            }
        }
    }
    w.slots[g] = slots; w.ins[2 * g] = i0; w.ins[2 * g + 1] = i1;
}

// exclusive prefix sum of slots over all nodes: block sums, their scan, the final values
__global__ void __launch_bounds__(256)
k_tp_scan1(TpWork w) {
    __shared__ int s_w[4];
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int v = g < w.nw ? w.slots[g] : 0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) w.bsum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ void __launch_bounds__(1024)
k_tp_scan2(TpWork w, const int nblocks) {
    __shared__ int s_wsum[16];
    __shared__ int s_base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int q0 = 0; q0 < nblocks; q0 += 1024) {
        const int q = q0 + t;
        const int c = q < nblocks ? w.bsum[q] : 0;
        int inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o2 = __shfl_up(inc, d, 64); if (lane >= d) inc += o2; }
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < wave; k++) woff += s_wsum[k];
        const int base = s_base;
        if (q < nblocks) w.bsum[q] = base + woff + inc - c;
        __syncthreads();
        if (t == 1023) s_base = base + woff + inc;
        __syncthreads();
    }
    if (t == 0) w.excl[w.nw] = s_base;
}
__global__ void __launch_bounds__(256)
k_tp_scan3(TpWork w) {
    __shared__ int s_w[4];
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = g < w.nw ? w.slots[g] : 0;
    int inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o2 = __shfl_up(inc, d, 64); if (lane >= d) inc += o2; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < wave; k++) woff += s_w[k];
    if (g < w.nw) w.excl[g] = w.bsum[blockIdx.x] + woff + inc - c;
}

// the path lists (path[0] = the best gene end, path[cnt - 1] = the head), ov_mark of the reverse stops spliced in
__global__ void __launch_bounds__(256)
k_tp_fill(TpWork w, OutArrays o, int32_t* __restrict__ path) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= w.nw) return;
    const int sl = w.slots[g];
    if (sl == 0) return;
    const int si = tp_seg_of(w.seg, w.n_seg, g);
    const TpSeg s = w.seg[si];
    const int q = w.excl[s.off + s.n] - w.excl[g] - sl;          // path positions taken by the nodes after g in walk order
    int32_t* pl = path + s.off;
    pl[q] = (int32_t)(g - s.off);
    if (sl >= 2) pl[q + 1] = w.ins[2 * g];
    if (sl == 3) { pl[q + 2] = w.ins[2 * g + 1]; o.ov_mark[s.off + w.ins[2 * g + 1]] = -1; }
    if (q == 0) w.cnt[si] = w.excl[s.off + s.n] - w.excl[s.off];      // the best gene end: first in walk order
}

// traceb / tracef along the final path
__global__ void __launch_bounds__(256)
k_tp_link(TpWork w, OutArrays o, const int32_t* __restrict__ path, int32_t* __restrict__ tracef) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= w.nw) return;
    const int si = tp_seg_of(w.seg, w.n_seg, g);
    const TpSeg s = w.seg[si];
    const int q = (int)(g - s.off);
    if (q + 1 >= w.cnt[si]) return;
    const int x = path[s.off + q], y = path[s.off + q + 1];
    o.traceb[s.off + x] = y; tracef[s.off + y] = x;
}

// Prodigal dprog.c eliminate_bad_genes, first loop: a node's start score receives at most two terms, in path order
__global__ void __launch_bounds__(256)
k_tp_elim_a(TpWork w, const TailDesc* __restrict__ td, OutArrays o, const int32_t* __restrict__ path) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= w.nw) return;
    const int si = tp_seg_of(w.seg, w.n_seg, g);
    const TpSeg s = w.seg[si];
    const int q = (int)(g - s.off), cnt = w.cnt[si];
    if (q >= cnt || cnt < 2) return;                  // cnt < 2: the best gene end has no traceb (ref: lib.pyx:1311)
    const TailDesc d = td[s.contig];
    NodeView v = node_view(d, o, nullptr, nullptr);
    const int32_t* pl = path + s.off;
    const int x = pl[q];
    if (q + 1 <= cnt - 1) {
        const int p = pl[q + 1];
        if (v.strand[p] == 1 && is_stop_n(v, p)) v.sscore[x] += igm_h(v, p, x, d.st_wt);
    }
    if (q >= 1 && v.strand[x] == -1 && !is_stop_n(v, x)) v.sscore[x] += igm_h(v, x, pl[q - 1], d.st_wt);
}
__global__ void __launch_bounds__(256)
k_tp_elim_b(TpWork w, const TailDesc* __restrict__ td, OutArrays o, const int32_t* __restrict__ path, uint8_t* __restrict__ elim) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= w.nw) return;
    const int si = tp_seg_of(w.seg, w.n_seg, g);
    const TpSeg s = w.seg[si];
    const int q = (int)(g - s.off), cnt = w.cnt[si];
    if (q < 1 || q >= cnt) return;
    const TailDesc d = td[s.contig];
    const NodeView v = node_view(d, o, nullptr, nullptr);
    const int32_t* pl = path + s.off;
    const int p = pl[q], f = pl[q - 1];
    const int sp = v.strand[p]; const bool stp = is_stop_n(v, p);
    const double gp = v.cscore[p] + v.sscore[p], gf = v.cscore[f] + v.sscore[f];
    if ((sp == 1 && !stp && gp < 0) || (sp == -1 && stp && gf < 0)) { elim[s.off + p] = 1; elim[s.off + f] = 1; }
}

// Genes._extract (ref: lib.pyx:3231-3270): one workgroup per contig goes over the path from its head in chunks; the
// reference's running begin / end / start node / stop node become "the last position that set it", a running maximum.
__global__ void __launch_bounds__(1024)
k_tp_extract(TpWork w, const TailDesc* __restrict__ td, OutArrays o, const int32_t* __restrict__ path, const uint8_t* __restrict__ elim,
             GeneRec* __restrict__ genes, int32_t* __restrict__ n_genes) {
    __shared__ int s_sc[16][4];      // per wave: running maxima of the four setters, then the emit count
    __shared__ int s_cnt[16];
    __shared__ int s_carry[4];       // last setter position (in processing order) of b, e, s, t before this chunk
    __shared__ int s_ng;
    const int si = blockIdx.x;
    const TpSeg s = w.seg[si];
    const TailDesc d = td[s.contig];
    const int cnt = w.cnt[si];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) { s_ng = 0; }
    if (t < 4) s_carry[t] = -1;
    __syncthreads();
    if (cnt < 2) { if (t == 0) n_genes[s.contig] = 0; return; }
    const NodeView v = node_view(d, o, nullptr, nullptr);
    const int32_t* pl = path + s.off;
    const uint8_t* el = elim + s.off;
    GeneRec* out = genes + d.gene_off;
    // processing order r = 0 .. cnt-1 is path position q = cnt-1-r (head first)
    auto val_b = [&](int r) { const int p = pl[cnt - 1 - r]; return v.strand[p] == 1 ? v.ndx[p] + 1 : v.ndx[p] - 1; };
    auto val_e = [&](int r) { const int p = pl[cnt - 1 - r]; return v.strand[p] == 1 ? v.ndx[p] + 3 : v.ndx[p] + 1; };
    const int nthr = blockDim.x;
    for (int r0 = 0; r0 < cnt; r0 += nthr) {
        const int r = r0 + t;
        bool live = false, fwd = false, stp = false;
        if (r < cnt) {
            const int p = pl[cnt - 1 - r];
            live = el[p] != 1; fwd = v.strand[p] == 1; stp = is_stop_n(v, p);
        }
        // who sets what (ref: the four branches of Genes._extract)
        const bool set_b = live && ((fwd && !stp) || (!fwd && stp));       // forward start: begin; reverse stop: begin
        const bool set_e = live && ((fwd && stp) || (!fwd && !stp));       // forward stop: end;   reverse start: end
        const bool set_s = live && !stp;                                    // start node
        const bool set_t = live && stp;                                     // stop node
        const bool emit = live && ((fwd && stp) || (!fwd && !stp));
        int m[4] = {set_b ? r : -1, set_e ? r : -1, set_s ? r : -1, set_t ? r : -1};
        int e1 = emit ? 1 : 0;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
#pragma unroll
            for (int k = 0; k < 4; k++) { const int o2 = __shfl_up(m[k], dd, 64); if (lane >= dd) m[k] = max(m[k], o2); }
            const int o3 = __shfl_up(e1, dd, 64); if (lane >= dd) e1 += o3;
        }
        if (lane == 63) { for (int k = 0; k < 4; k++) s_sc[wave][k] = m[k]; s_cnt[wave] = e1; }
        __syncthreads();
        int eoff = s_ng;
        for (int k2 = 0; k2 < wave; k2++) {
            for (int k = 0; k < 4; k++) m[k] = max(m[k], s_sc[k2][k]);
            eoff += s_cnt[k2];
        }
        for (int k = 0; k < 4; k++) m[k] = max(m[k], s_carry[k]);
        if (emit) {
            GeneRec gr;
            gr.begin = m[0] >= 0 ? val_b(m[0]) : 0; gr.end = m[1] >= 0 ? val_e(m[1]) : 0;
            gr.start_ndx = m[2] >= 0 ? pl[cnt - 1 - m[2]] : 0; gr.stop_ndx = m[3] >= 0 ? pl[cnt - 1 - m[3]] : 0;
            out[eoff + e1 - 1] = gr;
        }
        __syncthreads();
        if (t == nthr - 1) { for (int k = 0; k < 4; k++) s_carry[k] = m[k]; s_ng = eoff + e1; }
        __syncthreads();
    }
    if (t == 0) n_genes[s.contig] = s_ng;
}

// ---- all of the above in ONE launch when every winning chain fits the LDS of a workgroup (batches of short contigs) ----
// A workgroup per contig.  The chain's traceb goes to LDS in one coalesced pass; ONE lane then follows it from the best gene end
// to the head of the path -- a pointer chase, but through LDS (a path of a 20 kbp contig is some sixty nodes: a few microseconds) --
// and what it lists IS the path in walk order, so marks, pointer doubling and the prefix sum over all nodes of the batch (twenty
// launches over every node of every winning chain) are not needed: the splices are counted per path position, a prefix sum over
// the path places them, and the passes that follow (links, the two elimination loops, the gene list) run over the final path in
// LDS.  Same rules, same order of the floating-point additions as k_tp_slots .. k_tp_extract.
constexpr int TP_SMALL_MAX = 4096;        // nodes of the longest chain: 3 arrays of int32 in LDS (48 KB)
constexpr int TP_SPLICED_STOP = 1 << 30;  // tag of a path entry: the reverse stop the first untangling pass splices in (its ov_mark is reset)

__device__ inline int tp_block_excl_scan(int v, int* s_w, int& total) {     // exclusive scan over the workgroup; total = sum
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nwv = blockDim.x >> 6;
    int inc = v;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) { const int x = __shfl_up(inc, dd, 64); if (lane >= dd) inc += x; }
    __syncthreads();                        // s_w may still be read from the previous round
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int off = 0; total = 0;
    for (int k = 0; k < nwv; k++) { if (k < wave) off += s_w[k]; total += s_w[k]; }
    return off + inc - v;
}

__global__ void __launch_bounds__(256)
k_tp_small(TpWork w, const TailDesc* __restrict__ td, OutArrays o, int32_t* __restrict__ tracef, uint8_t* __restrict__ elim,
           GeneRec* __restrict__ genes, int32_t* __restrict__ n_genes, const int cap) {
    extern __shared__ int32_t s_dyn[];
    int32_t* s_tb = s_dyn; int32_t* s_base = s_dyn + cap; int32_t* s_pl = s_dyn + 2 * cap;
    __shared__ int s_w[4], s_m;
    __shared__ int s_sc[4][4], s_cnt[4], s_carry[4], s_ng;
    const int si = blockIdx.x;
    const TpSeg s = w.seg[si];
    const TailDesc d = td[s.contig];
    const int n = s.n, t = threadIdx.x, nthr = blockDim.x, lane = t & 63, wave = t >> 6, nwv = nthr >> 6;
    NodeView v = node_view(d, o, tracef, elim);
    for (int i = t; i < n; i += nthr) s_tb[i] = v.traceb[i];
    if (t == 0) { s_ng = 0; s_m = 0; }
    if (t < 4) s_carry[t] = -1;
    __syncthreads();
    if (t == 0 && d.mx >= 0) { int m = 0; for (int p = d.mx; p != -1; p = s_tb[p]) s_base[m++] = p; s_m = m; }
    __syncthreads();
    const int m = s_m;
    // what the two untangling passes splice in after path node p (edge p -> nx); ref: lib.pyx:1253-1295
    int cnt = 0;
    for (int k0 = 0; k0 < m; k0 += nthr) {
        const int k = k0 + t;
        int slots = 0, i0 = -1, i1 = -1, p = -1;
        if (k < m) {
            slots = 1; p = s_base[k];
            const int nx = s_tb[p];
            if (nx != -1) {
                const int sp = v.strand[p], sn = v.strand[nx], ov = v.ov_mark[p], np = v.ndx[p], nn = v.ndx[nx];
                const bool stp = is_stop_n(v, p), stn = is_stop_n(v, nx);
                if ((sp == -1) & stp & (sn == 1) & stn & (ov != -1) & (np > nn)) {
                    i0 = v.star_ptr[3 * p + ov];
                    i1 = walk_down_to(v, i0, v.stop_val[i0]);
                    slots = 3;
                } else {
                    const bool p_rb = sp == -1 && !stp, p_fs = sp == 1 && stp, p_rs = sp == -1 && stp;
                    const bool n_fs = sn == 1 && stn, n_rs = sn == -1 && stn;
                    if (p_rb && n_fs) i0 = walk_down_to(v, p, v.stop_val[p]);
                    if (p_fs && n_fs) i0 = v.star_ptr[3 * nx + np % 3];
                    if (p_rs && n_rs) i0 = v.star_ptr[3 * p + nn % 3];
                    if (i0 != -1) slots = 2;
                }
            }
        }
        int total;
        const int q = cnt + tp_block_excl_scan(slots, s_w, total);
        if (q + slots <= cap) {           // always: the spliced-in nodes are nodes of the chain that are not on the path
            if (slots >= 1) s_pl[q] = p;
            if (slots >= 2) s_pl[q + 1] = i0;
            if (slots == 3) s_pl[q + 2] = i1 | TP_SPLICED_STOP;
        }
        cnt += total;
    }
    cnt = min(cnt, cap);
    __syncthreads();
    // ov_mark of the spliced-in reverse stops (after every splice was decided: the decisions read ov_mark), traceb / tracef along the path
    for (int q = t; q < cnt; q += nthr) {
        const int e = s_pl[q];
        if (e & TP_SPLICED_STOP) { v.ov_mark[e & ~TP_SPLICED_STOP] = -1; s_pl[q] = e & ~TP_SPLICED_STOP; }
    }
    __syncthreads();
    for (int q = t; q + 1 < cnt; q += nthr) { const int x = s_pl[q], y = s_pl[q + 1]; v.traceb[x] = y; v.tracef[y] = x; }
    if (cnt < 2) { if (t == 0) n_genes[s.contig] = 0; return; }          // the best gene end has no traceb (ref: lib.pyx:1311)
    // Prodigal dprog.c eliminate_bad_genes, first loop: a node's start score receives at most two terms, in path order
    for (int q = t; q < cnt; q += nthr) {
        const int x = s_pl[q];
        if (q + 1 <= cnt - 1) {
            const int p = s_pl[q + 1];
            if (v.strand[p] == 1 && is_stop_n(v, p)) v.sscore[x] += igm_h(v, p, x, d.st_wt);
        }
        if (q >= 1 && v.strand[x] == -1 && !is_stop_n(v, x)) v.sscore[x] += igm_h(v, x, s_pl[q - 1], d.st_wt);
    }
    __threadfence_block();
    __syncthreads();
    for (int q = 1 + t; q < cnt; q += nthr) {
        const int p = s_pl[q], f = s_pl[q - 1];
        const int sp = v.strand[p]; const bool stp = is_stop_n(v, p);
        const double gp = v.cscore[p] + v.sscore[p], gf = v.cscore[f] + v.sscore[f];
        if ((sp == 1 && !stp && gp < 0) || (sp == -1 && stp && gf < 0)) { v.elim[p] = 1; v.elim[f] = 1; }
    }
    __threadfence_block();
    __syncthreads();
    // Genes._extract (ref: lib.pyx:3231-3270), as k_tp_extract
    GeneRec* out = genes + d.gene_off;
    auto val_b = [&](int r) { const int p = s_pl[cnt - 1 - r]; return v.strand[p] == 1 ? v.ndx[p] + 1 : v.ndx[p] - 1; };
    auto val_e = [&](int r) { const int p = s_pl[cnt - 1 - r]; return v.strand[p] == 1 ? v.ndx[p] + 3 : v.ndx[p] + 1; };
    for (int r0 = 0; r0 < cnt; r0 += nthr) {
        const int r = r0 + t;
        bool live = false, fwd = false, stp = false;
        if (r < cnt) {
            const int p = s_pl[cnt - 1 - r];
            live = v.elim[p] != 1; fwd = v.strand[p] == 1; stp = is_stop_n(v, p);
        }
        const bool set_b = live && ((fwd && !stp) || (!fwd && stp));
        const bool set_e = live && ((fwd && stp) || (!fwd && !stp));
        const bool set_s = live && !stp;
        const bool set_t = live && stp;
        const bool emit = live && ((fwd && stp) || (!fwd && !stp));
        int mm[4] = {set_b ? r : -1, set_e ? r : -1, set_s ? r : -1, set_t ? r : -1};
        int e1 = emit ? 1 : 0;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
#pragma unroll
            for (int k = 0; k < 4; k++) { const int o2 = __shfl_up(mm[k], dd, 64); if (lane >= dd) mm[k] = max(mm[k], o2); }
            const int o3 = __shfl_up(e1, dd, 64); if (lane >= dd) e1 += o3;
        }
        if (lane == 63) { for (int k = 0; k < 4; k++) s_sc[wave][k] = mm[k]; s_cnt[wave] = e1; }
        __syncthreads();
        int eoff = s_ng;
        for (int k2 = 0; k2 < wave; k2++) {
            for (int k = 0; k < 4; k++) mm[k] = max(mm[k], s_sc[k2][k]);
            eoff += s_cnt[k2];
        }
        for (int k = 0; k < 4; k++) mm[k] = max(mm[k], s_carry[k]);
        if (emit) {
            GeneRec gr;
            gr.begin = mm[0] >= 0 ? val_b(mm[0]) : 0; gr.end = mm[1] >= 0 ? val_e(mm[1]) : 0;
            gr.start_ndx = mm[2] >= 0 ? s_pl[cnt - 1 - mm[2]] : 0; gr.stop_ndx = mm[3] >= 0 ? s_pl[cnt - 1 - mm[3]] : 0;
            out[eoff + e1 - 1] = gr;
        }
        __syncthreads();
        if (t == nthr - 1) { for (int k = 0; k < 4; k++) s_carry[k] = mm[k]; s_ng = eoff + e1; }
        __syncthreads();
    }
    if (t == 0) n_genes[s.contig] = s_ng;
    (void)nwv;
}

// ---- Genes._extract for genomes: the path cut into chunks of 1024 positions, a workgroup per chunk -----------------------------
// k_tp_extract is one workgroup per contig; a genome's path has some 400 000 positions, i.e. 400 rounds of one workgroup while
// the rest of the chip idles (1.5 ms on config 5).  Here every chunk first reports what it sets -- the last position (in
// processing order) that sets begin / end / start node / stop node, and how many genes it emits (k_tp_extract_sum) -- and then
// emits its genes with the carries of the chunks before it, which it combines itself (k_tp_extract_chunk).
struct TpChunkSum { int32_t m[4]; int32_t emits; int32_t _pad[3]; };

struct TpExtractItem { bool live, fwd, stp; };
__device__ inline TpExtractItem tp_extract_item(const NodeView& v, const int32_t* __restrict__ pl, const uint8_t* __restrict__ el, const int cnt, const int r) {
    TpExtractItem it{false, false, false};
    if (r < cnt) { const int p = pl[cnt - 1 - r]; it.live = el[p] != 1; it.fwd = v.strand[p] == 1; it.stp = is_stop_n(v, p); }
    return it;
}

__global__ void __launch_bounds__(1024)
k_tp_extract_sum(TpWork w, const TailDesc* __restrict__ td, OutArrays o, const int32_t* __restrict__ path, const uint8_t* __restrict__ elim,
                 TpChunkSum* __restrict__ sums, const int chunks_per_seg) {
    __shared__ int s_sc[16][5];
    const int si = blockIdx.y, b = blockIdx.x;
    const TpSeg s = w.seg[si];
    const int cnt = w.cnt[si];
    if (cnt < 2 || b * 1024 >= cnt) return;
    const TailDesc d = td[s.contig];
    const NodeView v = node_view(d, o, nullptr, nullptr);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = b * 1024 + t;
    const TpExtractItem it = tp_extract_item(v, path + s.off, elim + s.off, cnt, r);
    int m[5] = {(it.live && ((it.fwd && !it.stp) || (!it.fwd && it.stp))) ? r : -1, (it.live && ((it.fwd && it.stp) || (!it.fwd && !it.stp))) ? r : -1,
                (it.live && !it.stp) ? r : -1, (it.live && it.stp) ? r : -1, (it.live && ((it.fwd && it.stp) || (!it.fwd && !it.stp))) ? 1 : 0};
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) {
#pragma unroll
        for (int k = 0; k < 4; k++) m[k] = max(m[k], __shfl_xor(m[k], dd, 64));
        m[4] += __shfl_xor(m[4], dd, 64);
    }
    if (lane == 0) for (int k = 0; k < 5; k++) s_sc[wave][k] = m[k];
    __syncthreads();
    if (t == 0) {
        TpChunkSum cs{{-1, -1, -1, -1}, 0, {0, 0, 0}};
        for (int k2 = 0; k2 < 16; k2++) { for (int k = 0; k < 4; k++) cs.m[k] = max(cs.m[k], s_sc[k2][k]); cs.emits += s_sc[k2][4]; }
        sums[(size_t)si * chunks_per_seg + b] = cs;
    }
}

__global__ void __launch_bounds__(1024)
k_tp_extract_chunk(TpWork w, const TailDesc* __restrict__ td, OutArrays o, const int32_t* __restrict__ path, const uint8_t* __restrict__ elim,
                   const TpChunkSum* __restrict__ sums, const int chunks_per_seg, GeneRec* __restrict__ genes, int32_t* __restrict__ n_genes) {
    __shared__ int s_sc[16][5];
    __shared__ int s_carry[5];
    const int si = blockIdx.y, b = blockIdx.x;
    const TpSeg s = w.seg[si];
    const int cnt = w.cnt[si];
    if (cnt < 2) { if (b == 0 && threadIdx.x == 0) n_genes[s.contig] = 0; return; }
    if (b * 1024 >= cnt) return;
    const TailDesc d = td[s.contig];
    const NodeView v = node_view(d, o, nullptr, nullptr);
    const int32_t* pl = path + s.off;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // what the chunks before this one leave behind
    {
        int c[5] = {-1, -1, -1, -1, 0};
        for (int k = t; k < b; k += 1024) {
            const TpChunkSum cs = sums[(size_t)si * chunks_per_seg + k];
            for (int q = 0; q < 4; q++) c[q] = max(c[q], cs.m[q]);
            c[4] += cs.emits;
        }
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) {
#pragma unroll
            for (int q = 0; q < 4; q++) c[q] = max(c[q], __shfl_xor(c[q], dd, 64));
            c[4] += __shfl_xor(c[4], dd, 64);
        }
        if (lane == 0) for (int q = 0; q < 5; q++) s_sc[wave][q] = c[q];
        __syncthreads();
        if (t == 0) {
            int cc[5] = {-1, -1, -1, -1, 0};
            for (int k2 = 0; k2 < 16; k2++) { for (int q = 0; q < 4; q++) cc[q] = max(cc[q], s_sc[k2][q]); cc[4] += s_sc[k2][4]; }
            for (int q = 0; q < 5; q++) s_carry[q] = cc[q];
        }
        __syncthreads();
    }
    GeneRec* out = genes + d.gene_off;
    auto val_b = [&](int r) { const int p = pl[cnt - 1 - r]; return v.strand[p] == 1 ? v.ndx[p] + 1 : v.ndx[p] - 1; };
    auto val_e = [&](int r) { const int p = pl[cnt - 1 - r]; return v.strand[p] == 1 ? v.ndx[p] + 3 : v.ndx[p] + 1; };
    const int r = b * 1024 + t;
    const TpExtractItem it = tp_extract_item(v, pl, elim + s.off, cnt, r);
    const bool set_b = it.live && ((it.fwd && !it.stp) || (!it.fwd && it.stp));
    const bool set_e = it.live && ((it.fwd && it.stp) || (!it.fwd && !it.stp));
    const bool emit = set_e;
    int mm[4] = {set_b ? r : -1, set_e ? r : -1, (it.live && !it.stp) ? r : -1, (it.live && it.stp) ? r : -1};
    int e1 = emit ? 1 : 0;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
#pragma unroll
        for (int k = 0; k < 4; k++) { const int o2 = __shfl_up(mm[k], dd, 64); if (lane >= dd) mm[k] = max(mm[k], o2); }
        const int o3 = __shfl_up(e1, dd, 64); if (lane >= dd) e1 += o3;
    }
    __syncthreads();                                   // s_sc is read above
    if (lane == 63) { for (int k = 0; k < 4; k++) s_sc[wave][k] = mm[k]; s_sc[wave][4] = e1; }
    __syncthreads();
    int eoff = s_carry[4];
    for (int k2 = 0; k2 < wave; k2++) {
        for (int k = 0; k < 4; k++) mm[k] = max(mm[k], s_sc[k2][k]);
        eoff += s_sc[k2][4];
    }
    for (int k = 0; k < 4; k++) mm[k] = max(mm[k], s_carry[k]);
    if (emit) {
        GeneRec gr;
        gr.begin = mm[0] >= 0 ? val_b(mm[0]) : 0; gr.end = mm[1] >= 0 ? val_e(mm[1]) : 0;
        gr.start_ndx = mm[2] >= 0 ? pl[cnt - 1 - mm[2]] : 0; gr.stop_ndx = mm[3] >= 0 ? pl[cnt - 1 - mm[3]] : 0;
        out[eoff + e1 - 1] = gr;
    }
    if (r == cnt - 1) n_genes[s.contig] = eoff + e1;      // the last position of the path closes the count
}
