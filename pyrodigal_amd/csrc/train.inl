// Single-genome training on the device (ref: lib.pyx:5236-5279 GeneFinder._train, TrainingInfo._calc_dicodon_gene
// 4284-4358, _train_starts_sd 4391-4599, _train_starts_nonsd 4601-4827; Prodigal node.c record_gc_bias /
// determine_sd_usage).  Included by finder.hip: the driver continues from the device arrays that the stage-level
// runs (extraction, scoring) leave behind.  Everything per base / per node runs in kernels; every accumulation is
// a count (an exact integer in a double, so the order of additions does not matter); libm's log stays on the host,
// where the reference calls it.
namespace {

constexpr int TR_GC_HALF = 60;          // GC_WINDOW / 2 (ref: lib.pyx:171)

__device__ inline int tr_is_gc(const uint8_t* __restrict__ d, int i) { const int x = d[i]; return x != 0 && x != 3; }   // unknown bases count as GC
__device__ inline int tr_max_fr(int a, int b, int c) { return a > b ? (a > c ? 0 : 2) : (b > c ? 1 : 2); }
__device__ inline int tr_comp(int d) { return d <= 3 ? (d ^ 3) : 6; }
// 2-bit word of `len` bases starting at strand position i (ref: _sequence.h:207-220)
__device__ inline int tr_mer(const uint8_t* __restrict__ d, int L, int i, int len, int strand) {
    int v = 0;
    if (strand == 1) { for (int j = 0; j < len; j++) v |= (d[i + j] & 3) << (2 * j); }
    else { const int k = L - 1 - i; for (int j = 0; j < len; j++) v |= (tr_comp(d[k - j]) & 3) << (2 * j); }
    return v;
}

// ref: lib.pyx:724-768 (Sequence._max_gc_frame_plot).  The running sums of the reference reduce to
// tot[i] = sum of gc[i + 3 m] for |m| < 20 inside the sequence; the codon at i (i % 3 == 0) gets the frame with the most.
__global__ void __launch_bounds__(256)
k_gc_frame(const uint8_t* __restrict__ d, int L, int8_t* __restrict__ gp) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;       // codon index
    const int i = 3 * c;
    if (i >= L) return;
    if (i >= L - 2) { for (int q = i; q < L; q++) gp[q] = -1; return; }
    int tot[3];
    for (int f = 0; f < 3; f++) {
        int s = 0;
        for (int m = -(TR_GC_HALF / 3 - 1); m <= TR_GC_HALF / 3 - 1; m++) {
            const int p = i + f + 3 * m;
            if (p >= 0 && p < L) s += tr_is_gc(d, p);
        }
        tot[f] = s;
    }
    const int w = tr_max_fr(tot[0], tot[1], tot[2]);
    gp[i] = gp[i + 1] = gp[i + 2] = (int8_t)w;
}

// Prodigal node.c record_gc_bias: per start node, how often each codon position is the GC-richest one between the
// start and its stop.
__global__ void __launch_bounds__(256)
k_gc_bias(int n, const int32_t* __restrict__ ndx, const int32_t* __restrict__ stop_val, const uint8_t* __restrict__ type,
          const int8_t* __restrict__ strand, const int8_t* __restrict__ gp, double* __restrict__ gc_score, uint8_t* __restrict__ gc_bias) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    gc_score[3 * i] = gc_score[3 * i + 1] = gc_score[3 * i + 2] = 0.0; gc_bias[i] = 0;
    if (type[i] == PGA_T_STOP) return;
    int ctr[3] = {0, 0, 0};
    const int fr = ndx[i] % 3;
    if (strand[i] == 1) {
        const int fm = 3 - fr;
        for (int j = stop_val[i]; j >= ndx[i]; j -= 3) ctr[(gp[j] + fm) % 3]++;
        for (int q = 0; q < 3; q++) { double g = 3.0 * ctr[q]; g /= 1.0 * (stop_val[i] - ndx[i] + 3); gc_score[3 * i + q] = g; }
    } else {
        const int fm = fr;
        for (int j = stop_val[i]; j <= ndx[i]; j += 3) ctr[((3 - gp[j]) + fm) % 3]++;
        for (int q = 0; q < 3; q++) { double g = 3.0 * ctr[q]; g /= 1.0 * (ndx[i] - stop_val[i] + 3); gc_score[3 * i + q] = g; }
    }
    gc_bias[i] = (uint8_t)tr_max_fr(ctr[0], ctr[1], ctr[2]);
}
// the one ordered floating-point sum of the training: node order, one thread
__global__ void k_bias_sum(int n, const int32_t* __restrict__ ndx, const int32_t* __restrict__ stop_val, const uint8_t* __restrict__ type,
                           const double* __restrict__ gc_score, const uint8_t* __restrict__ gc_bias, double* __restrict__ bias) {
    double b[3] = {0.0, 0.0, 0.0};
    for (int i = 0; i < n; i++) {
        if (type[i] == PGA_T_STOP) continue;
        const int len = abs(stop_val[i] - ndx[i]) + 1;
        b[gc_bias[i]] += (gc_score[3 * i + gc_bias[i]] * len) / 1000.0;
    }
    const double tot = b[0] + b[1] + b[2];
    for (int q = 0; q < 3; q++) bias[q] = b[q] * (3.0 / tot);
}
__global__ void __launch_bounds__(256)
k_gcb(int n, const double* __restrict__ gc_score, const double* __restrict__ bias, double* __restrict__ gcb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) gcb[i] = bias[0] * gc_score[3 * i] + bias[1] * gc_score[3 * i + 1] + bias[2] * gc_score[3 * i + 2];
}

// ref: lib.pyx:2279-2329 with flag == 0: the first start of each frame met while walking away from the stop
__global__ void __launch_bounds__(256)
k_ovl_starts0(int n, const int32_t* __restrict__ ndx, const int32_t* __restrict__ stop_val, const uint8_t* __restrict__ type,
              const int8_t* __restrict__ strand, const uint8_t* __restrict__ edge, int maxov, int32_t* __restrict__ star_ptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int sp[3] = {-1, -1, -1};
    if (type[i] == PGA_T_STOP && edge[i] != 1) {
        const int me = ndx[i];
        if (strand[i] == 1) {
            for (int j = i + 3; j >= 0; j--) {
                if (j >= n || ndx[j] > me + 2) continue;
                if (ndx[j] + maxov < me) break;
                if (strand[j] != 1 || type[j] == PGA_T_STOP) continue;
                if (stop_val[j] <= me) continue;
                const int f = ndx[j] % 3;
                if (sp[f] == -1) sp[f] = j;
            }
        } else {
            for (int j = i - 3; j < n; j++) {
                if (j < 0 || ndx[j] < me - 2) continue;
                if (ndx[j] - maxov > me) break;
                if (strand[j] != -1 || type[j] == PGA_T_STOP) continue;
                if (stop_val[j] >= me) continue;
                const int f = ndx[j] % 3;
                if (sp[f] == -1) sp[f] = j;
            }
        }
    }
    star_ptr[3 * i] = sp[0]; star_ptr[3 * i + 1] = sp[1]; star_ptr[3 * i + 2] = sp[2];
}

// hexamer statistics (ref: lib.pyx:4284-4358): every window of both strands, then the codons of the genes of the path
__global__ void __launch_bounds__(256)
k_hexamer_bg(const uint8_t* __restrict__ d, int L, unsigned int* __restrict__ counts) {
    __shared__ unsigned int s_c[4096];
    for (int q = threadIdx.x; q < 4096; q += blockDim.x) s_c[q] = 0;
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L - 5; i += gridDim.x * blockDim.x) {
        atomicAdd(&s_c[tr_mer(d, L, i, 6, 1)], 1u);
        atomicAdd(&s_c[tr_mer(d, L, i, 6, -1)], 1u);
    }
    __syncthreads();
    for (int q = threadIdx.x; q < 4096; q += blockDim.x) if (s_c[q]) atomicAdd(&counts[q], s_c[q]);
}
struct TrGene { int left, right, strand; };     // strand-local [left, right - 5) step 3
__global__ void __launch_bounds__(256)
k_hexamer_genes(const uint8_t* __restrict__ d, int L, const TrGene* __restrict__ genes, int n_genes, unsigned int* __restrict__ counts) {
    const int g = blockIdx.x;
    if (g >= n_genes) return;
    const TrGene G = genes[g];
    for (int i = G.left + 3 * threadIdx.x; i < G.right - 5; i += 3 * blockDim.x) atomicAdd(&counts[tr_mer(d, L, i, 6, G.strand)], 1u);
}


// ---- start training (ref: lib.pyx:4391-4599 _train_starts_sd, 4601-4827 _train_starts_nonsd) ----------------------
struct TrWeights {            // what changes from one iteration to the next
    double rbs_wt[28], type_wt[3], st_wt, sthresh, no_mot;
    int last_iter, stage, uses_sd;
};
struct TrCounts {             // everything counted in one iteration (integers)
    unsigned int rbg[28], rreal[28], treal[3], tbg[3], ngenes, zero_bg, zero_real, _pad;
    unsigned int ups[32][4];
};
__device__ inline int tr_pick_rbs(const double* __restrict__ w, int r0, int r1) {   // ref: lib.pyx:4441-4448
    const double w0 = w[r0], w1 = w[r1];
    if (w0 > w1 + 1.0 || r1 == 0) return r0;
    if (w0 < w1 - 1.0 || r0 == 0) return r1;
    return r0 > r1 ? r0 : r1;
}
// ref: lib.pyx:4360-4389 (TrainingInfo._count_upstream_composition)
__device__ inline void tr_count_upstream(const uint8_t* __restrict__ d, int L, int pos, int strand, TrCounts* __restrict__ cn) {
    int k = 0;
    for (int pass = 0; pass < 2; pass++) {
        const int lo = pass ? 15 : 1, hi = pass ? 45 : 3;
        for (int j = lo; j < hi; j++, k++) {
            if (strand == 1) { if (pos >= j) atomicAdd(&cn->ups[k][d[pos - j] & 3], 1u); }
            else { if (pos + j < L) atomicAdd(&cn->ups[k][tr_comp(d[pos + j]) & 3], 1u); }
        }
    }
}
// background of one iteration: start types (all starts) and the RBS bin each non-edge start would pick
__global__ void __launch_bounds__(256)
k_ts_background(int n, const uint8_t* __restrict__ type, const uint8_t* __restrict__ edge, const uint8_t* __restrict__ rbs,
                const TrWeights* __restrict__ w, TrCounts* __restrict__ cn, int count_types) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || type[i] == PGA_T_STOP) return;
    if (count_types) atomicAdd(&cn->tbg[type[i]], 1u);
    if (edge[i]) return;
    atomicAdd(&cn->rbg[tr_pick_rbs(w->rbs_wt, rbs[2 * i], rbs[2 * i + 1])], 1u);
}
// One thread per stop node: the best non-edge start of its ORF under the current weights; a confident one is counted.
// The reference sweeps the nodes in strand order with ">=", i.e. among equal best starts the one met last wins:
// the highest index on the forward strand, the lowest on the reverse strand.
template <bool SD>
__global__ void __launch_bounds__(256)
k_ts_best(int n, const int32_t* __restrict__ ndx, const int32_t* __restrict__ stop_val, const uint8_t* __restrict__ type,
          const int8_t* __restrict__ strand, const uint8_t* __restrict__ edge, const double* __restrict__ cscore,
          const uint8_t* __restrict__ rbs, const double* __restrict__ mot_score, const uint8_t* __restrict__ d, int L,
          const TrWeights* __restrict__ w, TrCounts* __restrict__ cn, int32_t* __restrict__ best_of_stop) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    if (best_of_stop) best_of_stop[s] = -1;
    if (type[s] != PGA_T_STOP) return;
    const int st = strand[s], ph = ndx[s] % 3, sv = stop_val[s];
    const double wt = w->st_wt;
    double best = 0.0; int bndx = -1, brbs = 0;
    const int step = st == 1 ? -1 : 1;
    for (int j = s + step; j >= 0 && j < n; j += step) {
        if (st == 1 ? ndx[j] <= sv : ndx[j] >= sv) break;            // past the other end of the ORF
        if (strand[j] != st || ndx[j] % 3 != ph) continue;
        if (type[j] == PGA_T_STOP) break;                             // the neighbouring stop of this frame (defensive: sv marks it)
        if (edge[j]) continue;
        int mr = 0; double v;
        if (SD) { mr = tr_pick_rbs(w->rbs_wt, rbs[2 * j], rbs[2 * j + 1]); v = cscore[j] + wt * w->rbs_wt[mr] + wt * w->type_wt[type[j]]; }
        else v = cscore[j] + wt * mot_score[j] + wt * w->type_wt[type[j]];
        if (bndx == -1 ? v >= 0.0 : v > best) { best = v; bndx = j; brbs = mr; }
    }
    if (bndx == -1 || !(best >= w->sthresh)) return;
    if (SD) atomicAdd(&cn->rreal[brbs], 1u);
    else { atomicAdd(&cn->ngenes, 1u); if (best_of_stop) best_of_stop[s] = bndx; }
    atomicAdd(&cn->treal[type[bndx]], 1u);
    if (w->last_iter) tr_count_upstream(d, L, ndx[bndx], st, cn);
}

// ---- motif statistics of the non-SD training -------------------------------------------------------------------------
struct TrMotifs { int32_t* ndx; uint8_t* len; uint8_t* spacer; uint8_t* spacendx; double* score; };   // per node
__device__ inline int tr_spacer_index(int j, int start, int i) {
    if (j <= start - 16 - i) return 3;
    if (j <= start - 14 - i) return 2;
    if (j >= start - 7 - i) return 1;
    return 0;
}
// ref: lib.pyx:1556-1616 (Node._find_best_upstream_motif) with the training stages
__global__ void __launch_bounds__(256)
k_mot_best(int n, const int32_t* __restrict__ ndx, const uint8_t* __restrict__ type, const int8_t* __restrict__ strand,
           const uint8_t* __restrict__ edge, const uint8_t* __restrict__ d, int L, const double* __restrict__ mot_wt /* [4][4][4096] */,
           const TrWeights* __restrict__ w, TrMotifs m) {
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (i0 >= n || type[i0] == PGA_T_STOP || edge[i0]) return;
    const int st = strand[i0], start = st == 1 ? ndx[i0] : L - 1 - ndx[i0];
    int bsp = 0, bsi = 0, blen = 0, bndx = 0; double bsc = -100.0;
    for (int i = 3; i >= 0; i--) {
        for (int j = start - 18 - i; j < start - 5 - i; j++) {
            if (j < 0) continue;
            const int si = tr_spacer_index(j, start, i);
            const int idx = tr_mer(d, L, j, i + 3, st);
            const double sc = mot_wt[(i * 4 + si) * 4096 + idx];
            if (sc > bsc) { bsc = sc; bsi = si; bsp = start - j - i - 3; bndx = idx; blen = i + 3; }
        }
    }
    if (w->stage == 2 && (bsc == -4.0 || bsc < w->no_mot + 0.69)) {
        m.ndx[i0] = 0; m.len[i0] = 0; m.spacendx[i0] = 0; m.spacer[i0] = 0; m.score[i0] = w->no_mot;
    } else {
        m.ndx[i0] = bndx; m.len[i0] = (uint8_t)blen; m.spacendx[i0] = (uint8_t)bsi; m.spacer[i0] = (uint8_t)(bsp & 15); m.score[i0] = bsc;
    }
}
// ref: lib.pyx:4225-4282 (TrainingInfo._update_motif_counts) for node i
__device__ inline void tr_update_motif_counts(int i0, const int32_t* __restrict__ ndx, const int8_t* __restrict__ strand,
                                              const uint8_t* __restrict__ d, int L, const TrMotifs& m, int stage,
                                              unsigned int* __restrict__ cnt /* [4][4][4096] */, unsigned int* __restrict__ zero) {
    if (m.len[i0] == 0) { atomicAdd(zero, 1u); return; }
    const int st = strand[i0], start = st == 1 ? ndx[i0] : L - 1 - ndx[i0];
    const int mlen = m.len[i0];
    if (stage == 0) {
        for (int i = 3; i >= 0; i--)
            for (int j = start - 18 - i; j < start - 5 - i; j++) {
                if (j < 0) continue;
                const int mer = tr_mer(d, L, j, i + 3, st);
                for (int k = 0; k < 4; k++) atomicAdd(&cnt[(i * 4 + k) * 4096 + mer], 1u);
            }
    } else if (stage == 1) {
        atomicAdd(&cnt[((mlen - 3) * 4 + m.spacendx[i0]) * 4096 + m.ndx[i0]], 1u);
        for (int i = 0; i < mlen - 3; i++)
            for (int j = start - m.spacer[i0] - mlen; j < start - m.spacer[i0] - i - 2; j++) {
                if (j < 0) continue;
                atomicAdd(&cnt[(i * 4 + tr_spacer_index(j, start, i)) * 4096 + tr_mer(d, L, j, i + 3, st)], 1u);
            }
    } else atomicAdd(&cnt[((mlen - 3) * 4 + m.spacendx[i0]) * 4096 + m.ndx[i0]], 1u);
}
// background: every non-edge start; real: the start chosen for each confident ORF (best_of_stop from k_ts_best<false>)
__global__ void __launch_bounds__(256)
k_mot_counts(int n, const int32_t* __restrict__ ndx, const uint8_t* __restrict__ type, const int8_t* __restrict__ strand,
             const uint8_t* __restrict__ edge, const uint8_t* __restrict__ d, int L, TrMotifs m, const TrWeights* __restrict__ w,
             const int32_t* __restrict__ best_of_stop, unsigned int* __restrict__ cnt, unsigned int* __restrict__ zero) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int node = i;
    if (best_of_stop) { node = best_of_stop[i]; if (node < 0) return; }
    if (type[node] == PGA_T_STOP || edge[node] == 1) return;
    tr_update_motif_counts(node, ndx, strand, d, L, m, w->stage, cnt, zero);
}

// stages of the driver, for step-by-step validation against the oracle (PGA_TRAIN_* in the header)
enum { TR_BIAS = 1, TR_DICODON = 2, TR_SD = 3, TR_ALL = 4 };

// log-odds of the hexamer usage in genes against the whole sequence (ref: lib.pyx:4336-4358); libm on the host
void tr_gene_dc(const unsigned int* bgc, const unsigned int* gc, pga_training* t) {
    unsigned long long glob_bg = 0, glob = 0;
    for (int i = 0; i < 4096; i++) { glob_bg += bgc[i]; glob += gc[i]; }
    for (int i = 0; i < 4096; i++) {
        const double bg = (double)(int)bgc[i] / (double)(int)glob_bg;
        const double prob = (double)(int)gc[i] / (double)(int)glob;
        double v;
        if (prob == 0 && bg != 0) v = -5.0;
        else if (bg == 0) v = 0.0;
        else v = log(prob / bg);
        if (v > 5.0) v = 5.0; else if (v < -5.0) v = -5.0;
        t->gene_dc[i] = v;
    }
}


void tr_log_odds(const unsigned int* real, const double* bgv, double* out, int nq) {     // ref: lib.pyx:4528-4569
    double sum = 0.0;
    for (int j = 0; j < nq; j++) sum += (double)real[j];
    if (sum == 0.0) { for (int j = 0; j < nq; j++) out[j] = 0.0; return; }
    for (int j = 0; j < nq; j++) {
        const double r = (double)real[j] / sum;
        out[j] = bgv[j] != 0 ? log(r / bgv[j]) : -4.0;
        if (out[j] > 4.0) out[j] = 4.0; else if (out[j] < -4.0) out[j] = -4.0;
    }
}
void tr_ups_to_log(const unsigned int (*ups)[4], pga_training* t) {     // ref: lib.pyx:4571-4599 / 4797-4827
    for (int i = 0; i < 32; i++) {
        double sum = 0.0;
        for (int j = 0; j < 4; j++) { t->ups_comp[i][j] = (double)ups[i][j]; sum += t->ups_comp[i][j]; }
        if (sum == 0.0) { for (int j = 0; j < 4; j++) t->ups_comp[i][j] = 0.0; continue; }
        for (int j = 0; j < 4; j++) {
            double* u = &t->ups_comp[i][j];
            *u /= sum;
            const bool at = (j == 0 || j == 3);
            if (t->gc <= 0.1) *u = log(*u * 2.0 / (at ? 0.90 : 0.10));
            else if (t->gc >= 0.9) *u = log(*u * 2.0 / (at ? 0.10 : 0.90));
            else *u = at ? log(*u * 2.0 / (1.0 - t->gc)) : log(*u * 2.0 / t->gc);
            if (*u > 4.0) *u = 4.0;
            if (*u < -4.0) *u = -4.0;
        }
    }
}
void tr_determine_sd_usage(pga_training* t) {       // Prodigal node.c determine_sd_usage
    t->uses_sd = 1;
    if (t->rbs_wt[0] >= 0.0) t->uses_sd = 0;
    if (t->rbs_wt[16] < 1.0 && t->rbs_wt[13] < 1.0 && t->rbs_wt[15] < 1.0 &&
        (t->rbs_wt[0] >= -0.5 || (t->rbs_wt[22] < 2.0 && t->rbs_wt[24] < 2.0 && t->rbs_wt[27] < 2.0))) t->uses_sd = 0;
}


// Prodigal node.c build_coverage_map: which motifs are frequent enough (and their one-mismatch neighbours) to be modelled
void tr_build_coverage_map(const unsigned int* real /* [4][4][4096] */, int* good /* [4][4][4096] */, double ng) {
    const double thresh = 0.2;
#define RC(a, b, l) real[((a) * 4 + (b)) * 4096 + (l)]
#define GD(a, b, l) good[((a) * 4 + (b)) * 4096 + (l)]
    memset(good, 0, sizeof(int) * 4 * 4 * 4096);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 64; j++)
        if ((double)RC(0, i, j) / ng >= thresh) for (int k = 0; k < 4; k++) GD(0, k, j) = 1;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 256; j++) {
        const int d0 = (j & 252) >> 2, d1 = j & 63;
        if (GD(0, i, d0) == 0 || GD(0, i, d1) == 0) continue;
        GD(1, i, j) = 1;
    }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 1024; j++) {
        const int d0 = (j & 1008) >> 4, d1 = (j & 252) >> 2, d2 = j & 63;
        if (GD(0, i, d0) == 0 || GD(0, i, d1) == 0 || GD(0, i, d2) == 0) continue;
        GD(2, i, j) = 1;
        int tmp = j;
        for (int k = 0; k <= 16; k += 16) {
            tmp ^= k;
            for (int l = 0; l <= 32; l += 32) { tmp ^= l; if (GD(2, i, tmp) == 0) GD(2, i, tmp) = 2; }
        }
    }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4096; j++) {
        const int d0 = (j & 4092) >> 2, d1 = j & 1023;
        if (GD(2, i, d0) == 0 || GD(2, i, d1) == 0) continue;
        GD(3, i, j) = (GD(2, i, d0) == 1 && GD(2, i, d1) == 1) ? 1 : 2;
    }
#undef RC
#undef GD
}

}  // namespace

static int find_impl(pga_ctx* c, const pga_batch* batch, const pga_params* pp, const int stage, const int tt_override, pga_result** out);

// ref: lib.pyx:5236-5279 (GeneFinder._train) for ONE sequence (the host layer joins several with the reference's spacer)
static int train_body(pga_ctx* c, const pga_batch* batch, const pga_params* pp, int tt, double start_weight, int force_nonsd,
                      int upto, pga_training* t, bool& models_replaced) {
    if (!c || !batch || !pp || !t || batch->ctx != c || batch->n != 1) { if (c) c->err = "pga_train: needs a batch of exactly one sequence"; return PGA_EINVAL; }
    memset(t, 0, sizeof *t);
    t->trans_table = tt; t->st_wt = start_weight; t->uses_sd = 1;
    pga_params P = *pp; P.meta = 0; P.want_nodes = 1;
    hipStream_t st = c->stream;
    // ---- nodes (ref: lib.pyx:5252-5257): extraction + sort through the stage-level path; its device arrays stay
    pga_result* r1 = nullptr;
    if (int rc = find_impl(c, batch, &P, PGA_STAGE_EXTRACT, tt, &r1)) return rc;
    struct Free { pga_result* r; ~Free() { if (r) pga_result_free(r); } } fr1{r1};
    t->gc = r1->contigs[0].gc;
    const int n = r1->nodes[0].n, L = batch->ct[0].len;
    if (n == 0) { c->err = "pga_train: no start / stop node in the sequence"; return PGA_EINVAL; }
    const pga_nodes& H = r1->nodes[0];                 // host copies of the topology
    FinderState* f = c->finder;
    const LastRun lr = f->last;
    const GroupArrays& ga = lr.ga;
    // ---- GC frame bias (ref: lib.pyx:5259-5261)
    DEVBUF(d_gp, int8_t, "tr_gp", L + 4);
    DEVBUF(d_gcs, double, "tr_gc_score", 3 * (size_t)n + 3);
    DEVBUF(d_gcbias, uint8_t, "tr_gc_bias", n + 1);
    DEVBUF(d_bias, double, "tr_bias", 4);
    DEVBUF(d_gcb, double, "tr_gcb", n + 1);
    DEVBUF(d_star, int32_t, "tr_star", 3 * (size_t)n + 3);
    const int nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_gc_frame, dim3((L / 3 + 256) / 256), dim3(256), 0, st, lr.d_dig, L, d_gp);
    hipLaunchKernelGGL(k_gc_bias, dim3(nb), dim3(256), 0, st, n, ga.ndx, ga.stop_val, ga.type, ga.strand, d_gp, d_gcs, d_gcbias);
    hipLaunchKernelGGL(k_bias_sum, dim3(1), dim3(1), 0, st, n, ga.ndx, ga.stop_val, ga.type, d_gcs, d_gcbias, d_bias);
    HT(c, hipMemcpyAsync(t->bias, d_bias, sizeof(double) * 3, hipMemcpyDeviceToHost, st));
    HT(c, hipGetLastError());
    HT(c, hipStreamSynchronize(st));
    if (upto == TR_BIAS) return PGA_OK;
    // ---- training pass of the dynamic programme (ref: lib.pyx:5263-5267)
    hipLaunchKernelGGL(k_gcb, dim3(nb), dim3(256), 0, st, n, d_gcs, d_bias, d_gcb);
    hipLaunchKernelGGL(k_ovl_starts0, dim3(nb), dim3(256), 0, st, n, ga.ndx, ga.stop_val, ga.type, ga.strand, ga.edge0, P.max_overlap, d_star);
    DpBuffers dp;
    {
        DEVBUF(b0, DpSrc, "dp_src", n + 1) DEVBUF(b1, DpTgt, "dp_tgt", n + 1) DEVBUF(b2, double, "dp_score", n + 1) DEVBUF(b3, int32_t, "dp_traceb", n + 1)
        DEVBUF(b4, int32_t, "dp_tbn", n + 1) DEVBUF(b5, int8_t, "dp_ov", n + 1) DEVBUF(b6, int32_t, "dp_maxidx", 2) DEVBUF(b7, double, "dp_maxscore", 2)
        DEVBUF(b8, int32_t, "dp_ipath", 2)
        dp = DpBuffers{b0, b1, b2, b3, b4, b5, b6, b7, b8, nullptr, {nullptr, nullptr, nullptr}, nullptr, nullptr, nullptr};
    }
    DEVBUF(d_chain, ChainDesc, "tr_chain", 2);
    DEVBUF(d_mc, ModelConst, "tr_mc", 2);
    ChainDesc ch{0, 0, n, 0, 0, 1};
    ModelConst mc; pga_fill_model_const(&mc, start_weight);
    HT(c, hipMemcpyAsync(d_chain, &ch, sizeof ch, hipMemcpyHostToDevice, st));
    HT(c, hipMemcpyAsync(d_mc, &mc, sizeof mc, hipMemcpyHostToDevice, st));
    NodeArrays na{ga.ndx, ga.stop_val, ga.type, ga.strand, d_gcb, d_gcb, d_gcb, d_gcb, d_star, d_gcb};   // scores are not read when final = 0
    pga_launch_dp_prepare(d_chain, 1, 0, n, na, d_mc, dp, st, 0);
    pga_launch_dp(d_chain, 1, d_mc, dp, 0, st);
    std::vector<int32_t> traceb((size_t)n), tracef((size_t)n, -1), star((size_t)3 * n), path((size_t)n + 1);
    std::vector<int8_t> ovm((size_t)n);
    std::vector<uint8_t> elim((size_t)n, 0);
    int32_t mx = -1;
    HT(c, hipMemcpyAsync(traceb.data(), dp.traceb, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    HT(c, hipMemcpyAsync(ovm.data(), dp.ov_mark, n, hipMemcpyDeviceToHost, st));
    HT(c, hipMemcpyAsync(star.data(), d_star, sizeof(int32_t) * 3 * n, hipMemcpyDeviceToHost, st));
    HT(c, hipMemcpyAsync(&mx, dp.max_index, 4, hipMemcpyDeviceToHost, st));
    HT(c, hipGetLastError());
    HT(c, hipStreamSynchronize(st));
    // ---- the genes of the best path (ref: lib.pyx:1253-1311 untangling; 4299-4334 walk from the path's end)
    std::vector<TrGene> genes;
    if (mx >= 0) {
        NodeView v{n, H.ndx, H.stop_val, H.type, H.strand, H.edge, nullptr, nullptr, nullptr, nullptr, nullptr, star.data(), traceb.data(),
                   tracef.data(), ovm.data(), nullptr, elim.data()};
        untangle(v, mx, path.data());
        const int ipath = traceb[mx] == -1 ? -1 : mx;
        int in_gene = 0, left = -1, right = -1;
        for (int p = ipath; p != -1; p = traceb[p]) {
            if (H.strand[p] == 1) {
                if (H.type[p] == PGA_T_STOP) { in_gene = 1; right = H.ndx[p] + 2; }
                else if (in_gene == 1) { left = H.ndx[p]; genes.push_back(TrGene{left, right, 1}); in_gene = 0; }
            } else {
                if (H.type[p] != PGA_T_STOP) { in_gene = -1; left = L - H.ndx[p] - 1; }
                else if (in_gene == -1) { right = L - H.ndx[p] + 1; genes.push_back(TrGene{left, right, -1}); in_gene = 0; }
            }
        }
    }
    // ---- hexamer statistics (ref: lib.pyx:5269, 4284-4358)
    DEVBUF(d_cnt, unsigned int, "tr_hex", 2 * 4096);
    DEVBUF(d_genes, TrGene, "tr_genes", genes.size() + 1);
    HT(c, hipMemsetAsync(d_cnt, 0, sizeof(unsigned int) * 2 * 4096, st));
    if (!genes.empty()) HT(c, hipMemcpyAsync(d_genes, genes.data(), sizeof(TrGene) * genes.size(), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_hexamer_bg, dim3(1024), dim3(256), 0, st, lr.d_dig, L, d_cnt);
    if (!genes.empty()) hipLaunchKernelGGL(k_hexamer_genes, dim3((unsigned)genes.size()), dim3(64), 0, st, lr.d_dig, L, d_genes, (int)genes.size(), d_cnt + 4096);
    std::vector<unsigned int> cnt(2 * 4096);
    HT(c, hipMemcpyAsync(cnt.data(), d_cnt, sizeof(unsigned int) * 2 * 4096, hipMemcpyDeviceToHost, st));
    HT(c, hipGetLastError());
    HT(c, hipStreamSynchronize(st));
    tr_gene_dc(cnt.data(), cnt.data() + 4096, t);
    if (upto == TR_DICODON) return PGA_OK;
    // ---- coding scores and RBS bins under the new statistics (ref: lib.pyx:5271-5273): the scoring stage of the path
    {
        const pga_training* tp = t;
        models_replaced = true;
        if (int rc = pga_set_models(c, &tp, 1)) return rc;
    }
    pga_result* r2 = nullptr;
    if (int rc = find_impl(c, batch, &P, PGA_STAGE_SCORE, tt, &r2)) return rc;
    Free fr2{r2};
    if (r2->nodes[0].n != n) { c->err = "pga_train: node count changed between the stages"; return PGA_EDEVICE; }
    f = c->finder;
    const LastRun ls = f->last;                       // topology (again) + chain arrays of the scoring stage
    const GroupArrays& gs = ls.ga;
    DEVBUF(d_w, TrWeights, "tr_weights", 2);
    DEVBUF(d_cn, TrCounts, "tr_counts", 2);
    TrWeights w; memset(&w, 0, sizeof w);
    w.st_wt = t->st_wt; w.sthresh = 35.0; w.uses_sd = 1;
    TrCounts cn;
    double tbg[3] = {0, 0, 0};
    memset(t->type_wt, 0, sizeof t->type_wt); memset(t->rbs_wt, 0, sizeof t->rbs_wt); memset(t->ups_comp, 0, sizeof t->ups_comp);
    // ---- Shine-Dalgarno start training: 10 rounds (ref: lib.pyx:4391-4599)
    for (int it = 0; it < 10; it++) {
        memcpy(w.rbs_wt, t->rbs_wt, sizeof w.rbs_wt); memcpy(w.type_wt, t->type_wt, sizeof w.type_wt);
        w.last_iter = it == 9;
        HT(c, hipMemcpyAsync(d_w, &w, sizeof w, hipMemcpyHostToDevice, st));
        HT(c, hipMemsetAsync(d_cn, 0, sizeof(TrCounts), st));
        hipLaunchKernelGGL(k_ts_background, dim3(nb), dim3(256), 0, st, n, gs.type, gs.edge0, ls.ca.rbs, d_w, d_cn, 1);
        hipLaunchKernelGGL(k_ts_best<true>, dim3(nb), dim3(256), 0, st, n, gs.ndx, gs.stop_val, gs.type, gs.strand, gs.edge0, ls.ca.cscore,
                           ls.ca.rbs, (const double*)nullptr, ls.d_dig, L, d_w, d_cn, (int32_t*)nullptr);
        HT(c, hipMemcpyAsync(&cn, d_cn, sizeof cn, hipMemcpyDeviceToHost, st));
        HT(c, hipGetLastError());
        HT(c, hipStreamSynchronize(st));
        if (it == 0) {
            double sum = 0.0;
            for (int j = 0; j < 3; j++) { tbg[j] = (double)cn.tbg[j]; sum += tbg[j]; }
            for (int j = 0; j < 3; j++) tbg[j] /= sum;
        }
        double rbg[28], sum = 0.0;
        for (int j = 0; j < 28; j++) { rbg[j] = (double)cn.rbg[j]; sum += rbg[j]; }
        for (int j = 0; j < 28; j++) rbg[j] /= sum;
        tr_log_odds(cn.rreal, rbg, t->rbs_wt, 28);
        sum = 0.0; for (int j = 0; j < 3; j++) sum += (double)cn.treal[j];
        tr_log_odds(cn.treal, tbg, t->type_wt, 3);
        if (sum * 2000.0 <= n) w.sthresh /= 2.0;
    }
    tr_ups_to_log(cn.ups, t);
    if (force_nonsd) t->uses_sd = 0; else tr_determine_sd_usage(t);
    if (upto == TR_SD || t->uses_sd) return PGA_OK;
    // ---- motif-based start training: 20 rounds in three stages (ref: lib.pyx:4601-4827)
    const size_t MT = (size_t)4 * 4 * 4096;
    DEVBUF(d_motwt, double, "tr_mot_wt", MT);
    DEVBUF(d_mcnt, unsigned int, "tr_mot_counts", 2 * MT + 8);
    DEVBUF(d_best, int32_t, "tr_best_of_stop", n + 1);
    TrMotifs mot;
    {
        DEVBUF(m0, int32_t, "tr_mot_ndx", n + 1) DEVBUF(m1, uint8_t, "tr_mot_len", n + 1) DEVBUF(m2, uint8_t, "tr_mot_spacer", n + 1)
        DEVBUF(m3, uint8_t, "tr_mot_spacendx", n + 1) DEVBUF(m4, double, "tr_mot_score", n + 1)
        mot = TrMotifs{m0, m1, m2, m3, m4};
        HT(c, hipMemsetAsync(m1, 0, (size_t)n + 1, st));
    }
    std::vector<unsigned int> hc(2 * MT);
    std::vector<double> mbg(MT), mreal(MT);
    std::vector<int> mgood(MT, 0);
    memset(t->ups_comp, 0, sizeof t->ups_comp);
    memset(t->type_wt, 0, sizeof t->type_wt);
    w.sthresh = 35.0; w.uses_sd = 0;
    for (int it = 0; it < 20; it++) {
        const int stage = it < 4 ? 0 : (it < 12 ? 1 : 2);
        memcpy(w.type_wt, t->type_wt, sizeof w.type_wt);
        w.no_mot = t->no_mot; w.stage = stage; w.last_iter = it == 19;
        HT(c, hipMemcpyAsync(d_w, &w, sizeof w, hipMemcpyHostToDevice, st));
        HT(c, hipMemcpyAsync(d_motwt, &t->mot_wt[0][0][0], sizeof(double) * MT, hipMemcpyHostToDevice, st));
        HT(c, hipMemsetAsync(d_cn, 0, sizeof(TrCounts), st));
        HT(c, hipMemsetAsync(d_mcnt, 0, sizeof(unsigned int) * (2 * MT + 8), st));
        unsigned int* d_zero = d_mcnt + 2 * MT;           // [0] background, [1] real
        hipLaunchKernelGGL(k_mot_best, dim3(nb), dim3(256), 0, st, n, gs.ndx, gs.type, gs.strand, gs.edge0, ls.d_dig, L, d_motwt, d_w, mot);
        hipLaunchKernelGGL(k_mot_counts, dim3(nb), dim3(256), 0, st, n, gs.ndx, gs.type, gs.strand, gs.edge0, ls.d_dig, L, mot, d_w,
                           (const int32_t*)nullptr, d_mcnt, d_zero);
        hipLaunchKernelGGL(k_ts_best<false>, dim3(nb), dim3(256), 0, st, n, gs.ndx, gs.stop_val, gs.type, gs.strand, gs.edge0, ls.ca.cscore,
                           ls.ca.rbs, mot.score, ls.d_dig, L, d_w, d_cn, d_best);
        hipLaunchKernelGGL(k_mot_counts, dim3(nb), dim3(256), 0, st, n, gs.ndx, gs.type, gs.strand, gs.edge0, ls.d_dig, L, mot, d_w,
                           d_best, d_mcnt + MT, d_zero + 1);
        unsigned int zeros[2];
        HT(c, hipMemcpyAsync(hc.data(), d_mcnt, sizeof(unsigned int) * 2 * MT, hipMemcpyDeviceToHost, st));
        HT(c, hipMemcpyAsync(zeros, d_zero, sizeof zeros, hipMemcpyDeviceToHost, st));
        HT(c, hipMemcpyAsync(&cn, d_cn, sizeof cn, hipMemcpyDeviceToHost, st));
        HT(c, hipGetLastError());
        HT(c, hipStreamSynchronize(st));
        // ---- the weights of the next round (host: sums of counts, libm log)
        double zbg = (double)zeros[0], zreal = (double)zeros[1], sum = 0.0;
        const double ngenes = (double)cn.ngenes;
        for (size_t q = 0; q < MT; q++) { mbg[q] = (double)hc[q]; sum += mbg[q]; }
        sum += zbg;
        for (size_t q = 0; q < MT; q++) mbg[q] /= sum;
        zbg /= sum;
        if (stage < 2) tr_build_coverage_map(hc.data() + MT, mgood.data(), ngenes);
        sum = 0.0;
        for (size_t q = 0; q < MT; q++) { mreal[q] = (double)hc[MT + q]; sum += mreal[q]; }
        sum += zreal;
        if (sum == 0.0) {
            memset(t->mot_wt, 0, sizeof t->mot_wt); t->no_mot = 0.0;
        } else {
            double* wt = &t->mot_wt[0][0][0];
            for (size_t q = 0; q < MT; q++) {
                if (mgood[q] == 0) { zreal += mreal[q]; zbg += mreal[q]; mreal[q] = 0.0; mbg[q] = 0.0; }
                mreal[q] /= sum;
                double v = mbg[q] != 0 ? log(mreal[q] / mbg[q]) : -4.0;
                if (v > 4.0) v = 4.0; else if (v < -4.0) v = -4.0;
                wt[q] = v;
            }
        }
        zreal /= sum;
        t->no_mot = zbg != 0 ? log(zreal / zbg) : -4.0;
        if (t->no_mot > 4.0) t->no_mot = 4.0; else if (t->no_mot < -4.0) t->no_mot = -4.0;
        sum = 0.0; for (int j = 0; j < 3; j++) sum += (double)cn.treal[j];
        tr_log_odds(cn.treal, tbg, t->type_wt, 3);
        if (sum * 2000.0 <= n) w.sthresh /= 2.0;
    }
    tr_ups_to_log(cn.ups, t);
    return PGA_OK;
}

// The scoring stage of the training runs through the context's model slot (the half-trained model is loaded as model 0);
// the caller's model set is put back afterwards, so that pga_set_models(bins) ... pga_train ... pga_find_genes keeps
// scoring with the bins.
static int train_impl(pga_ctx* c, const pga_batch* batch, const pga_params* pp, int tt, double start_weight, int force_nonsd,
                      int upto, pga_training* t) {
    if (!c) return PGA_EINVAL;
    const std::vector<pga_training> saved = c->models;
    bool replaced = false;
    const int rc = train_body(c, batch, pp, tt, start_weight, force_nonsd, upto, t, replaced);
    if (replaced) {
        const std::string err = c->err;
        std::vector<const pga_training*> ptrs;
        for (const pga_training& m : saved) ptrs.push_back(&m);
        const int rc2 = pga_set_models(c, ptrs.data(), (int)ptrs.size());
        // a failed restore must not pass unseen: the context would keep scoring with the half-trained model
        if (rc != PGA_OK) { c->err = rc2 != PGA_OK ? err + "; and the context's models could not be restored: " + c->err : err; return rc; }
        if (rc2 != PGA_OK) return rc2;
    }
    return rc;
}
