// Device-side translation of gene records into proteins: one thread per codon.
// ref: lib.pyx:2932-3047 (Gene.translate), 770-789 (Sequence._amino), _sequence.h:19-73 (stop / start codons per table),
// _translation.h:4-42 (the genetic codes; restated here from the NCBI tables in TCAG order and re-indexed by the digit
// alphabet of this library, A0 G1 C2 T3).
#include "pga_internal.h"
#include <mutex>
#include "pipeline.h"

#include <string.h>

#include <vector>

struct pga_batch_view { pga_ctx* ctx; int32_t n; int64_t total; const ContigDesc* ct; const char* d_seq; };
pga_batch_view pga_batch_peek(const pga_batch*);      // finder.hip

namespace {

// NCBI genetic codes, 64 codons in TCAG order (first base slowest)
struct Code { int tt; const char* aa; };
const Code NCBI[] = {
    {1, "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"}, {2, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSS**VVVVAAAADDEEGGGG"},
    {3, "FFLLSSSSYY**CCWWTTTTPPPPHHQQRRRRIIMMTTTTNNKKSSRRVVVVAAAADDEEGGGG"}, {4, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"},
    {5, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSSSVVVVAAAADDEEGGGG"}, {6, "FFLLSSSSYYQQCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"},
    {9, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG"}, {10, "FFLLSSSSYY**CCCWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"},
    {11, "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"}, {12, "FFLLSSSSYY**CC*WLLLSPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"},
    {13, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSGGVVVVAAAADDEEGGGG"}, {14, "FFLLSSSSYYY*CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG"},
    {15, "FFLLSSSSYY*QCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"}, {16, "FFLLSSSSYY*LCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"},
    {21, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNNKSSSSVVVVAAAADDEEGGGG"}, {22, "FFLLSS*SYY*LCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"},
    {23, "FF*LSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"}, {24, "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSSKVVVVAAAADDEEGGGG"},
    {25, "FFLLSSSSYY**CCGWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"}, {26, "FFLLSSSSYY**CC*WLLLAPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"},
    {29, "FFLLSSSSYYYYCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"}, {30, "FFLLSSSSYYEECC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"},
    {32, "FFLLSSSSYY*WCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"}, {33, "FFLLSSSSYYY*CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSSKVVVVAAAADDEEGGGG"},
};
__constant__ char c_code[34][64];      // [table][a << 4 | b << 2 | c] with A0 G1 C2 T3; all-zero rows = unknown tables
__constant__ unsigned char c_known[34];

__device__ __forceinline__ bool tt_in(const int tt, const unsigned long long set) { return (set >> tt) & 1ull; }
#define TTS(...) tts_of({__VA_ARGS__})
__device__ __host__ constexpr unsigned long long tts_of(std::initializer_list<int> l) { unsigned long long m = 0; for (int t : l) m |= 1ull << t; return m; }

// ref: _sequence.h:19-43
__device__ __forceinline__ bool codon_stop(const int x0, const int x1, const int x2, const int tt) {
    if (x0 == 0 && tt == 2) return x1 == 1 && (x2 == 0 || x2 == 1);                                   // AGA / AGG
    if (x0 != 3) return false;
    if (x1 == 0 && x2 == 1) return tt_in(tt, TTS(1, 2, 3, 4, 5, 9, 10, 11, 12, 13, 14, 21, 23, 24, 25, 26, 33));     // TAG
    if (x1 == 1 && x2 == 0) return tt_in(tt, TTS(1, 6, 11, 12, 15, 16, 22, 23, 26, 29, 30, 32));                      // TGA
    if (x1 == 0 && x2 == 0) return tt_in(tt, TTS(1, 2, 3, 4, 5, 9, 10, 11, 12, 13, 15, 16, 21, 22, 23, 24, 25, 26, 32));   // TAA
    if (tt == 22) return x1 == 2 && x2 == 0;                                                            // TCA
    if (tt == 23) return x1 == 3 && x2 == 0;                                                            // TTA
    return false;
}
// ref: _sequence.h:45-73
__device__ __forceinline__ bool codon_start(const int x0, const int x1, const int x2, const int tt) {
    if (x1 != 3 || x2 != 1) return false;
    if (x0 == 0) return true;
    if (tt_in(tt, TTS(6, 10, 14, 15, 16, 2))) return false;
    if (x0 == 1) return !(tt == 1 || tt == 3 || tt == 12 || tt == 2);
    if (x0 == 3) return !(tt < 4 || tt == 9 || (tt >= 21 && tt < 25));
    return false;
}
__device__ __forceinline__ int digit_of(const int ch, const bool comp) {
    int d;
    switch (ch) { case 'A': case 'a': d = 0; break; case 'G': case 'g': d = 1; break; case 'C': case 'c': d = 2; break;
                  case 'T': case 't': d = 3; break; default: return 6; }
    return comp ? 3 - d : d;            // A <-> T, G <-> C
}

__global__ void __launch_bounds__(256)
k_translate(const char* __restrict__ seq, const ContigDesc* __restrict__ ct, const pga_gene* __restrict__ genes, const int64_t n_genes,
            const int32_t* __restrict__ tt_of, const int64_t* __restrict__ off, const int unk, const int include_stop, const int strict,
            char* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= off[n_genes]) return;
    int64_t lo = 0, hi = n_genes - 1;
    while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if (off[mid] <= idx) lo = mid; else hi = mid - 1; }
    const pga_gene g = genes[lo];
    const int i = (int)(idx - off[lo]);
    const ContigDesc cd = ct[g.contig];
    const char* __restrict__ s = seq + cd.base;
    const int tt = tt_of[g.contig];
    int x0, x1, x2;
    if (g.strand == 1) {
        const int p = g.begin - 1 + 3 * i;
        x0 = digit_of(s[p], false); x1 = digit_of(s[p + 1], false); x2 = digit_of(s[p + 2], false);
    } else {
        const int p = g.end - 1 - 3 * i;
        x0 = digit_of(s[p], true); x1 = digit_of(s[p - 1], true); x2 = digit_of(s[p - 2], true);
    }
    // partial flags are in sequence orientation; the gene's own first codon follows its strand
    const bool start_edge = g.strand == 1 ? g.partial_begin : g.partial_end;
    int aa;
    if (x0 <= 3 && x1 <= 3 && x2 <= 3) {
        if (codon_stop(x0, x1, x2, tt)) aa = '*';
        else if (i == 0 && !start_edge && codon_start(x0, x1, x2, tt)) aa = 'M';
        else aa = c_code[tt][(x0 << 4) + (x1 << 2) + x2];
    } else {
        aa = 'X';
        if (!strict && x0 <= 3 && (x1 <= 3) != (x2 <= 3)) {
            // one unknown base in second or third position: unambiguous when all four completions agree
            aa = c_code[tt][(x0 << 4) + ((x1 <= 3 ? x1 : 0) << 2) + (x2 <= 3 ? x2 : 0)];
            for (int y = 1; y < 4; y++)
                if (c_code[tt][(x0 << 4) + ((x1 <= 3 ? x1 : y) << 2) + (x2 <= 3 ? x2 : y)] != aa) { aa = 'X'; break; }
        }
    }
    out[idx] = (char)(aa == 'X' ? unk : aa);
}

// the tables are __constant__ symbols: one copy per DEVICE, so readiness is tracked per device (several GPUs may be driven
// from one process) and the first use on a device uploads them under a lock
std::mutex g_tables_mu;
bool g_tables_ready[64] = {false};
int upload_tables() {
    static char code[34][64];
    static unsigned char known[34];
    memset(code, 0, sizeof code); memset(known, 0, sizeof known);
    const int ncbi_of_digit[4] = {2, 3, 1, 0};        // digit (A G C T) -> position in TCAG
    for (const Code& c : NCBI) {
        known[c.tt] = 1;
        for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) for (int d = 0; d < 4; d++)
            code[c.tt][(a << 4) + (b << 2) + d] = c.aa[ncbi_of_digit[a] * 16 + ncbi_of_digit[b] * 4 + ncbi_of_digit[d]];
    }
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_code), code, sizeof code) != hipSuccess) return PGA_EDEVICE;
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_known), known, sizeof known) != hipSuccess) return PGA_EDEVICE;
    return PGA_OK;
}
bool table_known(const int tt) { for (const Code& c : NCBI) if (c.tt == tt) return true; return false; }

}  // namespace

extern "C" int pga_translate_genes(pga_ctx* c, const pga_batch* batch, int64_t n_genes, const pga_gene* genes, const int32_t* table_of_contig,
                                   int unknown_residue, int include_stop, int strict, const int64_t* offsets, char* out) {
    if (!c || !batch || n_genes < 0 || (n_genes > 0 && (!genes || !table_of_contig || !offsets || !out))) { if (c) c->err = "pga_translate_genes: bad arguments"; return PGA_EINVAL; }
    const pga_batch_view bv = pga_batch_peek(batch);
    if (bv.ctx != c) { c->err = "pga_translate_genes: the batch belongs to another context"; return PGA_EINVAL; }
    if (n_genes == 0) return PGA_OK;
    if (unknown_residue <= 0 || unknown_residue > 127) { c->err = "pga_translate_genes: `unknown_residue` must be a single ASCII character"; return PGA_EINVAL; }
    for (int i = 0; i < bv.n; i++)
        if (!table_known(table_of_contig[i])) { c->err = "pga_translate_genes: not a valid translation table index"; return PGA_EINVAL; }
    // the caller's layout must be the one the kernel writes (ref: lib.pyx:3006-3018 for the lengths)
    if (offsets[0] != 0) { c->err = "pga_translate_genes: offsets[0] must be 0"; return PGA_EINVAL; }
    for (int64_t g = 0; g < n_genes; g++) {
        const pga_gene& G = genes[g];
        if (G.contig < 0 || G.contig >= bv.n || G.begin < 1 || G.end > bv.ct[G.contig].len || G.end < G.begin) { c->err = "pga_translate_genes: gene outside its contig"; return PGA_EINVAL; }
        const bool stop_edge = G.strand == 1 ? G.partial_end : G.partial_begin;
        const int64_t want = (G.end - G.begin + 1) / 3 - ((!stop_edge && !include_stop) ? 1 : 0);
        if (offsets[g + 1] - offsets[g] != (want > 0 ? want : 0)) { c->err = "pga_translate_genes: offsets do not match the gene lengths"; return PGA_EINVAL; }
    }
    const int64_t total = offsets[n_genes];
    if (total == 0) return PGA_OK;
    if (hipSetDevice(c->device) != hipSuccess) return PGA_EDEVICE;
    {
        std::lock_guard<std::mutex> lk(g_tables_mu);
        const int dev = c->device & 63;
        if (!g_tables_ready[dev]) { const int rc = upload_tables(); if (rc) return rc; g_tables_ready[dev] = true; }
    }
    pga_gene* d_genes = nullptr; int32_t* d_tt = nullptr; int64_t* d_off = nullptr; char* d_out = nullptr; ContigDesc* d_ct = nullptr;
    auto cleanup = [&]() { hipFree(d_genes); hipFree(d_tt); hipFree(d_off); hipFree(d_out); hipFree(d_ct); };
    hipStream_t st = c->stream;
    hipError_t e = hipMalloc((void**)&d_genes, sizeof(pga_gene) * (size_t)n_genes);
    if (e == hipSuccess) e = hipMalloc((void**)&d_tt, sizeof(int32_t) * (size_t)bv.n);
    if (e == hipSuccess) e = hipMalloc((void**)&d_off, sizeof(int64_t) * (size_t)(n_genes + 1));
    if (e == hipSuccess) e = hipMalloc((void**)&d_out, (size_t)total);
    if (e == hipSuccess) e = hipMalloc((void**)&d_ct, sizeof(ContigDesc) * (size_t)(bv.n + 1));
    if (e == hipSuccess) e = hipMemcpyAsync(d_genes, genes, sizeof(pga_gene) * (size_t)n_genes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tt, table_of_contig, sizeof(int32_t) * (size_t)bv.n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, offsets, sizeof(int64_t) * (size_t)(n_genes + 1), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_ct, bv.ct, sizeof(ContigDesc) * (size_t)(bv.n + 1), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_translate, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, bv.d_seq, d_ct, d_genes, n_genes, d_tt, d_off,
                           unknown_residue, include_stop, strict, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)total, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    cleanup();
    return pga_hip_try_(c, e, "pga_translate_genes");
}
