// Sequence digitising, node extraction (Prodigal add_nodes) and node scoring (score_nodes,
// record_overlapping_starts) kernels for gfx950.  All "ref:" citations are relative to
// /root/reference/src/pyrodigal.
//
// Parallel decomposition (the reference is a chain of sequential scans):
//   * extraction: an ORF (the codons between two in-frame stops) is independent of every other
//     ORF.  Each stop codon (plus three virtual stops at the sequence end) "owns" the ORF to its
//     left in the scan direction and walks it once, applying the per-frame state machine of
//     Nodes._extract to that ORF alone.  Nodes are flagged per position; an exclusive prefix sum
//     over the flags gives every node its index in (ndx, strand) order directly -- no sort.
//   * coding score: the hexamer log-odds sum is an ordered floating-point sum from the stop
//     outwards, so one thread per stop node walks its ORF and adds in the reference's order.
//   * everything else (RBS / motif search, upstream composition, start scoring, overlapping
//     starts) is one thread per node.
// These kernels are byte/integer work with a few ordered f64 sums: no MFMA, coalesced reads of
// the 1-byte digit array, per-node SoA writes.

#include "pga_internal.h"
#include "pipeline.h"
#include <atomic>
#include "dev_common.h"
#include "dpw_core.h"

namespace {

enum { NA = 0, NG = 1, NC = 2, NT = 3, NN = 6 };   // ref: _sequence.h:8-14

__device__ __forceinline__ int find_contig(const ContigDesc* __restrict__ ct, int n, int64_t g) {
    int lo = 0, hi = n - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (ct[mid].base <= g) lo = mid; else hi = mid - 1; }
    return lo;
}

// Contig / chain of element g for a block whose first element is block_first: one binary search per
// block (thread 0), then a short forward walk per thread (blocks rarely span more than two contigs).
__device__ __forceinline__ int block_contig(const ContigDesc* __restrict__ ct, int n, int64_t block_first, int64_t g, int* s_slot) {
    int c = block_search_le([&](const int k) { return ct[k].base; }, n, block_first, s_slot);
    while (c + 1 < n && ct[c + 1].base <= g) c++;
    return c;
}
__device__ __forceinline__ int block_chain(const ChainDesc* __restrict__ ch, int n, int64_t block_first, int64_t g, int* s_slot) {
    int c = block_search_le([&](const int k) { return ch[k].off; }, n, block_first, s_slot);
    while (c + 1 < n && ch[c + 1].off <= g) c++;
    return c;
}

__device__ __forceinline__ double readlane_f64_pl(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// strand-local base i of a contig (reverse strand is virtual; ref: _sequence.h:45-55)
__device__ __forceinline__ int sbase(const uint8_t* __restrict__ d, int L, int i, int strand) {
    return strand == 1 ? d[i] : (d[L - 1 - i] ^ 3);
}
__device__ __forceinline__ int comp2(int d) { return d <= 3 ? (d ^ 3) : NN; }
__device__ __forceinline__ int is_gc_n(int d) { return d != NA && d != NT; }   // ref: _sequence.h:35-43

// ref: _sequence.h:117-157
__host__ __device__ __forceinline__ bool codon_is_stop(int x0, int x1, int x2, int tt) {
    if (x0 != NT) {
        if (tt == 2) return x0 == NA && x1 == NG && (x2 == NA || x2 == NG);
        return false;
    }
    // sets of translation tables in which TAA / TAG / TGA terminate
    const unsigned long long taa = (1ULL<<1)|(1ULL<<2)|(1ULL<<3)|(1ULL<<4)|(1ULL<<5)|(1ULL<<9)|(1ULL<<10)|(1ULL<<11)|(1ULL<<12)|(1ULL<<13)
                                 | (1ULL<<15)|(1ULL<<16)|(1ULL<<21)|(1ULL<<22)|(1ULL<<23)|(1ULL<<24)|(1ULL<<25)|(1ULL<<26)|(1ULL<<32);
    const unsigned long long tag = (1ULL<<1)|(1ULL<<2)|(1ULL<<3)|(1ULL<<4)|(1ULL<<5)|(1ULL<<9)|(1ULL<<10)|(1ULL<<11)|(1ULL<<12)|(1ULL<<13)
                                 | (1ULL<<14)|(1ULL<<21)|(1ULL<<23)|(1ULL<<24)|(1ULL<<25)|(1ULL<<26)|(1ULL<<33);
    const unsigned long long tga = (1ULL<<1)|(1ULL<<6)|(1ULL<<11)|(1ULL<<12)|(1ULL<<15)|(1ULL<<16)|(1ULL<<22)|(1ULL<<23)|(1ULL<<26)
                                 | (1ULL<<29)|(1ULL<<30)|(1ULL<<32);
    if (x1 == NA && x2 == NG) return (tag >> tt) & 1;
    if (x1 == NG && x2 == NA) return (tga >> tt) & 1;
    if (x1 == NA && x2 == NA) return (taa >> tt) & 1;
    if (tt == 22) return x1 == NC && x2 == NA;
    if (tt == 23) return x1 == NT && x2 == NA;
    return false;
}
// ref: _sequence.h:45-73
__host__ __device__ __forceinline__ bool codon_is_start(int x0, int x1, int x2, int tt) {
    if (x1 != NT || x2 != NG) return false;
    if (x0 == NA) return true;
    if (tt == 6 || tt == 10 || tt == 14 || tt == 15 || tt == 16 || tt == 2) return false;
    if (x0 == NG) return !(tt == 1 || tt == 3 || tt == 12 || tt == 2);
    if (x0 == NT) return !(tt < 4 || tt == 9 || (tt >= 21 && tt < 25));
    return false;
}
__device__ __forceinline__ bool is_stop_at(const uint8_t* __restrict__ d, int L, int i, int strand, int tt) {
    return codon_is_stop(sbase(d, L, i, strand), sbase(d, L, i + 1, strand), sbase(d, L, i + 2, strand), tt);
}

// ---------------------------------------------------------------------------------- digitise
// ref: lib.pyx:664-697 (Sequence._build)
__device__ __forceinline__ int digit_of(int ch, int& gc, int& unk) {
    switch (ch) {
        case 'A': case 'a': return NA;
        case 'T': case 't': return NT;
        case 'G': case 'g': gc++; return NG;
        case 'C': case 'c': gc++; return NC;
        default: unk++; return NN;
    }
}

// Four letters of a 32-bit word -> four digits (A 0, G 1, C 2, T 3, anything else 6; either case), with the number of G / C and of
// unknown letters among them.  Byte-parallel: no per-letter branch or select chain.
__device__ __forceinline__ unsigned digits4(const unsigned w, int& gc, int& unk) {
    const unsigned u = w & 0xdfdfdfdfu;                           // upper case: only 'a' .. 'z' map onto 'A' .. 'Z'
    auto eq = [](const unsigned v) {                              // 0x01 in every byte of v that is zero (exact, no borrow between bytes)
        return (~(((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v | 0x7f7f7f7fu)) >> 7;
    };
    const unsigned a = eq(u ^ 0x41414141u), g = eq(u ^ 0x47474747u), c = eq(u ^ 0x43434343u), t = eq(u ^ 0x54545454u);
    const unsigned n = ~(a | g | c | t) & 0x01010101u;
    gc += __popc(g | c); unk += __popc(n);
    return g | (c << 1) | t | (t << 1) | (n << 1) | (n << 2);
}

// 64 bases per thread as four 16-byte pieces 4 KB apart (the pieces of a wavefront are contiguous: coalesced, four loads in flight
// per lane); GC / unknown counts are reduced per wavefront and piece and added with one atomic when the piece's 1 KB lies in
// one contig.
constexpr int DG_BLOCK = 16384;
__global__ void __launch_bounds__(256)
k_digitize(const char* __restrict__ seq, uint8_t* __restrict__ dig, int64_t total,
           const ContigDesc* __restrict__ ct, int n_contigs, int32_t* __restrict__ gc_count, int32_t* __restrict__ unk_count) {
    __shared__ int s_c0;
    const int64_t blk0 = (int64_t)blockIdx.x * DG_BLOCK;
    int c = block_contig(ct, n_contigs, blk0, blk0, &s_c0);       // contig of the block's first base; pieces walk on from there
    uint4 v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int64_t g0 = blk0 + q * 4096 + (int64_t)threadIdx.x * 16;
        v[q] = g0 + 16 <= total ? *reinterpret_cast<const uint4*>(seq + g0) : make_uint4(0, 0, 0, 0);
    }
    int acc_c = -1, acc_gc = 0, acc_unk = 0;            // wave-uniform: what the wavefront holds back for contig acc_c
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int64_t g0 = blk0 + q * 4096 + (int64_t)threadIdx.x * 16;
        const bool in = g0 < total;
        if (in) while (c + 1 < n_contigs && ct[c + 1].base <= g0) c++;
        int gc = 0, unk = 0;
        bool one_contig = true;
        if (in) {
            const int64_t next = ct[c + 1 <= n_contigs ? c + 1 : n_contigs].base;     // ct has n_contigs + 1 entries
            if (g0 + 16 <= total && g0 + 16 <= next) {
                const uint4 o = make_uint4(digits4(v[q].x, gc, unk), digits4(v[q].y, gc, unk), digits4(v[q].z, gc, unk), digits4(v[q].w, gc, unk));
                *reinterpret_cast<uint4*>(dig + g0) = o;
            } else {
                one_contig = false;
                int cc = c;
                for (int k = 0; k < 16 && g0 + k < total; k++) {
                    const int64_t g = g0 + k;
                    while (cc + 1 < n_contigs && ct[cc + 1].base <= g) cc++;
                    int a = 0, u = 0;
                    dig[g] = (uint8_t)digit_of(seq[g], a, u);
                    if (a) atomicAdd(&gc_count[cc], 1);
                    if (u) atomicAdd(&unk_count[cc], 1);
                }
            }
        }
        const int c0 = __builtin_amdgcn_readfirstlane(c);
        const bool uniform = __all(!in || (one_contig && c == c0));
        if (uniform) {
            int a = in ? gc : 0, u = in ? unk : 0;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m, 64); u += __shfl_xor(u, m, 64); }
            // a wavefront's pieces of one contig are added up before they leave it (one contig of 200 Mbp is 195 000 pieces: that many
            // atomics on one address took longer than the digitising)
            if (acc_c != c0) {
                if ((threadIdx.x & 63) == 0 && acc_c >= 0) { if (acc_gc) atomicAdd(&gc_count[acc_c], acc_gc); if (acc_unk) atomicAdd(&unk_count[acc_c], acc_unk); }
                acc_c = c0; acc_gc = 0; acc_unk = 0;
            }
            acc_gc += a; acc_unk += u;
        } else if (in && one_contig) {
            if (gc) atomicAdd(&gc_count[c], gc);
            if (unk) atomicAdd(&unk_count[c], unk);
        }
    }
    // the wavefronts of a block that ended on the same contig leave together
    __shared__ int s_c[4], s_g[4], s_u[4];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_c[wv] = acc_c; s_g[wv] = acc_gc; s_u[wv] = acc_unk; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (s_c[w] < 0) continue;
            int g = s_g[w], u = s_u[w];
#pragma unroll
            for (int v = w + 1; v < 4; v++) if (s_c[v] == s_c[w]) { g += s_g[v]; u += s_u[v]; s_c[v] = -1; }
            if (g) atomicAdd(&gc_count[s_c[w]], g);
            if (u) atomicAdd(&unk_count[s_c[w]], u);
        }
    }
}

// ------------------------------------------------------------------------- node extraction
// ref: lib.pyx:1905-2117 (Nodes._extract).  The reference sweeps each strand right to left with three
// per-frame automata whose whole state is "position of the next in-frame stop to the right" (`last`),
// plus "was a start seen in this ORF".  Both are scans:
//   NS(i) = next in-frame stop to the right of i (or the virtual right end of an open contig),
//   PS(i) = previous in-frame stop to the left of i,
// so every position decides on its own whether it is a start node, and every start also writes the
// stop node of its ORF (all starts of an ORF write identical values).  No walk, no divergence:
// tiles of 3072 positions, 12 consecutive positions per thread, coalesced byte reads.
constexpr int EX_TILE = 3072;          // positions per tile (256 threads x 12)
constexpr int EX_PER_THREAD = 12;
constexpr int EX_NONE_HI = 0x7fffffff; // "no stop to the right"

// 2-bit codes of four digit bytes, byte t at bits 2t (comp: the reverse strand reads the complement; an unknown base stays 2)
__device__ __forceinline__ unsigned pack4(const unsigned w, const bool comp) {
    unsigned c = w & 0x03030303u;
    if (comp) { const unsigned nm = (w >> 2) & 0x01010101u; c = (c ^ 0x03030303u) ^ (nm | (nm << 1)); }
    return (c | (c >> 6) | (c >> 12) | (c >> 18)) & 0xffu;
}
__device__ __forceinline__ unsigned long long pack16(const uint4 v, const bool comp) {      // byte t at bits 2t, 32 bits
    return (unsigned long long)(pack4(v.x, comp) | (pack4(v.y, comp) << 8) | (pack4(v.z, comp) << 16) | (pack4(v.w, comp) << 24));
}
__device__ __forceinline__ unsigned long long pairrev64(unsigned long long x) {              // pair t -> pair 31 - t
    x = __brevll(x);
    return ((x & 0x5555555555555555ull) << 1) | ((x >> 1) & 0x5555555555555555ull);
}

// Geometry: one block per tile of EX_TILE forward positions [a, a + len) of one contig, BOTH strands.  The reverse strand is
// handled in its own (strand-local) coordinates i = L - 1 - pos, i.e. over the mirrored range [L - a - len, L - a): thread t owns
// 12 consecutive strand-local positions of each strand.  Tile summaries (first / last stop of a frame) are kept per strand in
// strand-local positions; "the tiles to the right" of a reverse range are the tiles with SMALLER index.
struct TileGeom {
    int L, a, len;
    __device__ __forceinline__ int lo(const int strand) const { return strand == 1 ? a : L - a - len; }     // first strand-local position
};
__device__ __forceinline__ TileGeom tile_geom(const TileDesc td, const int L) { return TileGeom{L, td.start, min(EX_TILE, L - td.start)}; }

// pass 1: first / last stop position of every tile, per strand and frame (k_tile_stops, below the helpers of pass 2)
// pass 2: every position decides, the tile packs its nodes.
// A start node is decided at its own position from NS / PS (next / previous in-frame stop).  The stop node of an ORF exists
// when the ORF holds at least one start node; the thread that owns the stop position decides that from LSC(x) = the last
// in-frame start codon at or before x: a start at i counts when stop - i + 3 >= min_gene, i.e. i <= x = stop + 3 - min_gene,
// and i lies after the previous stop and outside the ORF's mask, so the ORF has one iff LSC(x) >= that lower bound (plus the
// edge start of an open contig, lib.pyx:1975-1982, 2070-2077).  LSC comes from a third scan over the tile; where it has to
// look left of the tile (the stop sits in the tile's first min_gene positions) the thread walks back codon by codon -- a
// start codon turns up every twenty codons or so.
struct ExShared {
    int lsc[3][256];                   // last start codon of relative frame r at or before the thread's positions (strand-local order)
    int scm[256];                      // the thread's 12 positions: bit q = start codon at i0 + q
    int32_t sv[2][EX_PER_THREAD][256]; // stop_val of the node at the thread's k-th forward position, per strand
    int wmin[3][4], wmax[3][4], wlsc[3][4];
    int carry_ns[3], carry_ps[3];
    int wsum[4];
};

// stop_codons / start_codons: bit (b0 | b1 << 2 | b2 << 4) set when the codon of 2-bit digits b0 b1 b2 is one (codon_is_stop /
// codon_is_start tabulated by the host for the translation table; a codon with an unknown base is neither)
struct ExParams { int tt, closed, min_gene, min_edge_gene; unsigned long long stop_codons, start_codons; };

__device__ __forceinline__ unsigned pairrev32(unsigned x) {      // pair t -> pair 15 - t
    x = __brev(x);
    return ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u);
}
__device__ __forceinline__ unsigned unknown4(const unsigned w) {  // bit t: byte t is an unknown base (digit 6, or 5 after the complement)
    const unsigned n = (w >> 2) & 0x01010101u;
    return (n | (n >> 7) | (n >> 14) | (n >> 21)) & 0xfu;
}

// Thread-level view of one strand: the twelve strand-local positions i0 .. i0 + 11.  inm: the positions that lie in the tile and
// can start a codon; cw: 2-bit digits of i0 .. i0 + 15 (pair q); stm / scm: stop / start codon at i0 + q (within inm).
template <int STRAND>
__device__ __forceinline__ void strand_codon_masks(const TileGeom G, const uint8_t* __restrict__ d, const int64_t base, const int64_t total, const int i0,
                                                   const unsigned long long stop_codons, const unsigned long long start_codons,
                                                   unsigned& inm, unsigned& cw, unsigned& stm, unsigned& scm) {
    constexpr bool FWD = STRAND == 1;
    const int L = G.L, lo = G.lo(STRAND), end = lo + G.len;
    // the thread's positions that exist and can start a codon inside the tile: q in [qa, qb]
    const int qa = max(0, lo - i0), qb = min(EX_PER_THREAD - 1, min(end - 1, L - 3) - i0);
    inm = qb >= qa ? ((2u << qb) - 1u) & ~((1u << qa) - 1u) : 0u;
    // 2-bit digits of positions i0 .. i0 + 15 (pair q), unknown or missing bases in `unk` (bit q)
    cw = 0; unsigned unk = 0xffffu;
    if (inm) {
        const int64_t ga0 = FWD ? base + i0 : base + (L - 1 - i0 - 15);     // lowest byte read
        if (ga0 >= 0 && ga0 <= total) {
            uint4 v; __builtin_memcpy(&v, (FWD ? d + i0 : d + (L - 1 - i0 - 15)), 16);
            const unsigned q0 = (unsigned)pack16(v, !FWD);
            const unsigned n0 = unknown4(v.x) | (unknown4(v.y) << 4) | (unknown4(v.z) << 8) | (unknown4(v.w) << 12);
            cw = FWD ? q0 : pairrev32(q0);
            unk = FWD ? n0 : (__brev(n0) >> 16);
        } else {
            unk = 0;
            for (int q = 0; q < 16; q++) {
                const int i = i0 + q;
                const int b = (i >= 0 && i < L) ? sbase(d, L, i, STRAND) : NN;
                if (b > 3) unk |= 1u << q; else cw |= (unsigned)b << (2 * q);
            }
        }
        // positions outside the contig
        if (i0 < 0) unk |= (1u << min(16, -i0)) - 1u;
        if (i0 + 16 > L) unk |= ~((1u << max(0, L - i0)) - 1u) & 0xffffu;
    }
    stm = 0; scm = 0;
#pragma unroll
    for (int q = 0; q < EX_PER_THREAD; q++) {
        const unsigned idx = (cw >> (2 * q)) & 63u;
        stm |= ((unsigned)(stop_codons >> idx) & 1u) << q;
        scm |= ((unsigned)(start_codons >> idx) & 1u) << q;
    }
    const unsigned okm = inm & ~(unk | (unk >> 1) | (unk >> 2));      // positions whose three bases are all known
    stm &= okm; scm &= okm;
}

// The nodes of one strand among the thread's twelve forward positions a + 12 t + k: bit k of `nodes`, type | edge << 2 in the
// k-th nibble of `infos`, stop_val in S.sv[strand][k][t].  Thread t owns the SAME forward positions on both strands; on the
// reverse strand they are the strand-local positions i0 .. i0 + 11 with i0 = (L - 1 - a) - 12 t - 11, so strand-local order runs
// against t.  Frames are counted relative to i0 (r = q % 3; i0 % 3 is the same for every thread of the tile).
template <int STRAND>
__device__ __forceinline__ void extract_strand(ExShared& S, const TileGeom G, const uint8_t* __restrict__ d, const int64_t base, const int64_t total,
                                               const int tile, const TileDesc* __restrict__ tiles, const int n_tiles,
                                               const int32_t* __restrict__ tile_first, const int32_t* __restrict__ tile_last, const ExParams P,
                                               const int n_masks, const int2* __restrict__ mv, unsigned& nodes, unsigned long long& infos) {
    constexpr bool FWD = STRAND == 1;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, L = G.L, s = FWD ? 0 : 1;
    const int lo = G.lo(STRAND), end = lo + G.len;
    const int top = FWD ? G.a : L - 1 - G.a - (EX_PER_THREAD - 1);        // i0 of thread 0
    const int i0 = FWD ? top + t * EX_PER_THREAD : top - t * EX_PER_THREAD;
    const int f0 = ((top % 3) + 3) % 3;                                    // absolute frame of relative frame 0
    nodes = 0; infos = 0;
    // tile carries: nearest stop of each frame in the tiles that follow / precede in strand-local order (same contig)
    if (t < 6) {
        const int f = t % 3;
        const int64_t so = (int64_t)s * n_tiles;
        const int contig = tiles[tile].contig;
        const int up = FWD ? 1 : -1;                    // tile index step towards larger strand-local positions
        if (t < 3) {
            int v = EX_NONE_HI;
            for (int k = tile + up; k >= 0 && k < n_tiles && tiles[k].contig == contig; k += up) {
                v = tile_first[(so + k) * 3 + f];
                if (v != EX_NONE_HI) break;
            }
            S.carry_ns[f] = v;
        } else {
            int v = -1;
            for (int k = tile - up; k >= 0 && k < n_tiles && tiles[k].contig == contig; k -= up) {
                v = tile_last[(so + k) * 3 + f];
                if (v != -1) break;
            }
            S.carry_ps[f] = v;
        }
    }
    unsigned inm, cw, stm, scm;
    strand_codon_masks<STRAND>(G, d, base, total, i0, P.stop_codons, P.start_codons, inm, cw, stm, scm);
    int mn[3], mx[3], ls[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const unsigned fr = 0x249u << r;
        const unsigned a = stm & fr, b = scm & fr;
        mn[r] = a ? i0 + __builtin_ctz(a) : EX_NONE_HI;
        mx[r] = a ? i0 + 31 - __builtin_clz(a) : -1;
        ls[r] = b ? i0 + 31 - __builtin_clz(b) : -1;
    }
    // Next / previous stop and last start codon of every frame over the 256 threads in strand-local order (FWD: rising t).  Positions
    // rise with the strand-local order, so "the next stop after my positions" is the first stop of the NEAREST later thread that has one:
    // a vote says which lanes have a stop of the frame, a bit scan over the lanes on the proper side picks the neighbour, one
    // shuffle fetches its value (six rounds of three shuffles per frame before: a third of the kernel's vector instructions);
    // across the four wavefronts through LDS as before.
    const unsigned long long lanes_up = FWD ? (~1ull << lane) : ((1ull << lane) - 1ull);      // the lanes later in strand-local order
    const unsigned long long lanes_dn = FWD ? ((1ull << lane) - 1ull) : (~1ull << lane);      // ... and earlier
    int xa[3], xb[3], xe[3];                // exclusive next stop, exclusive previous stop, inclusive last start codon -- within the wavefront
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const unsigned long long hs = __ballot(mx[r] >= 0), hc = __ballot(ls[r] >= 0);
        const unsigned long long mu = hs & lanes_up, md = hs & lanes_dn, mc = hc & lanes_dn;
        // nearest lane on that side (any lane when there is none: the value is not used)
        const int su = mu ? (FWD ? __builtin_ctzll(mu) : 63 - __builtin_clzll(mu)) : lane;
        const int sd = md ? (FWD ? 63 - __builtin_clzll(md) : __builtin_ctzll(md)) : lane;
        const int sc = mc ? (FWD ? 63 - __builtin_clzll(mc) : __builtin_ctzll(mc)) : lane;
        const int va = __shfl(mn[r], su, 64), vb = __shfl(mx[r], sd, 64), ve = __shfl(ls[r], sc, 64);
        xa[r] = mu ? va : EX_NONE_HI;
        xb[r] = md ? vb : -1;
        xe[r] = ls[r] >= 0 ? ls[r] : (mc ? ve : -1);
        // the wavefront's own first / last stop and last start codon, for the other wavefronts
        const int l_first = hs ? (FWD ? __builtin_ctzll(hs) : 63 - __builtin_clzll(hs)) : 0;
        const int l_last = hs ? (FWD ? 63 - __builtin_clzll(hs) : __builtin_ctzll(hs)) : 0;
        const int l_lastc = hc ? (FWD ? 63 - __builtin_clzll(hc) : __builtin_ctzll(hc)) : 0;
        const int w_mn = __shfl(mn[r], l_first, 64), w_mx = __shfl(mx[r], l_last, 64), w_ls = __shfl(ls[r], l_lastc, 64);
        if (lane == 0) { S.wmin[r][w] = hs ? w_mn : EX_NONE_HI; S.wmax[r][w] = hs ? w_mx : -1; S.wlsc[r][w] = hc ? w_ls : -1; }
    }
    S.scm[t] = (int)scm;
    __syncthreads();
    int nxt[3], prv[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        // then the wavefronts beyond this one, then the tile carry
        int a = xa[r], b = xb[r], e = xe[r];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool up_side = FWD ? k > w : k < w, dn_side = FWD ? k < w : k > w;
            if (up_side) a = min(a, S.wmin[r][k]);
            if (dn_side) { b = max(b, S.wmax[r][k]); e = max(e, S.wlsc[r][k]); }
        }
        const int f = (f0 + r) % 3;
        nxt[r] = a == EX_NONE_HI ? S.carry_ns[f] : a;
        prv[r] = b == -1 ? S.carry_ps[f] : b;
        S.lsc[r][t] = e;
    }
    __syncthreads();
    // region masks (ref: lib.pyx:1959-1966, 2053-2061).  The reference keeps one mask pointer per frame that follows the ORF's
    // stop and tests that single mask: forward, the last mask beginning at or before the stop; reverse, the first mask ending
    // at or after the stop's forward coordinate.  Either way a start at i of the ORF ending at `last` is masked iff i < bound.
    auto mask_bound = [&](const int last) -> int {
        if (n_masks <= 0) return INT_MIN;
        if (FWD) {
            int a = 0, b = n_masks;                 // first mask with begin > last
            while (a < b) { const int m = (a + b) >> 1; if (mv[m].x <= last) a = m + 1; else b = m; }
            if (a > 0) { const int2 m = mv[a - 1]; if (m.x < last) return m.y; }
        } else {
            const int x = L - last - 1;
            int a = 0, b = n_masks;                 // first mask with end >= x
            while (a < b) { const int m = (a + b) >> 1; if (mv[m].y < x) a = m + 1; else b = m; }
            if (a < n_masks) { const int2 m = mv[a]; if (x < m.y) return L - 1 - m.x; }
        }
        return INT_MIN;
    };
    // the last start codon of relative frame r (absolute f) at or before x, not looking below `lower`; -1: none
    auto last_start_codon = [&](const int r, const int f, const int x, const int lower) -> int {
        int y;                                      // where the codon-by-codon walk to the left starts
        if (x >= lo) {
            const int tx = FWD ? (x - top) / EX_PER_THREAD : (top + EX_PER_THREAD - 1 - x) / EX_PER_THREAD;
            const int i0x = FWD ? top + tx * EX_PER_THREAD : top - tx * EX_PER_THREAD;
            const unsigned m = (unsigned)S.scm[tx] & ((2u << (x - i0x)) - 1u) & (0x249u << r);
            if (m) return i0x + 31 - __builtin_clz(m);
            const int nb = FWD ? tx - 1 : tx + 1;   // the thread before in strand-local order
            const int v = (nb >= 0 && nb < 256) ? S.lsc[r][nb] : -1;
            if (v >= 0) return v;
            y = lo - 1 - ((lo - 1 - f) % 3 + 3) % 3;
        } else y = x - ((x - f) % 3 + 3) % 3;
        for (int i = y; i >= lower && i >= 0; i -= 3)
            if (codon_is_start(sbase(d, L, i, STRAND), sbase(d, L, i + 1, STRAND), sbase(d, L, i + 2, STRAND), P.tt)) return i;
        return -1;
    };
    // does the ORF of frame f that ends at `last` (previous stop `left`) hold a start node?
    auto orf_has_start = [&](const int r, const int f, const int last, const int left, const bool real) -> bool {
        const int mb = mask_bound(last);
        const int lower = max(left + 1, mb);
        const int mind = real ? P.min_gene : P.min_edge_gene;
        const int x = min(last + 3 - mind, last - 3);
        if (x >= lower && last_start_codon(r, f, x, lower) >= lower) return true;
        // the edge start of an open contig (i = f <= 2)
        return !P.closed && left < 0 && f < last && last - f > P.min_edge_gene && f >= mb;
    };
    auto put = [&](const int q, const int info, const int sv) {
        const int k = FWD ? q : EX_PER_THREAD - 1 - q;
        nodes |= 1u << k;
        infos |= (unsigned long long)info << (4 * k);
        S.sv[s][k][t] = sv;
    };
    // candidates: stop codons, start codons, the first position of each frame (edge starts) and the last (virtual stops)
    unsigned cand = stm | scm;
    if (!P.closed) {
        if (i0 <= 2 && i0 + EX_PER_THREAD > 0) cand |= (i0 >= 0 ? 7u >> i0 : 7u << -i0) & inm;        // i <= 2
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int f = (f0 + r) % 3;
            const int q = (L - 3 - ((L - 3 - f) % 3)) - i0;
            if (q >= 0 && q < EX_PER_THREAD) cand |= (1u << q) & inm;
        }
    }
    // One candidate.  Stop codons and the rest go through loops of their own (below): lanes of a wavefront that sit in the same
    // iteration then run the same path -- in one loop over all candidates an iteration cost both paths whenever the lanes' kinds differed.
    auto candidate = [&](const int q, const bool is_stop) {
        const int i = i0 + q;
        const int r = q % 3, f = (f0 + r) % 3;
        const unsigned fr = 0x249u << r;
        // the stop to the right: among the thread's own positions, else from the scan
        const unsigned above = stm & fr & ~((2u << q) - 1u);
        const unsigned below = stm & fr & ((1u << q) - 1u);
        const int left = below ? i0 + 31 - __builtin_clz(below) : (r == 0 ? prv[0] : (r == 1 ? prv[1] : prv[2]));
        const int sv_stop = FWD ? (left >= 0 ? left : f - 6) : (left >= 0 ? L - 1 - left : L - f + 5);
        if (is_stop) {
            if (orf_has_start(r, f, i, left, true)) put(q, PGA_T_STOP, sv_stop);
            return;
        }
        int last = above ? i0 + __builtin_ctz(above) : (r == 0 ? nxt[0] : (r == 1 ? nxt[1] : nxt[2]));
        const bool real = last != EX_NONE_HI;
        if (!real) {
            if (P.closed) return;                     // closed ends: nothing runs off the right edge
            last = L - 3 - ((L - 3 - f) % 3);           // virtual right end: last full codon position of the frame
            if (last < i) return;
            if (last == i) {                            // the virtual stop node itself
                if (orf_has_start(r, f, i, left, false)) put(q, PGA_T_STOP | (1 << 2), sv_stop);
                return;
            }
        }
        if (n_masks > 0 && i < mask_bound(last)) return;
        const int mind = real ? P.min_gene : P.min_edge_gene;
        int type = -1, edge = 0;
        const int c0 = (int)((cw >> (2 * q)) & 3u);
        if (last - i + 3 >= mind && ((scm >> q) & 1u)) type = c0 == NA ? 0 : (c0 == NT ? 2 : 1);
        else if (i <= 2 && !P.closed && last - i > P.min_edge_gene) { type = 0; edge = 1; }
        if (type < 0) return;
        put(q, type | (edge << 2), FWD ? last : L - 1 - last);
    };
    unsigned stops = stm;
    while (stops) { const int q = __builtin_ctz(stops); stops &= stops - 1u; candidate(q, true); }
    cand &= ~stm;
    while (cand) { const int q = __builtin_ctz(cand); cand &= cand - 1u; candidate(q, false); }
}

// One wavefront per tile (end of round 6; it was a thread per twelve positions, both strands: every position of the batch decoded once
// more).  A tile's FIRST stop of a frame lies, nearly always, within its first 128 codons and its LAST within its last 128: lanes 0 - 31
// decode the tile's first 384 strand-local positions of a strand, lanes 32 - 63 its last 384, a vote and a bit scan per frame and half give
// the six numbers; only where a frame has no stop there (long ORFs of GC-rich sequence) do the lanes go on to the next 384 positions.
__global__ void __launch_bounds__(64)
k_tile_stops(const uint8_t* __restrict__ dig, int64_t total, const ContigDesc* __restrict__ ct, const TileDesc* __restrict__ tiles, int n_tiles,
             unsigned long long stop_codons, int32_t* __restrict__ tile_first, int32_t* __restrict__ tile_last, const uint8_t* __restrict__ enabled) {
    const TileDesc td = tiles[blockIdx.x];
    if (enabled != nullptr && !enabled[td.contig]) return;       // no model of this translation table is scored on the contig
    const ContigDesc cd = ct[td.contig];
    const TileGeom G = tile_geom(td, cd.len);
    const int lane = threadIdx.x & 63, hl = lane & 31;
    const bool first = lane < 32;                                 // lanes 0 - 31: the first stops; lanes 32 - 63: the last stops
    const uint8_t* __restrict__ d = dig + cd.base;
    constexpr int CHUNK = 32 * EX_PER_THREAD;
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int lo = G.lo(s == 0 ? 1 : -1), end = lo + G.len;  // the tile's strand-local positions
        int res[3];                                               // by relative frame (position - i0) % 3; every i0 of a half is the same mod 3
        res[0] = res[1] = res[2] = first ? EX_NONE_HI : -1;
        unsigned found = 0;                                       // bits 0 - 2: the first stops, bits 3 - 5: the last stops (wave-uniform)
        int fbase = 0;
        for (int c = 0; c * CHUNK < G.len && found != 63u; c++) {
            // a half's lanes rise with the position (first) or fall (last): the half's answer is that of its lowest lane that has one
            const int i0 = first ? lo + c * CHUNK + hl * EX_PER_THREAD : end - c * CHUNK - (hl + 1) * EX_PER_THREAD;
            fbase = ((i0 % 3) + 3) % 3;
            unsigned inm, cw, stm, scm;
            if (s == 0) strand_codon_masks<1>(G, d, cd.base, total, i0, stop_codons, 0ull, inm, cw, stm, scm);
            else strand_codon_masks<-1>(G, d, cd.base, total, i0, stop_codons, 0ull, inm, cw, stm, scm);
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const unsigned m = stm & (0x249u << r);
                const unsigned long long hs = __ballot(m != 0u);
                const unsigned h_first = (unsigned)hs, h_last = (unsigned)(hs >> 32);
                const int v = m ? (first ? i0 + __builtin_ctz(m) : i0 + 31 - __builtin_clz(m)) : 0;
                const int src = first ? (h_first ? __builtin_ctz(h_first) : 0) : (h_last ? 32 + __builtin_ctz(h_last) : 32);
                const int wv = __shfl(v, src, 64);
                const bool mine_has = first ? h_first != 0u : h_last != 0u;
                const unsigned bit = first ? (1u << r) : (8u << r);
                if (mine_has && !(found & bit)) res[r] = wv;
                found |= (h_first ? (1u << r) : 0u) | (h_last ? (8u << r) : 0u);
            }
        }
        if (hl < 3) {
            const int f = (fbase + hl) % 3;                       // relative frame `hl` of the half is frame f of the strand
            const int64_t o = ((int64_t)s * n_tiles + blockIdx.x) * 3 + f;
            const int v = hl == 0 ? res[0] : (hl == 1 ? res[1] : res[2]);
            if (first) tile_first[o] = v; else tile_last[o] = v;
        }
    }
}

__global__ void __launch_bounds__(256)
k_extract_tile(const uint8_t* __restrict__ dig, int64_t total, const ContigDesc* __restrict__ ct, const TileDesc* __restrict__ tiles, int n_tiles,
               const int32_t* __restrict__ tile_first, const int32_t* __restrict__ tile_last, ExParams P, GroupArrays ga, MaskList masks,
               const uint8_t* __restrict__ enabled, int32_t* __restrict__ tile_count, int32_t* __restrict__ tile_scount) {
    __shared__ ExShared S;
    __shared__ int s_stops;
    const int tile = blockIdx.x, t = threadIdx.x;
    const TileDesc td = tiles[tile];
    if (enabled != nullptr && !enabled[td.contig]) { if (t == 0) { tile_count[tile] = 0; tile_scount[tile] = 0; } return; }     // the contig keeps zero nodes in this group
    if (t == 0) s_stops = 0;
    const ContigDesc cd = ct[td.contig];
    const TileGeom G = tile_geom(td, cd.len);
    const uint8_t* __restrict__ d = dig + cd.base;
    const int n_masks = masks.off ? masks.off[td.contig + 1] - masks.off[td.contig] : 0;
    const int2* __restrict__ mv = masks.off ? masks.iv + masks.off[td.contig] : nullptr;
    unsigned nf, nr; unsigned long long inf_f, inf_r;
    extract_strand<1>(S, G, d, cd.base, total, tile, tiles, n_tiles, tile_first, tile_last, P, n_masks, mv, nf, inf_f);
    __syncthreads();
    extract_strand<-1>(S, G, d, cd.base, total, tile, tiles, n_tiles, tile_first, tile_last, P, n_masks, mv, nr, inf_r);
    // pack: nodes in (position, strand) order -- forward sorts before reverse at equal ndx (Prodigal compare_nodes;
    // ref: lib.pyx:2489-2493) -- into the tile's staging slots; the per-position flags of the ORF walks
    const int r0 = t * EX_PER_THREAD;
    const int cnt = __popc(nf) + __popc(nr);
    int inc = cnt;
    const int lane = t & 63, w = t >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off, 64); if (lane >= off) inc += v; }
    if (lane == 63) S.wsum[w] = inc;
    __syncthreads();
    int nbase = inc - cnt;
    for (int k = 0; k < w; k++) nbase += S.wsum[k];
    if (t == 255) tile_count[tile] = nbase + cnt;
    const int64_t g0 = cd.base + G.a;
    if (r0 < G.len) {
        // what the ORF walks read of the thread's twelve positions: digit | forward-node flag << 4 | reverse-node flag << 5
        const int nk = min(EX_PER_THREAD, G.len - r0);
        auto spread4 = [](const unsigned x) { return ((x & 15u) * 0x00204081u) & 0x01010101u; };       // bit k -> bit 0 of byte k
        if (nk == EX_PER_THREAD) {
            unsigned w[3];
            __builtin_memcpy(w, dig + g0 + r0, 12);
#pragma unroll
            for (int q = 0; q < 3; q++) w[q] = (w[q] & 0x07070707u) | (spread4(nf >> (4 * q)) << 4) | (spread4(nr >> (4 * q)) << 5);
            __builtin_memcpy(ga.df + g0 + r0, w, 12);
        } else {
            for (int k = 0; k < nk; k++) ga.df[g0 + r0 + k] = (uint8_t)((dig[g0 + r0 + k] & 7u) | (((nf >> k) & 1u) << 4) | (((nr >> k) & 1u) << 5));
        }
    }
    // (with st_half every tile gets PGA_STAGE_SLACK slots on top of its share: the last tile of a contig can be a few bases long and
    //  still hold the six edge nodes of an open end)
    const int64_t stage0 = ga.st_half ? (g0 >> ga.st_half) + (int64_t)PGA_STAGE_SLACK * tile : 2 * g0;
    const int room = ga.st_half ? (G.len >> ga.st_half) + PGA_STAGE_SLACK : 2 * G.len;   // staging slots of the tile
    if (t == 255 && nbase + cnt > room) atomicOr(ga.st_overflow, 1);   // (only with st_half) the caller extracts again, full staging
    int64_t slot = stage0 + nbase;
    const int64_t slot_end = stage0 + room;
    unsigned both = nf | nr;
    int n_stop = 0;
    while (both) {
        const int k = __builtin_ctz(both);
        both &= both - 1u;
        if ((nf >> k) & 1u) {
            const int info = (int)((inf_f >> (4 * k)) & 7);
            if (slot < slot_end) { ga.st_ndx[slot] = G.a + r0 + k; ga.st_sv[slot] = S.sv[0][k][t]; ga.st_info[slot] = (uint8_t)info; }
            slot++; n_stop += (info & 3) == PGA_T_STOP;
        }
        if ((nr >> k) & 1u) {
            const int info = (int)((inf_r >> (4 * k)) & 7);
            if (slot < slot_end) { ga.st_ndx[slot] = G.a + r0 + k; ga.st_sv[slot] = S.sv[1][k][t]; ga.st_info[slot] = (uint8_t)(info | 8); }
            slot++; n_stop += (info & 3) == PGA_T_STOP;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) n_stop += __shfl_xor(n_stop, m, 64);
    if (lane == 0 && n_stop) atomicAdd(&s_stops, n_stop);
    __syncthreads();
    if (t == 0) tile_scount[tile] = s_stops;
}

// Exclusive scan of n counts into n + 1 offsets, one workgroup, 4096 counts per round.  One compute unit moves every byte, and what
// limits it is memory instructions that touch one 32-byte sector per lane: counts are therefore read and offsets written with
// lane-consecutive addresses and change hands through LDS (a thread scans four neighbours).  Two arrays at once when cnt2 is given
// (node and stop-node counts of the tiles share the launch and its barriers).
constexpr int SCAN_E = 4;
__device__ __forceinline__ int scan_pad(const int i) { return i + (i >> 5); }
__global__ void __launch_bounds__(1024)
k_scan_counts(const int32_t* __restrict__ cnt, int n, int32_t* __restrict__ off, const int32_t* __restrict__ cnt2, int32_t* __restrict__ off2) {
    __shared__ int s_a[SCAN_E * 1024 + SCAN_E * 32], s_b[SCAN_E * 1024 + SCAN_E * 32];
    __shared__ int s_w[16], s_w2[16];
    __shared__ int s_carry, s_carry2;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const bool two = cnt2 != nullptr;
    if (t == 0) { s_carry = 0; s_carry2 = 0; }
    __syncthreads();
    for (int base = 0; base < n; base += SCAN_E * 1024) {
        int c[SCAN_E], c2[SCAN_E];
#pragma unroll
        for (int k = 0; k < SCAN_E; k++) {
            const int g = base + k * 1024 + t;
            c[k] = g < n ? cnt[g] : 0;
            c2[k] = two && g < n ? cnt2[g] : 0;
        }
#pragma unroll
        for (int k = 0; k < SCAN_E; k++) { s_a[scan_pad(k * 1024 + t)] = c[k]; s_b[scan_pad(k * 1024 + t)] = c2[k]; }
        __syncthreads();
        int v[SCAN_E], v2[SCAN_E], sum = 0, sum2 = 0;
#pragma unroll
        for (int k = 0; k < SCAN_E; k++) { v[k] = s_a[scan_pad(t * SCAN_E + k)]; v2[k] = s_b[scan_pad(t * SCAN_E + k)]; sum += v[k]; sum2 += v2[k]; }
        int inc = sum, inc2 = sum2;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int x = __shfl_up(inc, o, 64), x2 = __shfl_up(inc2, o, 64);
            if (lane >= o) { inc += x; inc2 += x2; }
        }
        if (lane == 63) { s_w[w] = inc; s_w2[w] = inc2; }
        __syncthreads();
        int run = s_carry + inc - sum, run2 = s_carry2 + inc2 - sum2;
        for (int k = 0; k < w; k++) { run += s_w[k]; run2 += s_w2[k]; }
#pragma unroll
        for (int k = 0; k < SCAN_E; k++) {
            s_a[scan_pad(t * SCAN_E + k)] = run; s_b[scan_pad(t * SCAN_E + k)] = run2;       // the thread's own four slots
            run += v[k]; run2 += v2[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SCAN_E; k++) {
            const int g = base + k * 1024 + t;
            if (g < n) { off[g] = s_a[scan_pad(k * 1024 + t)]; if (two) off2[g] = s_b[scan_pad(k * 1024 + t)]; }
        }
        if (t == 1023) { s_carry = run; s_carry2 = run2; }
        __syncthreads();
    }
    if (t == 0) { off[n] = s_carry; if (two) off2[n] = s_carry2; }
}

// first node of every contig (n_contigs + 1 entries) from the tile offsets
__global__ void k_contig_node_base(const int32_t* __restrict__ tile0, int n_contigs, const int32_t* __restrict__ tile_off, int32_t* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c <= n_contigs) out[c] = tile_off[tile0[c]];
}

// ------------------------------------------------------------------------- placement
// The staged nodes of every tile go to their final index; the position of a node learns the index of its first node
// (the ORF walks of the coding score turn positions into node indices through it).
__global__ void __launch_bounds__(128)
k_place_nodes(const ContigDesc* __restrict__ ct, const TileDesc* __restrict__ tiles, const int32_t* __restrict__ tile_off,
              const int32_t* __restrict__ tile_soff, GroupArrays ga) {
    __shared__ int s_w[2];
    const TileDesc td = tiles[blockIdx.x];
    const int off = tile_off[blockIdx.x], cnt = tile_off[blockIdx.x + 1] - off;
    const ContigDesc cd = ct[td.contig];
    // c16 of the tile: entry g = the first node at or after position td.start + 16 g.  Node j answers for the groups behind its
    // predecessor's up to its own, the last node's thread for the rest of the tile as well (their first node is the next tile's).
    static_assert(EX_TILE == 192 * 16, "GroupArrays::c16 holds 192 entries per tile");
    int32_t* const c16 = ga.c16 + (size_t)blockIdx.x * 192;
    const int n_groups = (min(EX_TILE, cd.len - td.start) + 15) >> 4;
    if (cnt <= 0) {
        for (int g = threadIdx.x; g < n_groups; g += blockDim.x) c16[g] = off;
        return;
    }
    const int64_t base = cd.base;
    const int64_t s0 = ga.st_half ? ((base + td.start) >> ga.st_half) + (int64_t)PGA_STAGE_SLACK * blockIdx.x : 2 * (base + td.start);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int srun = tile_soff[blockIdx.x];                       // next free entry of the stop list
    const int start0 = off - srun;                          // start nodes before this tile = nodes before it - stop nodes before it
    for (int j0 = 0; j0 < cnt; j0 += 128) {
        const int j = j0 + threadIdx.x;
        bool is_stop = false;
        int k = 0;
        if (j < cnt) {
            const int ndx = ga.st_ndx[s0 + j], info = ga.st_info[s0 + j];
            k = off + j;
            ga.ndx[k] = ndx; ga.stop_val[k] = ga.st_sv[s0 + j]; ga.type[k] = info & 3; ga.strand[k] = (info >> 3) & 1 ? -1 : 1; ga.edge0[k] = (info >> 2) & 1;
            ga.contig_of[k] = td.contig;
            const int g_hi = (ndx - td.start) >> 4, g_lo = j == 0 ? 0 : ((ga.st_ndx[s0 + j - 1] - td.start) >> 4) + 1;
            for (int g = g_lo; g <= g_hi; g++) c16[g] = k;
            if (j == cnt - 1) for (int g = g_hi + 1; g < n_groups; g++) c16[g] = off + cnt;
            is_stop = (info & 3) == PGA_T_STOP;
        }
        const unsigned long long bal = __ballot(is_stop);
        if (lane == 0) s_w[wv] = __popcll(bal);
        __syncthreads();
        const int stops_before = srun + (wv ? s_w[0] : 0) + __popcll(bal & ((1ull << lane) - 1ull));      // stop-list entries before node j
        if (is_stop) ga.stop_list[stops_before] = k;
        else if (j < cnt && ga.start_list != nullptr) ga.start_list[start0 + j - (stops_before - tile_soff[blockIdx.x])] = k;
        srun += s_w[0] + s_w[1];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------- GC prefix
// Count of GC-or-unknown bases before every 16th position of the batch (k_orf_gc adds the bases of the last, partial group
// itself): per 4096-position block a sum, a scan of the sums, the prefix per group of 16.
__device__ __forceinline__ int gcn16(const uint4 v, const int nbytes /* leading bytes that count, 0 .. 16 */) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    int n = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        // a digit is G, C or unknown (1, 2, 6) iff its two low bits differ (A 0, T 3)
        unsigned b = (w[q] ^ (w[q] >> 1)) & 0x01010101u;
        const int keep = nbytes - 4 * q;
        if (keep <= 0) b = 0; else if (keep < 4) b &= (1u << (8 * keep)) - 1u;
        n += __popc(b);
    }
    return n;
}
__global__ void __launch_bounds__(256)
k_gcp_blocks(const uint8_t* __restrict__ dig, int64_t total, int32_t* __restrict__ block_sum) {
    __shared__ int s_w[4];
    const int64_t g0 = (int64_t)blockIdx.x * 4096 + (int64_t)threadIdx.x * 16;
    int n = 0;
    if (g0 < total) n = gcn16(*reinterpret_cast<const uint4*>(dig + g0), (int)min((int64_t)16, total - g0));
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) n += __shfl_xor(n, m, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) block_sum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ void __launch_bounds__(256)
k_gcp_final(const uint8_t* __restrict__ dig, int64_t total, const int32_t* __restrict__ block_off, int32_t* __restrict__ p16) {
    __shared__ int s_w[4];
    const int64_t g0 = (int64_t)blockIdx.x * 4096 + (int64_t)threadIdx.x * 16;
    int n = 0;
    if (g0 < total) n = gcn16(*reinterpret_cast<const uint4*>(dig + g0), (int)min((int64_t)16, total - g0));
    int inc = n;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off, 64); if (lane >= off) inc += v; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int run = block_off[blockIdx.x] + inc - n;
    for (int k = 0; k < w; k++) run += s_w[k];
    if (g0 <= total) p16[g0 >> 4] = run;
}
// GC-or-unknown bases before global position x
__device__ __forceinline__ int gc_prefix(const uint8_t* __restrict__ dig, const int32_t* __restrict__ p16, const int64_t x) {
    const int r = (int)(x & 15);
    int n = p16[x >> 4];
    if (r) n += gcn16(*reinterpret_cast<const uint4*>(dig + (x - r)), r);
    return n;
}

// ------------------------------------------------------------------------- ORF GC content
// ref: lib.pyx:1846-1896 (Nodes._calc_orf_gc).  The reference accumulates integer counts along
// the ORF; with a prefix count of GC-or-unknown bases every start reads its count in O(1).
// The reverse strand keeps the reference's shifted range (codons counted at j..j+2).
__global__ void __launch_bounds__(256)
k_orf_gc(const ContigDesc* __restrict__ ct, int n_contigs, const uint8_t* __restrict__ dig, const int32_t* __restrict__ p16, GroupArrays ga,
         int n_nodes_total, const int32_t* __restrict__ node_contig_base /* per contig: first node idx */) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes_total) return;
    if (ga.type[i] == PGA_T_STOP) { ga.gc_cont[i] = 0.f; return; }
    const int lo = ga.contig_of[i];
    const int L = ct[lo].len;
    const int64_t base = ct[lo].base;
    auto P = [&](const int x) { return gc_prefix(dig, p16, base + x); };
    const int ndx = ga.ndx[i], sv = ga.stop_val[i];
    int cnt;
    if (ga.strand[i] == 1) {
        cnt = P(sv + 3) - P(ndx);
    } else {
        const int hi2 = min(ndx + 2, L - 1);
        cnt = (P(sv + 1) - P(sv - 2)) + (hi2 >= sv + 3 ? P(hi2 + 1) - P(sv + 3) : 0);
    }
    const double gsize = abs(sv - ndx) + 3.0;
    ga.gc_cont[i] = (float)((double)cnt / gsize);
}

// -------------------------------------------------------------------------- coding score
// One thread per stop node: pass 1 walks the ORF from the stop outwards adding gene_dc[hexamer]
// in the reference's order; passes 2 and 3 revisit the ORF's starts from the outermost one
// inwards.   ref: lib.pyx:2119-2239 (Nodes._raw_coding_score)
__device__ __forceinline__ int hexamer(const uint8_t* __restrict__ d, int pos, int strand) {
    int v = 0;   // ref: _sequence.h:207-220; pos = forward coordinate of the first base read
    if (strand == 1) { for (int j = 0; j < 6; j++) v |= (d[pos + j] & 3) << (2 * j); }
    else             { for (int j = 0; j < 6; j++) v |= (comp2(d[pos - j] & 7) & 3) << (2 * j); }       // & 7: GroupArrays::df carries node flags above the digit
    return v;
}

// Coding score of every start, for every model scored on the contig (ref: lib.pyx:2119-2239).
// The hexamer log-odds sum is an ordered f64 sum from the stop outwards (no reassociation allowed).
// One thread per stop node of the TOPOLOGY walks a short ORF on its own and accumulates the sums of up
// to CS_MODELS models at once (the walk -- rolling hexamer, start flags -- is model-independent).
// ORFs longer than CS_LONG codons would make one lane a long serial chain of dependent loads, so they are
// handed to the whole wavefront instead: 64 codons at a time every lane loads the table values of its own
// codon, the ordered sum then runs through the wave with v_readlane broadcasts (every lane adds the same
// values in the same order and keeps the prefix of its codon), and passes 2 / 3 -- which only need the
// running maximum of the ORIGINAL values of the starts further out, exact in any order -- are wave scans.
// A block first compacts its stop nodes into the leading lanes.
constexpr int CS_MODELS = 4;
constexpr int CS_LONG = 192;
constexpr int CSQ_SERIAL_MAX = 256;                    // k_coding_score_quads: the longest ORF a lane walks on its own (four 64-bit masks of start nodes)

struct OrfCtx {
    const uint8_t* d;             // GroupArrays::df of the contig: digit | forward-node flag << 4 | reverse-node flag << 5, by position
    const int32_t* c16;           // GroupArrays::c16 of the contig: entry g = index (in the group) of the first node at or after position 16 g
    int tbase, p, q, L, strand, step, ncod;
    int kstop;                    // index of the ORF's stop node in its contig
    int2 cc;
    // the walk's own strand has a node at position j / that node's index in the contig
    __device__ __forceinline__ bool node_at(const int j) const { return (d[j] >> (strand == 1 ? 4 : 5)) & 1; }
    // (a count: the first node of the sixteen positions from j & ~15 on, plus the nodes -- either strand -- at the positions before j, which
    //  the bytes of those positions carry; the reverse node of a position follows its forward node)
    __device__ __forceinline__ int node_index(const int j) const {
        const int r = j & 15;
        uint4 v; __builtin_memcpy(&v, d + (j - r), 16);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        int n = c16[j >> 4] - tbase;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int keep = r - 4 * q;                       // bytes of this word before position j
            unsigned m = w[q] & 0x30303030u;
            if (keep <= 0) m = 0; else if (keep < 4) m &= (1u << (8 * keep)) - 1u;
            n += __popc(m);
        }
        const unsigned bj = (w[r >> 2] >> (8 * (r & 3))) & 0xffu;        // the byte of position j itself
        return n + (strand == 1 ? 0 : (int)((bj >> 4) & 1u));
    }
};

__device__ __forceinline__ double wave_incl_max(double v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_up(v, off, 64);
        if (lane >= off) v = fmax(v, o);
    }
    return v;
}

__device__ __forceinline__ double length_factor(const ModelScoreConst* mc, int ncodons) {   // ref: lib.pyx:2203-2210
    const double gsize = (double)ncodons;
    if (gsize > 1000.0) return (mc->lfac_max - mc->lfac_min) * (gsize - 80) / 920.0;
    return mc->lfac_tab[ncodons];
}

// codons of the ORF of a stop node at p (far end q) in walk order: x(ci) = p + step * (ci + 1), ci = 0 .. ncod-1
__device__ __forceinline__ int orf_codons(const int p, const int q, const int strand, const int L) {
    if (strand == 1) {
        const int lowb = max(q + 1, 0);
        const int xmin = lowb + (((p - lowb) % 3) + 3) % 3;
        return xmin <= p - 3 ? (p - 3 - xmin) / 3 + 1 : 0;
    }
    const int highb = min(q - 1, L - 1);
    const int xmax = highb - (((highb - p) % 3) + 3) % 3;
    return xmax >= p + 3 ? (xmax - (p + 3)) / 3 + 1 : 0;
}

// one lane, one (short) ORF
// gil / il_stride / rank: the hexamer tables of the group's models interleaved, gil[hexamer * il_stride + rank[model]].  The
// models scored on a contig usually are neighbours in that order (a GC window over bins sorted by GC): their values for one
// hexamer then sit side by side and come in with one 16-byte load per pair instead of one 8-byte gather per model -- every
// lane walks its own ORF, so what a load costs is its address, not its width.
__device__ __forceinline__ void orf_serial(const OrfCtx& o, const ChainDesc* __restrict__ chains, const pga_training* __restrict__ models,
                           const ModelScoreConst* __restrict__ msc, const ChainArrays& ca, const double* __restrict__ gil, const int il_stride,
                           const int32_t* __restrict__ rank, const double* quad = nullptr, const int m_lo = 0, const int m_hi = 0x7fffffff) {
    // quad != nullptr: the tables of the four models [m_lo, m_lo + 4) of the contig sit interleaved in LDS, quad[hexamer * 4 + m]
    const uint8_t* __restrict__ d = o.d;
    const int strand = o.strand, step = o.step, p = o.p;
    const int m_end = min(o.cc.y, m_hi);
    for (int m0 = m_lo; m0 < m_end; m0 += CS_MODELS) {
        const int nm = min(CS_MODELS, m_end - m0);
        const double* gdc[CS_MODELS]; double* csp[CS_MODELS]; const ModelScoreConst* mcp[CS_MODELS];
#pragma unroll
        for (int m = 0; m < CS_MODELS; m++) {
            const ChainDesc ch = chains[o.cc.x + m0 + (m < nm ? m : 0)];
            gdc[m] = models[ch.model].gene_dc; csp[m] = ca.cscore_raw + ch.off; mcp[m] = &msc[ch.model];
        }
        double sum[CS_MODELS];
#pragma unroll
        for (int m = 0; m < CS_MODELS; m++) sum[m] = 0.0;
        int far = -1, mer = 0;
        unsigned long long sm0 = 0, sm1 = 0, sm2 = 0;           // codons below CS_LONG = 192 that are start nodes; beyond, the second pass reads the flags again
        // neighbours in the interleaved table?
        const int r0 = rank[chains[o.cc.x + m0].model];
        bool side_by_side = gil != nullptr;
#pragma unroll
        for (int m = 1; m < CS_MODELS; m++) if (m < nm) side_by_side = side_by_side && rank[chains[o.cc.x + m0 + m].model] == r0 + m;
        const double* __restrict__ row0 = gil + r0;
        // one codon: its hexamer `mer` joins the sums; a start node (k = its index in the contig) keeps the sums so far
        auto visit = [&](const int ci, const bool isnode, const int k) {
            if (quad != nullptr) {
                const double* q4 = quad + 4 * mer;               // one 32-byte row of LDS: two ds_read_b128
                sum[0] += q4[0]; sum[1] += q4[1]; sum[2] += q4[2]; sum[3] += q4[3];
            } else if (side_by_side) {
                struct P2 { double a, b; } u, v;
                const double* __restrict__ rp = row0 + (size_t)mer * il_stride;
                __builtin_memcpy(&u, rp, 16);
                sum[0] += u.a; sum[1] += u.b;                     // sums of models the lane does not have are never stored
                if (nm > 2) { __builtin_memcpy(&v, rp + 2, 16); sum[2] += v.a; sum[3] += v.b; }
            } else {
#pragma unroll
                for (int m = 0; m < CS_MODELS; m++) if (m < nm) sum[m] += gdc[m][mer];      // a load only where the lane has a model m
            }
            if (isnode) {
#pragma unroll
                for (int m = 0; m < CS_MODELS; m++) if (m < nm) csp[m][k] = sum[m];
                far = ci;
                if (ci < 64) sm0 |= 1ull << ci; else if (ci < 128) sm1 |= 1ull << (ci - 64); else if (ci < CS_LONG) sm2 |= 1ull << (ci - 128);
            }
        };
        if (o.ncod > 0) {
            const int j = p + step; mer = hexamer(d, j, strand);
            const bool isnode = o.node_at(j);
            visit(0, isnode, isnode ? o.node_index(j) : 0);
        }
        // Five codons at a time: their 15 bases and the start flags of those positions are 16 contiguous bytes of GroupArrays::df,
        // one (unaligned) load instead of four byte loads per codon -- every lane walks its own ORF, so each load
        // instruction costs the address path 64 distinct lines whatever its width.  Nothing a group needs is loaded while the
        // group is walked: its bytes were asked for two groups earlier, and the node indices of its flagged positions one group
        // earlier (from the flags, by then in registers) -- in lock step some lane meets a start node at nearly every codon, and
        // a load there would make every codon wait out a memory round trip.
        struct W16 { unsigned long long a, b; };
        auto group_lo = [&](const int c0) { return strand == 1 ? p - 3 * (c0 + 5) : p + 3 * c0 + 1; };       // lowest position of the group
        auto byte_of = [](const W16& w, const int k) { return (unsigned)((k < 8 ? w.a >> (8 * k) : w.b >> (8 * (k - 8))) & 0xffull); };
        auto flag_off = [&](const int u) { return strand == 1 ? 12 - 3 * u : 3 * u + 2; };                 // offset of codon u's node flag in the group
        const int fbit = strand == 1 ? 4 : 5;
        // the index of the walk's node at offset k of a group starting at lo: a reverse node follows the forward node of its position
        auto node_of = [&](const W16& w, const int lo, const int k) { return o.node_index(lo + k) + o.tbase; };
        W16 D1{0, 0}, D2{0, 0};
        int kq[5] = {0, 0, 0, 0, 0};
        const int ncod = o.ncod;
        // prologue: the bytes of the first and the second group; then the node indices of the first
        if (ncod > 1 && group_lo(1) >= 0) __builtin_memcpy(&D1, d + group_lo(1), 16);
        if (ncod > 6 && group_lo(6) >= 0) __builtin_memcpy(&D2, d + group_lo(6), 16);
        if (ncod > 1 && group_lo(1) >= 0) {
#pragma unroll
            for (int u = 0; u < 5; u++) if (1 + u < ncod && ((byte_of(D1, flag_off(u)) >> fbit) & 1u)) kq[u] = node_of(D1, group_lo(1), flag_off(u));
        }
        for (int c0 = 1; c0 < ncod; c0 += 5) {
            const int lo = group_lo(c0);
            if (lo < 0) {
                // the group hangs over the contig's first base (only codons beyond the ORF do): byte loads
                for (int ci = c0; ci < min(c0 + 5, ncod); ci++) {
                    const int j = p + step * (ci + 1);
                    const int lo3 = (d[j] & 3) | ((d[j + 1] & 3) << 2) | ((d[j + 2] & 3) << 4);          // strand == 1 here
                    mer = ((mer << 6) & 0xfc0) | lo3;
                    const bool isnode = o.node_at(j);
                    visit(ci, isnode, isnode ? o.node_index(j) : 0);
                }
                continue;
            }
            // everything of this group is here; ask for what the next groups need
            const W16 B = D1;
            int kc[5];
#pragma unroll
            for (int u = 0; u < 5; u++) kc[u] = kq[u] - o.tbase;
            D1 = D2;
            if (c0 + 5 < ncod) {
                const int lo1 = group_lo(c0 + 5);
                if (lo1 >= 0) {
#pragma unroll
                    for (int u = 0; u < 5; u++) if (c0 + 5 + u < ncod && ((byte_of(D1, flag_off(u)) >> fbit) & 1u)) kq[u] = node_of(D1, lo1, flag_off(u));
                }
                if (c0 + 10 < ncod) { const int lo2 = group_lo(c0 + 10); if (lo2 >= 0) __builtin_memcpy(&D2, d + lo2, 16); }
            }
            auto bytes_at = [](const W16& w, const int k) {          // the (up to 8) bytes from offset k on, k <= 13
                return k < 8 ? (w.a >> (8 * k)) | (k ? w.b << (64 - 8 * k) : 0ull) : w.b >> (8 * (k - 8));
            };
#pragma unroll
            for (int u = 0; u < 5; u++) {
                const int ci = c0 + u;
                if (ci >= ncod) break;
                const int k = strand == 1 ? 12 - 3 * u : 3 * u;             // offset of the codon's lowest position
                const unsigned t3 = (unsigned)bytes_at(B, k);
                const int b0 = t3 & 7, b1 = (t3 >> 8) & 7, b2 = (t3 >> 16) & 7;
                // rolling update: the three bases nearest to the walk direction are new (ref: _sequence.h:207-220)
                const int lo3 = strand == 1 ? (b0 & 3) | ((b1 & 3) << 2) | ((b2 & 3) << 4)
                                            : (comp2(b2) & 3) | ((comp2(b1) & 3) << 2) | ((comp2(b0) & 3) << 4);
                mer = ((mer << 6) & 0xfc0) | lo3;
                visit(ci, ((byte_of(B, flag_off(u)) >> fbit) & 1u) != 0, kc[u]);
            }
        }
        if (far < 0) continue;
        double run_c[CS_MODELS], run_l[CS_MODELS];
#pragma unroll
        for (int m = 0; m < CS_MODELS; m++) { run_c[m] = -10000.0; run_l[m] = -10000.0; }
        // the start nodes from the outermost inwards: codons from CS_LONG on by their flags (one 16-byte load per group of five, as
        // in the first pass), the others from the masks
        int ci = far, fc0 = -1;
        W16 G2{0, 0};
        while (ci >= 0) {
            if (ci >= CS_LONG) {
                const int c0 = 1 + 5 * ((ci - 1) / 5), u = ci - c0;
                const int lo = group_lo(c0);
                bool isnode;
                if (lo < 0) isnode = o.node_at(p + step * (ci + 1));
                else {
                    if (c0 != fc0) { __builtin_memcpy(&G2, d + lo, 16); fc0 = c0; }
                    const int kn = strand == 1 ? 12 - 3 * u : 3 * u + 2;
                    isnode = ((byte_of(G2, kn) >> fbit) & 1u) != 0;
                }
                if (!isnode) { ci--; continue; }
            } else {
                const int wsel = ci >> 6;
                const unsigned long long bits = (wsel == 2 ? sm2 : (wsel == 1 ? sm1 : sm0)) & ((2ull << (ci & 63)) - 1ull);
                if (!bits) { ci = wsel * 64 - 1; continue; }
                ci = wsel * 64 + 63 - __builtin_clzll(bits);
            }
            const int j = p + step * (ci + 1);
            const int k = o.node_index(j);
#pragma unroll
            for (int m = 0; m < CS_MODELS; m++) {
                if (m >= nm) continue;
                double cs = csp[m][k];
                if (cs > run_c[m]) run_c[m] = cs; else cs -= (run_c[m] - cs);
                double lfac = length_factor(mcp[m], ci + 2);
                if (lfac > run_l[m]) run_l[m] = lfac; else lfac -= fmax(fmin(run_l[m] - lfac, lfac), 0.0);
                if (lfac > 3.0 && cs < 0.5 * lfac) cs = 0.5 * lfac;
                cs += lfac;
                csp[m][k] = cs;
            }
            ci--;
        }
    }
}

// orf_serial for k_coding_score_quads: the four table columns [m0, m0 + 4) of the contig's quad sit in LDS (quad[hexamer * 4 + m]).
// Same sums, same order; written for instruction count, because 64 ORFs walk in lock step and whatever one lane needs at a
// codon every lane pays for: a group's 16 bases become sixteen 2-bit digits once (in walk order on either strand), its flags
// sixteen bits; a start node stores all four sums without asking which columns the contig has (the others go to
// ChainArrays::cs_sink); nothing is loaded inside a group (see orf_serial).
__device__ __forceinline__ void orf_serial_quad(const OrfCtx& o, const ChainDesc* __restrict__ chains, const ModelScoreConst* __restrict__ msc,
                                                const ChainArrays& ca, const double* __restrict__ quad, const int m0,
                                                unsigned long long* __restrict__ prof = nullptr) {
    unsigned long long tq = prof ? __builtin_readcyclecounter() : 0;
    auto qmark = [&](const int slot) {
        if (!prof) return;
        const unsigned long long now = __builtin_readcyclecounter();
        if ((threadIdx.x & 63) == 0) atomicAdd(&prof[slot], now - tq);
        tq = now;
    };
    const uint8_t* __restrict__ d = o.d;
    const int strand = o.strand, step = o.step, p = o.p, ncod = o.ncod;
    const bool fwd = strand == 1;
    const int nm = max(0, min(CS_MODELS, o.cc.y - m0));
    double* csp[CS_MODELS]; const ModelScoreConst* mcp[CS_MODELS];
#pragma unroll
    for (int m = 0; m < CS_MODELS; m++) {
        const ChainDesc ch = chains[o.cc.x + m0 + (m < nm ? m : 0)];
        csp[m] = m < nm ? ca.cscore_raw + ch.off : ca.cs_sink; mcp[m] = &msc[ch.model];
    }
    double sum[CS_MODELS];
#pragma unroll
    for (int m = 0; m < CS_MODELS; m++) sum[m] = 0.0;
    int far = -1, kfar = 0, mer = 0;
    unsigned long long sm0 = 0, sm1 = 0, sm2 = 0, sm3 = 0;  // the codons that are start nodes (every ORF walked here has at most CSQ_SERIAL_MAX = 256)
    // one codon (ci is the same in every lane): its hexamer joins the sums; a start node (k = its index in the contig) keeps them
    auto visit = [&](const int ci, const bool isnode, const int k) {
        const double* q4 = quad + 4 * mer;                   // one 32-byte row of LDS: two ds_read_b128
        sum[0] += q4[0]; sum[1] += q4[1]; sum[2] += q4[2]; sum[3] += q4[3];
        if (isnode) { csp[0][k] = sum[0]; csp[1][k] = sum[1]; csp[2][k] = sum[2]; csp[3][k] = sum[3]; }
        far = isnode ? ci : far; kfar = isnode ? k : kfar;
        const unsigned long long bit = isnode ? 1ull << (ci & 63) : 0ull;
        // ci is scalar: three scalar branches (the empty asm statements keep them apart; merged, the masks become a private array in scratch)
        if (ci < 64) { sm0 |= bit; asm volatile("; mask 0"); } else if (ci < 128) { sm1 |= bit; asm volatile("; mask 1"); }
        else if (ci < 192) { sm2 |= bit; asm volatile("; mask 2"); }
        else if (ci < CSQ_SERIAL_MAX) { sm3 |= bit; asm volatile("; mask 3"); }
    };
    // Where a start node sits in the contig's node list is a COUNT: nodes are in (position, strand) order, the stop node's index is
    // known, and the bytes the walk reads anyway carry the node flags of EVERY position it passes, either strand (bits 4 and 5).
    // A forward walk runs down from the stop: a start at x has index kstop - (nodes at positions x .. p - 1); a reverse walk runs up:
    // index (nodes before p) + (nodes at p .. x) - 1, the reverse node of a position being its last.  kb = the count at the boundary
    // of the group at hand, cnt = the nodes met inside it so far.  (Until round 5 the index came from GroupArrays::pre, one gather
    // per start node asked for a group ahead: 14 % of the kernel.)
    int kb = 0;
    const int ksgn = fwd ? -1 : 1, koff = fwd ? 0 : -1;
    if (ncod > 0) {
        const int j = p + step; mer = hexamer(d, j, strand);
        unsigned w4; __builtin_memcpy(&w4, fwd ? d + (p - 3) : d + p, 4);      // positions p - 3 .. p (forward) / p .. p + 3 (reverse)
        const int c012 = __popc(w4 & 0x00303030u);
        // codon 0 at p - 3 (p + 3); the group behind it begins where it ends
        int k0;
        if (fwd) { k0 = o.kstop - c012; kb = k0; }
        else { const int base = o.kstop - (int)((w4 >> 4) & 1u); k0 = base + c012 + (int)((w4 >> 28) & 1u); kb = base + __popc(w4 & 0x30303030u); }
        const bool isnode = ((w4 >> (fwd ? 4 : 29)) & 1u) != 0;              // forward: byte 0 (p - 3), bit 4; reverse: byte 3 (p + 3), bit 5
        visit(0, isnode, k0);
    }
    struct W16 { unsigned long long a, b; };
    auto group_lo = [&](const int c0) { return fwd ? p - 3 * (c0 + 5) : p + 3 * c0 + 1; };       // lowest position of the group
    // Codon u of a group: bases at offsets 12 - 3u .. 14 - 3u (forward strand, walking down) or 3u .. 3u + 2 (reverse, walking up,
    // read back to front and complemented); its node flag at offset 12 - 3u or 3u + 2.  With the reverse strand's sixteen digits
    // and flags turned end for end both become "13 - 3u (digits: 12 - 3u) counted from the end the walk starts at".
    auto digits_of = [&](const W16& w) -> unsigned {
        const uint4 v = make_uint4((unsigned)w.a, (unsigned)(w.a >> 32), (unsigned)w.b, (unsigned)(w.b >> 32));
        const unsigned q = (unsigned)pack16(v, !fwd);
        return fwd ? q : pairrev32(q);
    };
    // bit `bit` of the sixteen bytes as sixteen bits (4: forward-node flags, 5: reverse-node flags)
    auto flags_of = [&](const W16& w, const int bit) -> unsigned {
        auto bits4 = [&](const unsigned x) { const unsigned n = (x >> bit) & 0x01010101u; return (n | (n >> 7) | (n >> 14) | (n >> 21)) & 0xfu; };
        const unsigned f = bits4((unsigned)w.a) | (bits4((unsigned)(w.a >> 32)) << 4) | (bits4((unsigned)w.b) << 8) | (bits4((unsigned)(w.b >> 32)) << 12);
        return fwd ? f : (__brev(f) >> 16);
    };
    const int dsh = fwd ? 24 : 26, fsh = fwd ? 12 : 13;       // codon u: digits at bit dsh - 6u, flag at bit fsh - 3u
    const unsigned m3 = 0x00070007u << fsh;                   // the three positions of codon 0 in either half of (forward | reverse << 16) flags
    W16 D1{0, 0}, D2{0, 0};
    // prologue: the bytes of the first and of the second group
    if (ncod > 1 && group_lo(1) >= 0) __builtin_memcpy(&D1, d + group_lo(1), 16);
    if (ncod > 6 && group_lo(6) >= 0) __builtin_memcpy(&D2, d + group_lo(6), 16);
    qmark(10);
    // the group counter is the same in every lane (scalar): a lane whose ORF has ended idles through the rest, loading nothing
    int ncod_max = ncod;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) ncod_max = max(ncod_max, __shfl_xor(ncod_max, m, 64));
    ncod_max = __builtin_amdgcn_readfirstlane(ncod_max);
    for (int c0 = 1; c0 < ncod_max; c0 += 5) {
        const int lo = c0 < ncod ? group_lo(c0) : 0;
        if (lo < 0) {
            // the group hangs over the contig's first base (only codons beyond the ORF do): byte loads
            for (int ci = c0; ci < min(c0 + 5, ncod); ci++) {
                const int j = p + step * (ci + 1);
                const int lo3 = (d[j] & 3) | ((d[j + 1] & 3) << 2) | ((d[j + 2] & 3) << 4);          // forward strand here
                mer = ((mer << 6) & 0xfc0) | lo3;
                const bool isnode = o.node_at(j);
                visit(ci, isnode, isnode ? o.node_index(j) : 0);
            }
            continue;
        }
        // everything of this group is here; ask for what the group after the next needs
        const unsigned dg = digits_of(D1);
        const unsigned f4 = flags_of(D1, 4), f5 = flags_of(D1, 5);
        const unsigned fb = fwd ? f4 : f5, ff = f4 | (f5 << 16);
        D1 = D2;
        if (c0 + 10 < ncod) { const int lo2 = group_lo(c0 + 10); if (lo2 >= 0) __builtin_memcpy(&D2, d + lo2, 16); }
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < 5; u++) {
            const int ci = c0 + u;
            // a codon past the ORF's end (last group only) still joins the sums, which nobody reads any more; it is no start node
            mer = ((mer << 6) & 0xfc0) | ((dg >> (dsh - 6 * u)) & 63u);
            cnt += __popc(ff & (m3 >> (3 * u)));                  // the nodes at the codon's three positions
            visit(ci, ci < ncod && ((fb >> (fsh - 3 * u)) & 1u), kb + ksgn * cnt + koff);
        }
        kb += ksgn * cnt;
    }
    qmark(11);
    if (far < 0) return;
    double run_c[CS_MODELS], run_l[CS_MODELS];
#pragma unroll
    for (int m = 0; m < CS_MODELS; m++) { run_c[m] = -10000.0; run_l[m] = -10000.0; }
    // The start nodes from the outermost inwards.  Every ORF that walks here has at most CS_LONG codons (the 64-to-a-wave classes
    // of k_coding_score_quads), so the masks hold all of its start nodes and the outermost one's index is at hand (kfar).  What a
    // start node costs is round trips: its index (two loads), then per model the sum the walk stored and the length factor.  Asked
    // for one after the other -- a store of one model stands between the loads of the next, and the compiler must assume they
    // meet -- that was five trips per start node; here the next start node's index is asked for first, then all of this one's
    // sums and factors at once, and the four stores come last: one trip.  (Distinct start nodes have distinct k, the chains of a
    // contig distinct ranges: no load below can see a store it overtakes.)
    auto start_below = [&](int c) -> int {                    // the outermost start node at or below codon c, or -1
        while (c >= 0) {
            const int wsel = c >> 6;
            const unsigned long long bits = (wsel == 3 ? sm3 : (wsel == 2 ? sm2 : (wsel == 1 ? sm1 : sm0))) & ((2ull << (c & 63)) - 1ull);
            if (bits) return wsel * 64 + 63 - __builtin_clzll(bits);
            c = wsel * 64 - 1;
        }
        return -1;
    };
    if (far >= CSQ_SERIAL_MAX) __builtin_trap();              // cannot happen (see above): never walk past the masks
    // (two start nodes ahead: while this one is priced, the next one's sums and the index of the one after it are on their way)
    int ci = far, k = kfar;
    int c1 = start_below(ci - 1), k1 = 0;
    if (c1 >= 0) k1 = o.node_index(p + step * (c1 + 1));
    double cs[CS_MODELS];
#pragma unroll
    for (int m = 0; m < CS_MODELS; m++) cs[m] = csp[m][k];      // (columns the contig lacks: the sink)
    while (ci >= 0) {
        const int c2 = c1 >= 0 ? start_below(c1 - 1) : -1;
        int k2 = 0;
        if (c2 >= 0) k2 = o.node_index(p + step * (c2 + 1));
        double csn[CS_MODELS], lf[CS_MODELS];
#pragma unroll
        for (int m = 0; m < CS_MODELS; m++) { csn[m] = c1 >= 0 ? csp[m][k1] : 0.0; lf[m] = length_factor(mcp[m], ci + 2); }
#pragma unroll
        for (int m = 0; m < CS_MODELS; m++) {
            double c = cs[m], lfac = lf[m];
            if (c > run_c[m]) run_c[m] = c; else c -= (run_c[m] - c);
            if (lfac > run_l[m]) run_l[m] = lfac; else lfac -= fmax(fmin(run_l[m] - lfac, lfac), 0.0);
            if (lfac > 3.0 && c < 0.5 * lfac) c = 0.5 * lfac;
            cs[m] = c + lfac;
        }
#pragma unroll
        for (int m = 0; m < CS_MODELS; m++) if (m < nm) csp[m][k] = cs[m];
#pragma unroll
        for (int m = 0; m < CS_MODELS; m++) cs[m] = csn[m];
        ci = c1; k = k1; c1 = c2; k1 = k2;
    }
    qmark(12);
}

// the whole wavefront, one (long) ORF; `o` is wave-uniform
__device__ __forceinline__ void orf_wave(const OrfCtx& o, const int lane, const ChainDesc* __restrict__ chains, const pga_training* __restrict__ models,
                         const ModelScoreConst* __restrict__ msc, const ChainArrays& ca, const double* quad = nullptr, const int m_lo = 0,
                         const int m_hi = 0x7fffffff) {
    const double NEG_INF = -__builtin_huge_val();
    const int strand = o.strand, step = o.step, p = o.p, ncod = o.ncod;
    const int m_end = min(o.cc.y, m_hi);
    for (int m0 = m_lo; m0 < m_end; m0 += CS_MODELS) {
        const int nm = min(CS_MODELS, m_end - m0);
        const double* gdc[CS_MODELS]; double* csp[CS_MODELS]; const ModelScoreConst* mcp[CS_MODELS];
#pragma unroll
        for (int m = 0; m < CS_MODELS; m++) {
            const ChainDesc ch = chains[o.cc.x + m0 + (m < nm ? m : 0)];
            gdc[m] = models[ch.model].gene_dc; csp[m] = ca.cscore_raw + ch.off; mcp[m] = &msc[ch.model];
        }
        double sum[CS_MODELS];
#pragma unroll
        for (int m = 0; m < CS_MODELS; m++) sum[m] = 0.0;
        bool any_start = false;
        for (int c0 = 0; c0 < ncod; c0 += 64) {
            const int ci = c0 + lane;
            const bool valid = ci < ncod;
            const int x = p + step * (ci + 1);
            const int mer = valid ? hexamer(o.d, x, strand) : 0;
            const bool fl = valid && o.node_at(x);
            const int k = fl ? o.node_index(x) : 0;
            double v[CS_MODELS], pref[CS_MODELS];
#pragma unroll
            for (int m = 0; m < CS_MODELS; m++) { v[m] = (valid && m < nm) ? (quad != nullptr ? quad[4 * mer + m] : gdc[m][mer]) : 0.0; pref[m] = 0.0; }
            const int nv = min(64, ncod - c0);
            for (int l = 0; l < nv; l++) {
#pragma unroll
                for (int m = 0; m < CS_MODELS; m++) {
                    if (m >= nm) break;
                    sum[m] += readlane_f64_pl(v[m], l);
                    pref[m] = lane == l ? sum[m] : pref[m];
                }
            }
            if (fl) {
#pragma unroll
                for (int m = 0; m < CS_MODELS; m++) if (m < nm) csp[m][k] = pref[m];
            }
            any_start = any_start || __any(fl);
        }
        if (!any_start) continue;
        double carry_c[CS_MODELS], carry_l[CS_MODELS];
#pragma unroll
        for (int m = 0; m < CS_MODELS; m++) { carry_c[m] = -10000.0; carry_l[m] = -10000.0; }
        for (int c0 = ((ncod - 1) >> 6) << 6; c0 >= 0; c0 -= 64) {
            const int ci = c0 + (63 - lane);            // lane order = outermost first
            const bool valid = ci < ncod;
            const int x = p + step * (ci + 1);
            const bool fl = valid && o.node_at(x);
            const int k = fl ? o.node_index(x) : 0;
#pragma unroll
            for (int m = 0; m < CS_MODELS; m++) {
                if (m >= nm) break;
                const double cs0 = fl ? csp[m][k] : NEG_INF;
                const double lf0 = fl ? length_factor(mcp[m], ci + 2) : NEG_INF;
                const double inc_c = wave_incl_max(cs0, lane), inc_l = wave_incl_max(lf0, lane);
                double ex_c = __shfl_up(inc_c, 1, 64), ex_l = __shfl_up(inc_l, 1, 64);
                if (lane == 0) { ex_c = NEG_INF; ex_l = NEG_INF; }
                const double run_c = fmax(ex_c, carry_c[m]), run_l = fmax(ex_l, carry_l[m]);
                if (fl) {
                    double cs = cs0, lfac = lf0;
                    if (!(cs > run_c)) cs -= (run_c - cs);
                    if (!(lfac > run_l)) lfac -= fmax(fmin(run_l - lfac, lfac), 0.0);
                    if (lfac > 3.0 && cs < 0.5 * lfac) cs = 0.5 * lfac;
                    cs += lfac;
                    csp[m][k] = cs;
                }
                carry_c[m] = fmax(carry_c[m], readlane_f64_pl(inc_c, 63));
                carry_l[m] = fmax(carry_l[m], readlane_f64_pl(inc_l, 63));
            }
        }
    }
}

// Four long ORFs to a wavefront, sixteen lanes each (tables in LDS).  A whole wavefront per ORF (orf_wave) keeps the ordered sum
// short but pays its 64-step accumulation for one ORF; here the same steps serve four.  `o` is the same in the sixteen lanes of
// a group, `has` says whether the group holds an ORF at all.  Sums are the reference's: lane s of a group ends a round with
// carry + v0 + v1 + ... + vs, added in that order.
__device__ __forceinline__ void orf_quarter(const OrfCtx& o, const bool has, const int lane, const ChainDesc* __restrict__ chains,
                                            const ModelScoreConst* __restrict__ msc, const ChainArrays& ca, const double* quad, const int m0) {
    const double NEG_INF = -__builtin_huge_val();
    const int sl = lane & 15, gb = lane & 48;
    const int strand = o.strand, step = o.step, p = o.p, ncod = has ? o.ncod : 0;
    const int nm = has ? max(0, min(CS_MODELS, o.cc.y - m0)) : 0;
    double* csp[CS_MODELS]; const ModelScoreConst* mcp[CS_MODELS];
#pragma unroll
    for (int m = 0; m < CS_MODELS; m++) {
        const ChainDesc ch = chains[has ? o.cc.x + m0 + (m < nm ? m : 0) : 0];
        csp[m] = ca.cscore_raw + ch.off; mcp[m] = &msc[ch.model];
    }
    double carry[CS_MODELS];
#pragma unroll
    for (int m = 0; m < CS_MODELS; m++) carry[m] = 0.0;
    bool any_start = false;
    for (int c0 = 0; __any(c0 < ncod); c0 += 16) {
        const int ci = c0 + sl;
        const bool valid = ci < ncod;
        const int x = p + step * (ci + 1);
        const int mer = valid ? hexamer(o.d, x, strand) : 0;
        const bool fl = valid && o.node_at(x);
        const int k = fl ? o.node_index(x) : 0;
        double v[CS_MODELS], acc[CS_MODELS];
#pragma unroll
        for (int m = 0; m < CS_MODELS; m++) { v[m] = (valid && m < nm) ? quad[4 * mer + m] : 0.0; acc[m] = carry[m]; }
#pragma unroll
        for (int l = 0; l < 16; l++) {
            double t[CS_MODELS];
#pragma unroll
            for (int m = 0; m < CS_MODELS; m++) t[m] = __shfl(v[m], gb | l, 64);
            if (sl >= l) {
#pragma unroll
                for (int m = 0; m < CS_MODELS; m++) acc[m] += t[m];
            }
        }
        const int lastv = max(0, min(15, ncod - 1 - c0));
#pragma unroll
        for (int m = 0; m < CS_MODELS; m++) { const double nc = __shfl(acc[m], gb | lastv, 64); if (c0 < ncod) carry[m] = nc; }
        if (fl) {
#pragma unroll
            for (int m = 0; m < CS_MODELS; m++) if (m < nm) csp[m][k] = acc[m];
        }
        any_start = any_start || fl;
    }
    // penalties from the outermost start inwards (ref: lib.pyx:2182-2239), sixteen codons of a group at a time
    const bool grp_start = ((__ballot(any_start) >> gb) & 0xffffull) != 0ull;
    const int top = grp_start ? ((ncod - 1) >> 4) << 4 : -16;
    double carry_c[CS_MODELS], carry_l[CS_MODELS];
#pragma unroll
    for (int m = 0; m < CS_MODELS; m++) { carry_c[m] = -10000.0; carry_l[m] = -10000.0; }
    for (int c0 = top; __any(c0 >= 0); c0 -= 16) {
        const int ci = c0 + (15 - sl);              // lane order = outermost first
        const bool valid = c0 >= 0 && ci < ncod;
        const int x = p + step * (ci + 1);
        const bool fl = valid && o.node_at(x);
        const int k = fl ? o.node_index(x) : 0;
        // every model's sum and length factor asked for before the first store (a store between two loads holds the second back:
        // the compiler must assume they meet; the chains of a contig never do)
        double cs_in[CS_MODELS], lf_in[CS_MODELS];
#pragma unroll
        for (int m = 0; m < CS_MODELS; m++) {
            const bool mine = fl && m < nm;
            cs_in[m] = mine ? csp[m][k] : NEG_INF;
            lf_in[m] = mine ? length_factor(mcp[m], ci + 2) : NEG_INF;
        }
#pragma unroll
        for (int m = 0; m < CS_MODELS; m++) {
            const bool mine = fl && m < nm;
            const double cs0 = cs_in[m], lf0 = lf_in[m];
            double inc_c = cs0, inc_l = lf0;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const double a = __shfl_up(inc_c, off, 16), b = __shfl_up(inc_l, off, 16);
                if (sl >= off) { inc_c = fmax(inc_c, a); inc_l = fmax(inc_l, b); }
            }
            double ex_c = __shfl_up(inc_c, 1, 16), ex_l = __shfl_up(inc_l, 1, 16);
            if (sl == 0) { ex_c = NEG_INF; ex_l = NEG_INF; }
            const double run_c = fmax(ex_c, carry_c[m]), run_l = fmax(ex_l, carry_l[m]);
            if (mine) {
                double cs = cs0, lfac = lf0;
                if (!(cs > run_c)) cs -= (run_c - cs);
                if (!(lfac > run_l)) lfac -= fmax(fmin(run_l - lfac, lfac), 0.0);
                if (lfac > 3.0 && cs < 0.5 * lfac) cs = 0.5 * lfac;
                cs += lfac;
                csp[m][k] = cs;
            }
            carry_c[m] = fmax(carry_c[m], __shfl(inc_c, gb | 15, 64));
            carry_l[m] = fmax(carry_l[m], __shfl(inc_l, gb | 15, 64));
        }
    }
}

__global__ void __launch_bounds__(256)
k_coding_score(const ChainDesc* __restrict__ chains, const int2* __restrict__ contig_chains /* per contig: first chain, count */,
               const int32_t* __restrict__ node_contig_base, int n_contigs, int node_begin, int n_nodes,
               const uint8_t* __restrict__ dig, const ContigDesc* __restrict__ ct, GroupArrays ga,
               const pga_training* __restrict__ models, const ModelScoreConst* __restrict__ msc, ChainArrays ca,
               const double* __restrict__ gil, int il_stride, const int32_t* __restrict__ rank) {
    __shared__ int s_list[256];
    __shared__ int s_wtot[4];
    const int blk0 = node_begin + blockIdx.x * blockDim.x;
    const int me = blk0 + threadIdx.x;
    const bool is_stop = me < node_begin + n_nodes && ga.type[me] == PGA_T_STOP;
    const unsigned long long bm = __ballot(is_stop);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) s_wtot[wv] = __popcll(bm);
    __syncthreads();
    int off = 0;
    for (int k = 0; k < wv; k++) off += s_wtot[k];
    if (is_stop) s_list[off + __popcll(bm & ((1ull << lane) - 1ull))] = me;
    __syncthreads();
    const int cnt = s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
    if (wv * 64 >= cnt) return;                        // this wave has no stop node after compaction
    const bool mine = (int)threadIdx.x < cnt;
    OrfCtx o{};
    if (mine) {
        const int t = s_list[threadIdx.x];             // topology index of my stop node
        int lo = 0, hi = n_contigs - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (node_contig_base[mid] <= t) lo = mid; else hi = mid - 1; }
        const int c = lo;
        o.cc = contig_chains[c];
        const ContigDesc cd = ct[c];
        o.d = ga.df + cd.base;
        o.strand = ga.strand[t];
        o.c16 = ga.c16 + (size_t)ga.tile0[c] * 192;
        o.tbase = node_contig_base[c];
        o.p = ga.ndx[t]; o.q = ga.stop_val[t]; o.L = cd.len;
        o.step = o.strand == 1 ? -3 : 3;
        // codons of the ORF in walk order: x(ci) = p + step * (ci + 1), ci = 0 .. ncod-1
        if (o.strand == 1) {
            const int lowb = max(o.q + 1, 0);
            const int xmin = lowb + (((o.p - lowb) % 3) + 3) % 3;
            o.ncod = xmin <= o.p - 3 ? (o.p - 3 - xmin) / 3 + 1 : 0;
        } else {
            const int highb = min(o.q - 1, o.L - 1);
            const int xmax = highb - (((highb - o.p) % 3) + 3) % 3;
            o.ncod = xmax >= o.p + 3 ? (xmax - (o.p + 3)) / 3 + 1 : 0;
        }
        if (o.cc.y <= 0) o.ncod = 0;
    }
    const bool is_long = mine && o.ncod > CS_LONG;
    if (mine && !is_long && o.ncod > 0) orf_serial(o, chains, models, msc, ca, gil, il_stride, rank);
    // long ORFs of this wave, one after the other, all 64 lanes on each
    unsigned long long longs = __ballot(is_long);
    while (longs) {
        const int src = __builtin_ctzll(longs);
        longs &= longs - 1ull;
        OrfCtx w;
        const unsigned long long pd = (unsigned long long)o.d, pp = (unsigned long long)o.c16;
        auto bc64 = [&](unsigned long long v) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
            return ((unsigned long long)hi << 32) | lo;
        };
        w.d = (const uint8_t*)bc64(pd); w.c16 = (const int32_t*)bc64(pp);
        w.tbase = __builtin_amdgcn_readlane(o.tbase, src); w.p = __builtin_amdgcn_readlane(o.p, src); w.q = __builtin_amdgcn_readlane(o.q, src);
        w.L = __builtin_amdgcn_readlane(o.L, src); w.strand = __builtin_amdgcn_readlane(o.strand, src); w.step = __builtin_amdgcn_readlane(o.step, src);
        w.ncod = __builtin_amdgcn_readlane(o.ncod, src);
        w.cc.x = __builtin_amdgcn_readlane(o.cc.x, src); w.cc.y = __builtin_amdgcn_readlane(o.cc.y, src);
        orf_wave(w, lane, chains, models, msc, ca);
    }
}

// The same walks with the hexamer tables in LDS.  Work comes in TASKS: a run of (contig, first model) entries whose four
// models [m0, m0 + 4) are the same four columns q .. q + 3 of the group's interleaved table (contigs with neighbouring GC
// share them).  A workgroup copies those four columns into LDS once (128 KB, interleaved per hexamer) and then walks every
// ORF of its contigs against them: a codon step is two ds_read_b128 instead of two 64-line gathers through the texture
// path, which is what bounds the global-memory form (every lane walks its own ORF).
struct CsTask { int32_t q, first, count, _pad; };      // columns q .. q+3; entries [first, first + count)
struct CsEntry { int32_t contig, m0, first, count; };      // nodes [first, first + count) of the contig (a large contig is cut into several entries)
constexpr int CS_TASK_THREADS = 1024;
constexpr int CS_TASK_MAX_ENTRIES = 256;
constexpr int CS_ROUND = 8192;                         // nodes examined per round of a task (= the task size pga_cs_tasks aims at)
constexpr int CS_LIST = 5120;                          // stop nodes of a round: at most half of its nodes (every ORF with a stop node has a start node), plus
                                                       // up to twelve where a task holds a PIECE of a contig (ORFs across the cut; pga_cs_tasks allows CS_TASK_MAX_CUTS)
constexpr int CS_TASK_MAX_CUTS = 64;
// what the walks of a task need of one of its entries, staged in LDS once (a stop node then costs one round trip, not four)
struct CsEnt { int64_t base; int32_t len, tbase, ccx, ccy, m0, first, tile0, _pad; };
constexpr int CS_CLASSES = 12;                         // ORF length classes of a round
constexpr int CS_WAVE = 2048;                          // ORFs longer than this take a whole wave (orf_wave); the others walk 64 to a wave
__global__ void __launch_bounds__(CS_TASK_THREADS)
k_coding_score_quads(const CsTask* __restrict__ tasks, const CsEntry* __restrict__ entries, const ChainDesc* __restrict__ chains,
                     const int2* __restrict__ contig_chains, const int32_t* __restrict__ node_contig_base,
                     const uint8_t* __restrict__ dig, const ContigDesc* __restrict__ ct, GroupArrays ga,
                     const pga_training* __restrict__ models, const ModelScoreConst* __restrict__ msc, ChainArrays ca,
                     const double* __restrict__ gil, int il_stride, const int32_t* __restrict__ rank, const int cs_wave, unsigned long long* __restrict__ prof,
                     const int q_classes /* length classes 1 .. q_classes go four to a wave: 4 = ORFs beyond 192 codons, 3 = beyond 256 */) {
    extern __shared__ __attribute__((aligned(16))) double s_quad[];                  // [4096][4]
    __shared__ int s_pre[CS_TASK_MAX_ENTRIES + 1];      // first node (task-local numbering) of every entry
    __shared__ int s_list[CS_LIST];                     // node of the round (bits 0-12) | entry << 13
    __shared__ CsEnt s_ent[CS_TASK_MAX_ENTRIES];
    __shared__ int s_count, s_next_long, s_next_q, s_next, s_qend;
    __shared__ int s_cls[CS_CLASSES];
    const CsTask task = tasks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // PGA_CS_PROFILE=1: cycles per phase, summed over the waves of all workgroups (slot 7: workgroups)
    unsigned long long tp = prof ? __builtin_readcyclecounter() : 0;
    auto mark = [&](const int slot) {
        if (!prof) return;
        const unsigned long long now = __builtin_readcyclecounter();
        if (lane == 0) atomicAdd(&prof[slot], now - tp);
        tp = now;
    };
    if (prof && tid == 0) atomicAdd(&prof[7], 1ull);
    for (int idx = tid; idx < 4096 * 4; idx += CS_TASK_THREADS) {
        const int h = idx >> 2, k = idx & 3;
        s_quad[idx] = task.q + k < il_stride ? gil[(size_t)h * il_stride + task.q + k] : 0.0;
    }
    if (tid <= CS_TASK_MAX_ENTRIES) {
        int cnt = 0;
        if (tid < task.count) {
            const CsEntry en = entries[task.first + tid];
            const ContigDesc cd = ct[en.contig];
            const int2 cc = contig_chains[en.contig];
            s_ent[tid] = CsEnt{cd.base, cd.len, node_contig_base[en.contig], cc.x, cc.y, en.m0, en.first, ga.tile0[en.contig], 0};
            cnt = en.count;
        }
        s_pre[tid] = cnt;
    }
    __syncthreads();
    if (wv == 0) {          // counts -> first nodes: four entries per lane
        int v[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { v[q] = s_pre[4 * lane + q]; sum += v[q]; }
        int inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int x = __shfl_up(inc, off, 64); if (lane >= off) inc += x; }
        int run = inc - sum;
#pragma unroll
        for (int q = 0; q < 4; q++) { s_pre[4 * lane + q] = run; run += v[q]; }
        if (lane == 63) s_pre[CS_TASK_MAX_ENTRIES] = run;
    }
    __syncthreads();
    mark(0);
    const int total = s_pre[task.count];
    // Rounds of CS_ROUND nodes: every thread looks at CS_ROUND / 1024 nodes, the stop nodes among them (about one node in five) are
    // listed in LDS, then each wave takes 64 entries of the list at a time -- all sixteen waves walk, whatever the mix of nodes.
    for (int base = 0; base < total; base += CS_ROUND) {
        // the stop nodes of the round, listed by length class of their ORF (longest first): the 64 lanes of a wave then walk ORFs of
        // similar length instead of waiting for the longest of a random 64
        if (tid < CS_CLASSES) s_cls[tid] = 0;
        __syncthreads();
        int my_cls[CS_ROUND / CS_TASK_THREADS];             // length class | entry << 4, or -1
#pragma unroll
        for (int r = 0; r < CS_ROUND / CS_TASK_THREADS; r++) {
            const int local = base + tid + r * CS_TASK_THREADS;
            my_cls[r] = -1;
            if (local >= total) continue;
            int lo = 0, hi = task.count - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_pre[mid] <= local) lo = mid; else hi = mid - 1; }
            const int t = s_ent[lo].tbase + s_ent[lo].first + (local - s_pre[lo]);
            if (ga.type[t] != PGA_T_STOP) continue;
            const int ncod = orf_codons(ga.ndx[t], ga.stop_val[t], ga.strand[t], s_ent[lo].len);
            if (ncod <= 0) continue;
            my_cls[r] = ncod > cs_wave ? 0 : ncod > 512 ? 1 : ncod > 384 ? 2 : ncod > 256 ? 3 : ncod > 192 ? 4 : ncod > 128 ? 5 : ncod > 96 ? 6 :
                        ncod > 64 ? 7 : ncod > 48 ? 8 : ncod > 32 ? 9 : ncod > 16 ? 10 : 11;
            atomicAdd(&s_cls[my_cls[r]], 1);
            my_cls[r] |= lo << 4;
        }
        __syncthreads();
        if (tid == 0) {
            int acc = 0;
            for (int k = 0; k < CS_CLASSES; k++) { const int n = s_cls[k]; s_cls[k] = acc; acc += n; }
            if (acc > CS_LIST) __builtin_trap();             // cannot happen (see CS_LIST); never write past the list
            // classes 0: a wave each; 1 .. q_classes: four to a wave (3: ORFs beyond 256 codons -- round 6, fourth session: a lane walks up to
            // CSQ_SERIAL_MAX codons on its own, 655 -> 642 us per launch; PGA_CS_QCLASSES=4: beyond 192 as before); the others 64 to a wave
            s_count = acc; s_next_long = 0; s_next_q = s_cls[1]; s_qend = s_cls[q_classes + 1]; s_next = s_cls[q_classes + 1];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < CS_ROUND / CS_TASK_THREADS; r++)
            if (my_cls[r] >= 0) s_list[atomicAdd(&s_cls[my_cls[r] & 15], 1)] = (tid + r * CS_TASK_THREADS) | ((my_cls[r] >> 4) << 13);
        __syncthreads();
        mark(1);
        const int cnt = s_count, n_long = s_cls[0];       // after the placement s_cls[k] is where class k ends
        auto orf_of = [&](const int packed, int& m0) {
            OrfCtx o{};
            const int lo = packed >> 13, loc = base + (packed & 8191);
            const CsEnt en = s_ent[lo];
            m0 = en.m0;
            const int tt = en.tbase + en.first + (loc - s_pre[lo]);
            o.cc = make_int2(en.ccx, en.ccy);
            o.d = ga.df + en.base;
            o.strand = ga.strand[tt];
            o.c16 = ga.c16 + (size_t)en.tile0 * 192;
            o.tbase = en.tbase;
            o.p = ga.ndx[tt]; o.q = ga.stop_val[tt]; o.L = en.len; o.kstop = tt - en.tbase;
            o.step = o.strand == 1 ? -3 : 3;
            o.ncod = o.cc.y <= 0 ? 0 : orf_codons(o.p, o.q, o.strand, o.L);
            return o;
        };
        // Waves pull work until the round is empty (no wave waits for another before the end of the round).
        // long ORFs (the first n_long entries): one per wave at a time, all 64 lanes on each
        for (;;) {
            int k = 0;
            if (lane == 0) k = atomicAdd(&s_next_long, 1);
            k = __builtin_amdgcn_readfirstlane(k);
            if (k >= n_long) break;
            int m0;
            const OrfCtx w = orf_of(s_list[k], m0);          // the same entry in every lane
            if (w.ncod > 0) orf_wave(w, lane, chains, models, msc, ca, s_quad, m0, m0 + 4);
        }
        mark(2);
        // long ORFs, four to a wave
        const int q_end = s_qend;
        for (;;) {
            int k = 0;
            if (lane == 0) k = atomicAdd(&s_next_q, 4);
            k = __builtin_amdgcn_readfirstlane(k);
            if (k >= q_end) break;
            const int e = k + (lane >> 4);
            int m0 = 0;
            OrfCtx o{};
            const bool has = e < q_end;
            if (has) o = orf_of(s_list[e], m0);
            orf_quarter(o, has && o.ncod > 0, lane, chains, msc, ca, s_quad, m0);
        }
        mark(3);
        // the others, 64 of similar length per wave, longest class first
        for (;;) {
            int lb = 0;
            if (lane == 0) lb = atomicAdd(&s_next, 64);
            lb = __builtin_amdgcn_readfirstlane(lb);
            if (lb >= cnt) break;
            if (lb + lane < cnt) {
                int m0;
                const OrfCtx o = orf_of(s_list[lb + lane], m0);
                if (prof && lane == 0) atomicAdd(&prof[9], __builtin_readcyclecounter() - tp);
                if (o.ncod > 0) orf_serial_quad(o, chains, msc, ca, s_quad, m0, prof);
            }
            if (prof && lane == 0) atomicAdd(&prof[8], 1ull);
        }
        mark(4);
        __syncthreads();
        mark(5);
    }
}

// ---------------------------------------------------------------- ribosome binding site search
// The 45 bases upstream of a start (strand-local positions start-1 .. start-45) are read once into
// bit masks; bit u describes the base at position start - u.  Every model scored on the contig then
// searches the same registers.
struct UpWin {
    unsigned long long p0, p1;     // 2-bit code of the base at start - u (digit & 3 after strand mapping; unknown -> 2 as in the reference) at bits 2u, 2u + 1; p1: u >= 32
    unsigned long long zm;         // the same codes for u = 1 .. 21 in falling order: base u at bits 2 (21 - u) -- a motif read downstream is one shift
    unsigned isA, isG;             // u = 1 .. 20: exact identity on this strand, position inside the sequence (unknown bases match nothing)
    __device__ __forceinline__ int code(int u) const { return (int)(((u < 32 ? p0 : p1) >> (2 * (u & 31))) & 3ull); }
};

__device__ __forceinline__ unsigned even_bits(unsigned long long x) {                        // bit 2u of x -> bit u
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
    x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
    return (unsigned)(x | (x >> 16));
}

__device__ __forceinline__ UpWin load_upwin(const uint8_t* __restrict__ d, int L, int start, int strand) {
    UpWin w;
    if (start >= 48) {
        // 48 bytes in address order, three unaligned 16-byte loads (every lane reads its own window: an instruction costs the
        // address path 64 lines whatever its width): byte t at bits 2t of q0 (t < 32) / q1
        const uint8_t* __restrict__ a = strand == 1 ? d + (start - 48) : d + (L - start);
        uint4 v0, v1, v2;
        __builtin_memcpy(&v0, a, 16); __builtin_memcpy(&v1, a + 16, 16); __builtin_memcpy(&v2, a + 32, 16);
        const bool comp = strand != 1;
        const unsigned long long q0 = pack16(v0, comp) | (pack16(v1, comp) << 32), q1 = pack16(v2, comp);
        if (comp) {
            // byte t is the base at u = t + 1
            w.p0 = q0 << 2; w.p1 = (q1 << 2) | (q0 >> 62);
            w.zm = pairrev64(q0) >> 22;                  // pair t -> pair 31 - t, u = t + 1 <= 21 wanted at pair 21 - u = 20 - t
        } else {
            // byte t is the base at u = 48 - t
            const unsigned long long r = pairrev64(q0);                    // t < 32 -> pair 31 - t = u - 17
            const unsigned long long s2 = pairrev64(q1) >> 32;             // t = 32 .. 47 -> pair 47 - t = u - 1
            const unsigned long long lo = s2 | (r << 32), hi = r >> 32;    // base u at pair u - 1
            w.p0 = lo << 2; w.p1 = (hi << 2) | (lo >> 62);
            w.zm = (q0 >> 54) | (q1 << 10);              // u <= 21 <=> t >= 27: pair t -> pair t - 27 = 21 - u
        }
        w.zm &= (1ull << 42) - 1ull;
    } else {
        w.p0 = w.p1 = w.zm = 0;
        for (int u = 1; u <= 45; u++) {
            const int p = start - u;
            if (p < 0) break;
            const int raw = strand == 1 ? d[p] : d[L - 1 - p];
            const unsigned long long c2 = (unsigned long long)((strand == 1 ? raw : comp2(raw)) & 3);   // ref: _sequence.h:207-220
            if (u < 32) w.p0 |= c2 << (2 * u); else w.p1 |= c2 << (2 * (u - 32));
            if (u <= 21) w.zm |= c2 << (2 * (21 - u));
        }
    }
    // identity on this strand (ref: _sequence.h:45-55): A is code 0, G code 1 after the strand mapping; an unknown base is code 2
    const unsigned lo = even_bits(w.p0), hi = even_bits(w.p0 >> 1);
    const unsigned inr = start >= 31 ? 0xfffffffeu : ((2u << start) - 2u);          // u = 1 .. min(start, 31)
    w.isA = ~lo & ~hi & inr; w.isG = lo & ~hi & inr;
    return w;
}

// ref: lib.pyx:791-881 (exact) / 883-979 (one mismatch); mm selects the variant.  As in the
// reference the mismatch variant keeps the previous cur_val when no table row matches.
// The search itself does not look at the model: which motif bins match in the window is a property of the sequence.  The
// reference keeps, per window, the bin of greatest weight (ties to the larger bin) among bin 0 and the bins that match:
// sd_hits returns those bins as a bit mask (once per node), sd_pick chooses among them with one model's weights.
// A window is the six bases at strand-local positions pos .. pos + 5, pos = start - 20 + q for q = 0 .. 14; all the search
// looks at is, per base, "is the A (bases 0 and 3) / the G (others) of AGGAGG there": six bits.  sd_hits therefore is a function
// of (pattern, q, variant) alone and is tabulated once per context (k_sd_lut, 1920 words); the scoring kernel looks it up.
__device__ unsigned sd_hits(const int pat, const int q, const int mm) {
    int match[6], limit, cur = 0;
    unsigned hits = 1u;
    const int start = 20, pos = q;                      // only start - pos matters
    limit = min(6, start - 4 - pos);
#pragma unroll
    for (int i = 0; i < 6; i++) {
        match[i] = -10;
        if (i >= limit) continue;
        const bool hit = (pat >> i) & 1;
        if (!mm) { if (hit) match[i] = i % 3 == 0 ? 2 : 3; }
        else match[i] = hit ? (i % 3 == 0 ? 2 : 3) : (i % 3 == 0 ? -3 : -2);
    }
    for (int i = limit; i > (mm ? 4 : 2); i--) {
        for (int j = 0; j < limit + 1 - i; j++) {
            int ctr = -2, mism = 0, flag;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                if (k < j || k >= j + i) continue;
                ctr += match[k];
                if (mm && match[k] < 0) { mism++; if (k <= j + 1 || k >= j + i - 2) ctr -= 10; }
            }
            if (mm ? (mism != 1 || ctr < 6) : (ctr < 6)) continue;
            const int rdis = start - (pos + j + i);
            if (!mm) {
                if (rdis < 5) flag = i < 5 ? 2 : 1;
                else if (rdis < 11) flag = 0;
                else if (rdis < 13) flag = i < 5 ? 1 : 2;
                else if (rdis < 16) flag = 3;
                else continue;
                // rows: GGA, AGGA, GGAG, AGGAG, GGAGG, AGGAGG by distance class
                switch (ctr) {
                    case 6:  cur = flag == 0 ? 13 : flag == 1 ? 6 : flag == 2 ? 1 : 2; break;
                    case 8:  cur = flag == 0 ? 15 : flag == 1 ? 12 : flag == 2 ? 11 : 3; break;
                    case 9:  cur = flag == 0 ? 16 : flag == 1 ? 12 : flag == 2 ? 11 : 3; break;
                    case 11: cur = flag == 0 ? 22 : flag == 1 ? 21 : flag == 2 ? 20 : 10; break;
                    case 12: cur = flag == 0 ? 24 : flag == 1 ? 23 : flag == 2 ? 20 : 10; break;
                    case 14: cur = flag == 0 ? 27 : flag == 1 ? 26 : flag == 2 ? 25 : 10; break;
                    default: cur = 0;
                }
            } else {
                if (rdis < 5) flag = 1;
                else if (rdis < 11) flag = 0;
                else if (rdis < 13) flag = 2;
                else if (rdis < 16) flag = 3;
                else continue;
                switch (ctr) {
                    case 6: cur = flag == 0 ? 9 : flag == 1 ? 5 : flag == 2 ? 4 : 2; break;
                    case 7: cur = flag == 0 ? 14 : flag == 1 ? 8 : flag == 2 ? 7 : 2; break;
                    case 9: cur = flag == 0 ? 19 : flag == 1 ? 18 : flag == 2 ? 17 : 3; break;
                    default: break;
                }
            }
            hits |= 1u << cur;
        }
    }
    return hits;
}
#define PGA_SD_LUT 1920          // [variant 2][window 15][pattern 64]
__global__ void k_sd_lut(unsigned* __restrict__ lut) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < PGA_SD_LUT) lut[t] = sd_hits(t & 63, (t >> 6) % 15, (t >> 6) / 15);
}
__device__ __forceinline__ int sd_pick(unsigned hits, const double* __restrict__ w) {
    int maxv = 0;
    hits &= ~1u;
    while (hits) {
        const int cur = __builtin_ctz(hits);
        hits &= hits - 1u;
        if (w[cur] < w[maxv]) continue;
        if (w[cur] == w[maxv] && cur < maxv) continue;
        maxv = cur;
    }
    return maxv;
}

// ------------------------------------------------------------------------- start scoring
// One thread per node of the TOPOLOGY; it loads the upstream window once and scores the node for
// every model (chain) of its contig.  ref: lib.pyx:2331-2487 (Nodes._score), 2241-2277 (_rbs_score),
// 1556-1616 (_find_best_upstream_motif, stage 2), 1618-1650 (_score_upstream_composition).
// The reference mutates node.edge while it scans (lib.pyx:2424-2434) and the flag survives into
// the next model of a meta run; `first` and the index comparisons below reproduce the value each
// read would have seen.
// What the start scorer reads of a model, staged in LDS once per workgroup and model: every thread of a workgroup scores its node
// for the same model at the same time (a workgroup is 256 consecutive nodes, nearly always of one or two contigs), so the
// 32 upstream-composition weights, the RBS weights and the two small motif tables (3- and 4-base motifs: 26 of the 52 lookups
// of the motif search) come from LDS instead of 60 gathers per node and model.
struct StartModel {
    double st_wt, no_mot;
    int tt, uses_sd;
    double type_wt[3];
    double rbs_wt[28];
    double ups[32][4];      // 0.4 * st_wt * ups_comp
    double mk0[4][64];      // mot_wt[0][spacer class][3-base motif]
    double mk1[4][256];     // mot_wt[1][spacer class][4-base motif]
};

constexpr int SS_EDGE_SPAN = 12;        // the edge nodes of a contig are among its first and its last twelve nodes (see the scan below)
constexpr int SS_EDGE_CONTIGS = 4;      // contigs of a workgroup whose edge candidates are staged in LDS (the others read global memory)
constexpr int SS_MASK_WORDS = 8;        // models per pass over the workgroup's model set: 512

// What the scan for an edge node of the same ORF needs of candidate q of a contig (nodes [tb, tb + nn), length Lc): its stop_val
// and 4 (it exists) | 1 (edge node) | 2 (this pass converts it to one, ref: lib.pyx:2424-2434).
__device__ __forceinline__ int2 edge_candidate(const GroupArrays& ga, const int tb, const int nn, const int Lc, const int q, const bool closed) {
    const int j = q < SS_EDGE_SPAN ? q : nn - 2 * SS_EDGE_SPAN + q;
    if (j < 0 || j >= nn || (q >= SS_EDGE_SPAN && j < SS_EDGE_SPAN)) return make_int2(0, 0);       // short contig: each node once
    const int svj = ga.stop_val[tb + j], ej = ga.edge0[tb + j], ty = ga.type[tb + j], x = ga.ndx[tb + j], sj = ga.strand[tb + j];
    const bool cv = !closed && ty != PGA_T_STOP && !ej && ((x <= 2 && sj == 1) || (x >= Lc - 3 && sj == -1));
    return make_int2(svj, 4 | (ej ? 1 : 0) | (cv ? 2 : 0));
}

// The model loop keeps TWO staged models: while the workgroup scores its nodes for one, the next one it will need (known from a
// bit set of the models its contigs have) is on its way into the other buffer -- one barrier per model.
template <int OCC>
__global__ void __launch_bounds__(256, OCC)
k_score_starts(const ChainDesc* __restrict__ chains, const int2* __restrict__ contig_chains,
               const int32_t* __restrict__ node_contig_base, int n_contigs, int n_nodes,
               const uint8_t* __restrict__ dig, const ContigDesc* __restrict__ ct, GroupArrays ga,
               const pga_training* __restrict__ models, ChainArrays ca, ScoreParams sp, const unsigned* __restrict__ sd_lut,
               const int32_t* __restrict__ start_list /* or nullptr */, const int n_items) {
    // start_list: a thread per START node of the group (n_items of them; GroupArrays::start_list) instead of a thread per node.  A
    // stop node only gets its fields zeroed, and on sequence with short ORFs nearly half of the nodes are stop nodes: with a thread per
    // node every wavefront dragged them through the model loop, and the kernel's time is its workgroup count times the latency of
    // that loop.  The stop nodes' zeros are written by the workgroup at the end, a thread per node of its range (see there).
    __shared__ unsigned long long s_present[SS_MASK_WORDS];
    __shared__ unsigned s_lut[PGA_SD_LUT];
    __shared__ int2 s_edge[SS_EDGE_CONTIGS][2 * SS_EDGE_SPAN];
    __shared__ StartModel SMB[2];
    unsigned long long* __restrict__ prof = sp.prof;
    unsigned long long tp = prof ? __builtin_readcyclecounter() : 0;
    // the first active lane books the time since the wave's last mark; conv: all lanes are here again, some may have skipped marks
    auto mark = [&](const int slot, const bool conv = false) {
        if (!prof) return;
        if (conv) {
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { const unsigned long long o = __shfl_xor(tp, m, 64); tp = o > tp ? o : tp; }
        }
        const unsigned long long now = __builtin_readcyclecounter();
        if ((int)(threadIdx.x & 63) == __builtin_ctzll(__ballot(1))) atomicAdd(&prof[slot], now - tp);
        tp = now;
    };
    const int tid = threadIdx.x;
    const int blk0 = blockIdx.x * blockDim.x;
    const bool listed = start_list != nullptr;
    const int n_work = listed ? n_items : n_nodes;
    const bool in_range = blk0 + tid < n_work;
    const int t = listed ? start_list[min(blk0 + tid, n_work - 1)] : blk0 + tid;       // the node this thread scores
    const bool closed = sp.closed != 0, is_meta = sp.is_meta != 0;
    if (tid < SS_MASK_WORDS) s_present[tid] = 0ull;
    const int c0 = ga.contig_of[listed ? start_list[blk0] : blk0];                     // blk0 < n_work; the same address in every thread
    const int c = in_range ? ga.contig_of[t] : c0;
    const int2 cc = in_range ? contig_chains[c] : make_int2(0, 0);
    const bool has = in_range && cc.y > 0;
    // (the model set is empty from here on)  A launch over the re-scored winners finds most workgroups without a chain.
    if (!__syncthreads_or(has) && !listed) return;      // (with a work list the workgroup still has the stop nodes of its range to clear)
    for (int k = tid; k < PGA_SD_LUT; k += blockDim.x) s_lut[k] = sd_lut[k];
    // the thread's next chain (what the scorer reads of its descriptor), fetched one model ahead
    int64_t ch_off = 0, ch_raw = -1; int ch_model = 0x7fffffff, ch_first = 0;
    auto fetch_chain = [&](const int k) { const ChainDesc* __restrict__ q = &chains[k]; ch_off = q->off; ch_model = q->model; ch_first = q->first; ch_raw = q->raw_off; };
    if (has) fetch_chain(cc.x);
    const int tbase = node_contig_base[c];
    const int i = t - tbase, n = has ? node_contig_base[c + 1] - tbase : 1;
    const int type = has ? ga.type[t] : PGA_T_STOP;
    const int e0 = has ? ga.edge0[t] : 0;
    const bool is_start = has && type != PGA_T_STOP;
    // the models the workgroup scores, as a bit set (models [mb, mb + 512)): the first thread of every contig enters its chains
    // (with a work list the first node of a contig may be a stop node: there the thread whose neighbour sits on another contig enters)
    const int c_prev = __shfl_up(c, 1, 64);
    const bool enters = has && (listed ? ((tid & 63) == 0 || c != c_prev) : (tid == 0 || i == 0));
    auto enter_models = [&](const int mb) {
        if (!enters) return;
        for (int m = 0; m < cc.y; m++) {
            const int rel = chains[cc.x + m].model - mb;
            if (rel >= 0 && rel < sp.models_per_pass) atomicOr(&s_present[rel >> 6], 1ull << (rel & 63));
        }
    };
    enter_models(0);
    // the edge candidates of the workgroup's first contigs
    if (tid < SS_EDGE_CONTIGS * 2 * SS_EDGE_SPAN) {
        const int ec = c0 + tid / (2 * SS_EDGE_SPAN), q = tid % (2 * SS_EDGE_SPAN);
        int2 e = make_int2(0, 0);
        if (ec < n_contigs) { const int tb = node_contig_base[ec]; e = edge_candidate(ga, tb, node_contig_base[ec + 1] - tb, ct[ec].len, q, closed); }
        s_edge[tid / (2 * SS_EDGE_SPAN)][q] = e;
    }
    mark(0, true);
    // ---- what does not depend on the model (start nodes only)
    int L = 3, ndx = 0, sv = 0, strand = 1, start = 0;
    const uint8_t* __restrict__ d = dig;
    bool conv = false, ups_first = false, ups_later = false, ups_near_edge = false;
    UpWin W{0, 0, 0, 0, 0};
    unsigned long long ucodes = 0ull; int nups = 0;
    long orf = 1;
    int stop3 = 0;
    if (is_start) {
        const ContigDesc cd = ct[c];
        L = cd.len;
        d = dig + cd.base;
        ndx = ga.ndx[t]; sv = ga.stop_val[t]; strand = ga.strand[t];
        // does this pass turn the node into an edge node?  (ref: lib.pyx:2424-2434)
        conv = !closed && !e0 && ((ndx <= 2 && strand == 1) || (ndx >= L - 3 && strand == -1));
        if (conv && sp.conv_flag != nullptr) sp.conv_flag[c] = 1;
        start = strand == 1 ? ndx : L - 1 - ndx;     // strand-local start position
        W = load_upwin(d, L, start, strand);
        // upstream composition (ref: lib.pyx:1618-1650): the bases at u = 1, 2 and 15 .. 44 as far as they lie inside the sequence
        ucodes = ((W.p0 >> 2) & 0xfull) | ((W.p0 >> 30) << 4) | ((W.p1 & 0x3ffffffull) << 38);
        nups = start >= 2 ? 2 + min(max(start - 14, 0), 30) : (start >= 1 ? 1 : 0);
        orf = ndx > sv ? ndx - sv : sv - ndx;
        // the three bases at the ORF's stop, strand-local order: whether they are a stop codon depends on the model's table
        const int s0 = strand == 1 ? sv : L - 1 - sv;
        stop3 = sbase(d, L, s0, strand) | (sbase(d, L, s0 + 1, strand) << 8) | (sbase(d, L, s0 + 2, strand) << 16);
    }
    mark(1, true);
    __syncthreads();            // the model set and the edge candidates are complete
    if (is_start) {
        // does an edge node share this ORF?  (lib.pyx:2413-2422; the answer depends on whether the edge
        // conversion of this pass has already reached that node, hence the two variants)
        if (!closed && ndx <= 2 && strand == 1) ups_near_edge = true;
        else if (!closed && ndx >= L - 3 && strand == -1) ups_near_edge = true;
        else if ((i < 500 && strand == 1) || (i + 500 >= n && strand == -1)) {
            // The reference scans the 500 nodes before a forward start (after a reverse one) for an edge node, or one that this
            // pass converts to an edge node, of the same ORF.  Such nodes sit on the first or last three positions of the contig
            // (ref: lib.pyx:2413-2434, node.c add_nodes), i.e. among the first or last SS_EDGE_SPAN nodes (one node per position
            // and strand): those are the only candidates worth a look.
            const int ecx = c - c0;
            for (int q = 0; q < 2 * SS_EDGE_SPAN; q++) {
                const int j = q < SS_EDGE_SPAN ? q : n - 2 * SS_EDGE_SPAN + q;
                if (strand == 1 ? j >= i : j <= i) continue;
                const int2 e = ecx < SS_EDGE_CONTIGS ? s_edge[ecx][q] : edge_candidate(ga, tbase, n, L, q, closed);
                if (!(e.y & 4) || sv != e.x) continue;
                if (e.y & 1) { ups_first = ups_later = true; }
                else if (e.y & 2) { if (strand == 1) ups_first = true; ups_later = true; }   // after i: counts only once an earlier model of the run converted it
            }
        }
    }
    mark(2, true);
    int tt_cached = -1; bool stop_missing = false;
    // per search window of the RBS search: the six "is the A / G of AGGAGG there" bits (five windows per word); the bins that
    // match come from the table in LDS when a model asks
    bool have_pats = false;
    unsigned pats[3] = {0u, 0u, 0u};
    int mi = 0;                                                 // the thread's next chain
    // the lowest model of the set above `after` (relative to mb), or -1
    const int present_words = min(SS_MASK_WORDS, (min(sp.models_per_pass, sp.n_models) + 63) >> 6);      // (sixteen models: one word, not eight)
    auto next_present = [&](const int after) {
        for (int w = (after + 1) >> 6; w < present_words; w++) {
            unsigned long long bits = s_present[w];
            if (w == (after + 1) >> 6) bits &= ~0ull << ((after + 1) & 63);
            if (bits) return w * 64 + __builtin_ctzll(bits);
        }
        return -1;
    };
    // staging a model: every thread fetches its share (r0: the scalars, the RBS / type / upstream weights; r1: one entry of the
    // 3-base motif table) early and stores it late; the 4-base motif table (four entries per thread) goes through in one piece
    struct StageRegs { double r0, r1; int sd; };
    auto stage_fetch = [&](const int model) {
        const pga_training* __restrict__ tmg = &models[model];
        StageRegs R{0.0, 0.0, tmg->uses_sd};
        if (tid == 0) R.r0 = tmg->st_wt;
        else if (tid == 1) R.r0 = tmg->no_mot;
        else if (tid == 2) R.r0 = __hiloint2double(tmg->uses_sd, tmg->trans_table);
        else if (tid >= 4 && tid < 7) R.r0 = tmg->type_wt[tid - 4];
        else if (tid >= 32 && tid < 60) R.r0 = tmg->rbs_wt[tid - 32];
        else if (tid >= 128) R.r0 = 0.4 * tmg->st_wt * (&tmg->ups_comp[0][0])[tid - 128];      // the term the upstream composition adds (ref: lib.pyx:1618-1650), multiplied once per workgroup and model
        R.r1 = tmg->mot_wt[0][tid >> 6][tid & 63];          // asked for whatever uses_sd says: no load waits for another
        return R;
    };
    auto stage_store = [&](const int model, const StageRegs& R, StartModel& S) {
        if (tid == 0) S.st_wt = R.r0;
        else if (tid == 1) S.no_mot = R.r0;
        else if (tid == 2) { S.tt = __double2loint(R.r0); S.uses_sd = __double2hiint(R.r0); }
        else if (tid >= 4 && tid < 7) S.type_wt[tid - 4] = R.r0;
        else if (tid >= 32 && tid < 60) S.rbs_wt[tid - 32] = R.r0;
        else if (tid >= 128) (&S.ups[0][0])[tid - 128] = R.r0;
        if (!R.sd) {
            const pga_training* __restrict__ tmg = &models[model];
            S.mk0[tid >> 6][tid & 63] = R.r1;
            double v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) { const int e = tid + 256 * r; v[r] = tmg->mot_wt[1][e >> 8][e & 255]; }
#pragma unroll
            for (int r = 0; r < 4; r++) { const int e = tid + 256 * r; S.mk1[e >> 8][e & 255] = v[r]; }
        }
    };
    int buf = 0;
    for (int mb = 0; ; ) {
        int rel = next_present(-1);
        if (rel >= 0) { const StageRegs R = stage_fetch(mb + rel); stage_store(mb + rel, R, SMB[buf]); }
        __syncthreads();
        while (rel >= 0) {
            const int cur = mb + rel;
            const int nrel = next_present(rel);
            StageRegs RN{0.0, 0.0, 1};
            if (nrel >= 0) RN = stage_fetch(mb + nrel);
            const StartModel& SM = SMB[buf];
            const bool mine = has && ch_model == cur;
            mark(3, true);
            if (mine) {
                const int64_t g = ch_off + i;
                const int first = ch_first;
                const double cs_raw = is_start ? ca.cscore_raw[(ch_raw >= 0 ? ch_raw : ch_off) + i] : 0.0;
                mi++;
                if (mi < cc.y) fetch_chain(cc.x + mi); else ch_model = 0x7fffffff;
                if (!is_start) {        // stop nodes carry no start scores (reset_node_scores)
                    ca.edge[g] = (uint8_t)e0;
                    ca.cscore[g] = 0.0; ca.sscore[g] = 0.0; ca.rscore[g] = 0.0; ca.uscore[g] = 0.0; ca.tscore[g] = 0.0; ca.mot_score[g] = 0.0;
                    ca.mot_ndx[g] = 0; ca.mot_len[g] = 0; ca.mot_spacer[g] = 0; ca.mot_spacendx[g] = 0; ca.rbs[2 * g] = 0; ca.rbs[2 * g + 1] = 0;
                    if (sp.cs_out != nullptr) sp.cs_out[g] = 0.0;
                } else {
        const pga_training* __restrict__ tm = &models[cur];          // the large motif tables stay in global memory
        const double st_wt = SM.st_wt;
        const int tt = SM.tt;
        if (tt != tt_cached) {
            tt_cached = tt;
            stop_missing = !codon_is_stop(stop3 & 0xff, (stop3 >> 8) & 0xff, (stop3 >> 16) & 0xff, tt);
        }
        const bool edge_in = e0 || (conv && !first);
        mark(4);
        int rbs0 = 0, rbs1 = 0, m_ndx = 0, m_len = 0, m_sp = 0, m_si = 0;
        double m_score = 0.0;
        const bool search_sd = !edge_in && SM.uses_sd, search_mot = !edge_in && !SM.uses_sd;
        // The motif of k + 3 bases whose first base sits u0 = 18 + k - t2 upstream (the reference's j = start - u0, ascending j),
        // longest motifs first: every shift and spacer class below is a constant of the unrolled bodies.  6- and 5-base motifs come
        // from the model's tables in global memory, 4- and 3-base motifs from LDS.  (Asking for both batches of gathers at once, or
        // computing the upstream composition under the first, costs more in spilled registers than the overlap returns.)
        auto midx = [&](const int k, const int t2) { return (int)((W.zm >> (2 * (21 - (18 + k - t2)))) & ((1ull << (2 * (k + 3))) - 1ull)); };
        auto msi = [](const int t2) { return t2 <= 2 ? 3 : (t2 <= 4 ? 2 : (t2 >= 11 ? 1 : 0)); };      // u0 >= 16 + k, >= 14 + k, <= 7 + k
        // upstream composition (ref: lib.pyx:1618-1650)
        // (which weight of a row: the node's codes at u = 1, 2, 15 .. 44, packed once per node -- `ucodes`, `nups` -- instead of dug out of
        //  the window for every model; the sum is the reference's, term by term)
        auto upstream = [&]() {
            double v = 0.0;              // (SM.ups holds 0.4 * st_wt * ups_comp: stage_fetch)
            unsigned lo = (unsigned)ucodes, hi = (unsigned)(ucodes >> 32);
            if (nups == 32) {
#pragma unroll
                for (int q = 0; q < 16; q++) { v += SM.ups[q][(lo >> (2 * q)) & 3u]; if ((q & 7) == 7) __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int q = 0; q < 16; q++) { v += SM.ups[16 + q][(hi >> (2 * q)) & 3u]; if ((q & 7) == 7) __builtin_amdgcn_sched_barrier(0); }
            } else {
                for (int q = 0; q < nups; q++) v += SM.ups[q][(int)(ucodes >> (2 * q)) & 3];
            }
            return v;
        };
        double u = 0.0;
        if (search_sd) {
                if (!have_pats) {
                    have_pats = true;
                    const unsigned hasA = W.isA, hasG = W.isG;
#pragma unroll
                    for (int q = 0; q < 15; q++) {
                        // base i of the window is bit 5 - i of the six bits from position start - (j + 5) up
                        const unsigned a6 = (hasA >> (15 - q)) & 63u, g6 = (hasG >> (15 - q)) & 63u;
                        const unsigned pat = ((a6 >> 5) & 1u) | (((g6 >> 4) & 1u) << 1) | (((g6 >> 3) & 1u) << 2) | (((a6 >> 2) & 1u) << 3) |
                                             (((g6 >> 1) & 1u) << 4) | ((g6 & 1u) << 5);
                        pats[q / 5] |= pat << (6 * (q % 5));
                    }
                }
#pragma unroll
                for (int q = 0; q < 15; q++) {
                    // the reference skips windows starting before the sequence on the forward strand only; on the reverse
                    // strand it tests "j >= slen", never true, so windows hanging off the end are searched with the missing
                    // bases matching nothing (ref: lib.pyx:2256-2275)
                    if (start - 20 + q < 0 && strand == 1) continue;
                    const unsigned pat = (pats[q / 5] >> (6 * (q % 5))) & 63u;
                    const unsigned he = s_lut[(q << 6) | pat], hm = s_lut[((15 + q) << 6) | pat];
                    if (he > 1u) { const int a = sd_pick(he, SM.rbs_wt); if (a > rbs0) rbs0 = a; }
                    if (hm > 1u) { const int b = sd_pick(hm, SM.rbs_wt); if (b > rbs1) rbs1 = b; }
                }
        } else if (search_mot) {
            double bsc = -100.0; int bsp = 0, bsi = 0, blen = 0, bndx = 0;
#pragma unroll 1
            for (int k = 3; k >= 0; k--) {
                const int stride = k == 0 ? 64 : (k == 1 ? 256 : 4096);        // entries per spacer class of the table read
                const double* __restrict__ gt = &tm->mot_wt[k][0][0];
                const double* lt = k == 1 ? &SM.mk1[0][0] : &SM.mk0[0][0];
                double scv[13];
                if (k >= 2) {
#pragma unroll
                    for (int t2 = 0; t2 < 13; t2++) scv[t2] = 18 + k - t2 > start ? -1000.0 : gt[msi(t2) * stride + midx(k, t2)];
                } else {
#pragma unroll
                    for (int t2 = 0; t2 < 13; t2++) scv[t2] = 18 + k - t2 > start ? -1000.0 : lt[msi(t2) * stride + midx(k, t2)];
                }
#pragma unroll
                for (int t2 = 0; t2 < 13; t2++)
                    if (scv[t2] > bsc) { bsc = scv[t2]; bsi = msi(t2); bsp = 15 - t2; blen = k + 3; bndx = midx(k, t2); }
            }
            if (bsc == -4.0 || bsc < SM.no_mot + 0.69) { m_score = SM.no_mot; }
            else { m_ndx = bndx; m_len = blen; m_si = bsi; m_sp = bsp & 15; m_score = bsc; }
        }
        if (!edge_in) u = upstream();
        mark(5);
        double edge_gene = 0;
        if (edge_in) edge_gene += 1;
        if (stop_missing) edge_gene += 1;

        double tscore, uscore, rscore, sscore, cscore = cs_raw;
        if (edge_in) {
            tscore = 0.74 * st_wt / edge_gene; uscore = 0.0; rscore = 0.0;
        } else {
            tscore = SM.type_wt[type] * st_wt;
            const double r1 = SM.rbs_wt[rbs0], r2 = SM.rbs_wt[rbs1];
            const double sd = fmax(r1, r2) * st_wt;
            if (SM.uses_sd) rscore = sd;
            else { rscore = st_wt * m_score; if (rscore < sd && SM.no_mot > -0.5) rscore = sd; }
            uscore = u;
            if (ups_near_edge || (first ? ups_first : ups_later)) uscore += -1.00 * st_wt;
        }
        mark(6);
        bool edge_now = edge_in;
        if (conv && !edge_in) {
            edge_gene += 1; edge_now = true; tscore = 0.0;
            uscore = 0.74 * st_wt / edge_gene; rscore = 0.0;
        }
        if (!edge_now && edge_gene == 1) uscore -= 0.5 * 0.74 * st_wt;
        if (edge_gene == 0 && orf < 250) {
            const double negf = 250.0 / (float)orf, posf = (float)orf / 250.0;
            rscore *= rscore < 0 ? negf : posf;
            uscore *= uscore < 0 ? negf : posf;
            tscore *= tscore < 0 ? negf : posf;
        }
        if (is_meta && L < 3000 && edge_gene == 0 && (cscore < 5.0 || orf < 120))
            cscore -= 7.5 * fmax(0.0, (3000.0 - L) / 2700.0);
        sscore = tscore + rscore + uscore;
        if (cscore < 0.0) {
            if (edge_gene > 0 && !edge_now) {
                if (!is_meta || L > 1500) sscore -= st_wt; else sscore -= 10.31 - 0.004 * L;
            } else if (is_meta && L < 3000 && edge_now) {
                const double mml = sqrt((double)L) * 5.0;
                if (orf >= mml) { if (cscore >= 0) cscore = -1.0; sscore = 0.0; uscore = 0.0; }
            } else sscore -= 0.5;
        } else if (is_meta && cscore < 5.0 && orf < 120 && sscore < 0.0) sscore -= st_wt;

        ca.cscore[g] = cscore; ca.sscore[g] = sscore; ca.rscore[g] = rscore; ca.uscore[g] = uscore; ca.tscore[g] = tscore;
        ca.mot_score[g] = m_score; ca.mot_ndx[g] = m_ndx;
        ca.mot_len[g] = (uint8_t)m_len; ca.mot_spacer[g] = (uint8_t)m_sp; ca.mot_spacendx[g] = (uint8_t)m_si;
        ca.rbs[2 * g] = (uint8_t)rbs0; ca.rbs[2 * g + 1] = (uint8_t)rbs1;
        ca.edge[g] = (uint8_t)edge_now;
        if (sp.cs_out != nullptr) sp.cs_out[g] = cscore + sscore;
                }

            }
            mark(8, true);
            if (nrel >= 0) stage_store(mb + nrel, RN, SMB[buf ^ 1]);
            mark(9, true);
            __syncthreads();            // everybody is done with this model, and the next one is staged
            mark(10, true);
            buf ^= 1; rel = nrel;
        }
        mb += sp.models_per_pass;
        if (mb >= sp.n_models) break;
        // (more models than a pass holds: the next set)
        if (tid < SS_MASK_WORDS) s_present[tid] = 0ull;
        __syncthreads();
        enter_models(mb);
        __syncthreads();
    }
    // (listed) The stop nodes: no start scores (reset_node_scores), in every chain of their contig.  The workgroup answers for the nodes
    // behind the last start node of the workgroup before it up to its own last start node (the last workgroup: to the end), its
    // threads take them in node order -- the same stores the stop nodes' own threads made when every node had one, without the trip
    // through the model loop.
    if (listed) {
        const int z0 = blk0 == 0 ? 0 : start_list[blk0 - 1] + 1;
        const int z1 = blk0 + 256 >= n_work ? n_nodes : start_list[blk0 + 255] + 1;
        for (int z = z0 + tid; z < z1; z += 256) {
            if (ga.type[z] != PGA_T_STOP) continue;
            const int cz = ga.contig_of[z];
            const int2 zc = contig_chains[cz];
            const int zb = node_contig_base[cz];
            const uint8_t ez = ga.edge0[z];
            for (int m = 0; m < zc.y; m++) {
                const int64_t gz = chains[zc.x + m].off + (z - zb);
                ca.edge[gz] = ez;
                if (sp.lean_stops) continue;
                ca.cscore[gz] = 0.0; ca.sscore[gz] = 0.0; ca.rscore[gz] = 0.0; ca.uscore[gz] = 0.0; ca.tscore[gz] = 0.0; ca.mot_score[gz] = 0.0;
                ca.mot_ndx[gz] = 0; ca.mot_len[gz] = 0; ca.mot_spacer[gz] = 0; ca.mot_spacendx[gz] = 0; ca.rbs[2 * gz] = 0; ca.rbs[2 * gz + 1] = 0;
                if (sp.cs_out != nullptr) sp.cs_out[gz] = 0.0;
            }
        }
    }
}

// ------------------------------------------------------------------- overlapping starts
// ref: lib.pyx:2279-2329 (Nodes._record_overlapping_starts with flag = 1)
// The three overlapping starts of stop node i (not an edge stop) of a chain.
struct OvlChain {
    const int32_t* __restrict__ ndx; const int32_t* __restrict__ stv; const uint8_t* __restrict__ typ; const int8_t* __restrict__ str;   // topology of the contig
    const double* __restrict__ cs; const double* __restrict__ ss; const double* __restrict__ rs; const double* __restrict__ us;           // the chain's scores
    int n;
    const double* __restrict__ css = nullptr;    // or: cscore + sscore as the start scorer left it for the wave-batch scorer (one load instead of two)
};
// Which of the first OV_SPEC neighbours of stop node i count (bit k: the k-th neighbour in the reference's walking order -- forward
// stop: j = i + 3 - k; reverse stop: j = i - 3 + k), and whether the walk ends among them (bit 31).  Positions, strands and
// types only: the same for every model scored on the contig, so a launch over (chain, stop) pairs reads it from OvlTopo.
constexpr int OV_SPEC = 16;
__device__ __forceinline__ unsigned overlap_neighbours(const int32_t* __restrict__ ndx, const int32_t* __restrict__ stv, const uint8_t* __restrict__ typ,
                                                       const int8_t* __restrict__ str, const int n, const int i, const int maxov) {
    const int my = ndx[i];
    const bool fwd = str[i] == 1;
    bool ended = false;
    unsigned elig = 0;
    // (the first OV_NEAR neighbours at once; the walk nearly always ends among them -- max_overlap is sixty bases, a node sits every
    //  dozen -- and the others are only asked for where it does not: nothing of a neighbour behind the end of the walk is ever used)
    constexpr int OV_NEAR = 10;
#pragma unroll
    for (int k = 0; k < OV_SPEC; k++) {
        if (k == OV_NEAR && ended) break;
        const int j = fwd ? i + 3 - k : i - 3 + k;
        const int jj = min(max(j, 0), n - 1);
        const int nd = ndx[jj], sv = stv[jj], ty = typ[jj], sd = str[jj];       // unconditional within either part: every load can be in flight at once
        bool stop_here, ok;
        if (fwd) {
            stop_here = j < 0 || (j < n && nd <= my + 2 && nd + maxov < my);
            ok = j >= 0 && j < n && nd <= my + 2 && sd == 1 && ty != PGA_T_STOP && sv > my;
        } else {
            stop_here = j >= n || (j >= 0 && nd >= my - 2 && nd - maxov > my);
            ok = j >= 0 && j < n && nd >= my - 2 && sd == -1 && ty != PGA_T_STOP && sv < my;
        }
        ended = ended || stop_here;
        if (ok && !ended) {
            elig |= 1u << k;
            // bits 16 .. 30: the pair is adjacent (the start's RBS / upstream scores enter its price, _connection.h:60-66) -- only a node within
            // two bases is, so never the sixteenth neighbour; who prices the pair knows which scores to ask for before it has the positions
            if (k < 15 && (fwd ? (my + 2 == nd || my == nd + 1) : (nd + 2 == my || nd == my + 1))) elig |= 0x10000u << k;
        }
    }
    return elig | (ended ? 0x80000000u : 0u);
}

__device__ __forceinline__ void overlapping_starts_of(const OvlChain& C, const int i, const ModelConst* __restrict__ mc, const int maxov,
                                                      int& sp0, int& sp1, int& sp2, const unsigned neighbours) {
    const int32_t* __restrict__ ndx = C.ndx; const int32_t* __restrict__ stv = C.stv;
    const uint8_t* __restrict__ typ = C.typ; const int8_t* __restrict__ str = C.str;
    const double* __restrict__ cs = C.cs; const double* __restrict__ ss = C.ss; const double* __restrict__ rs = C.rs; const double* __restrict__ us = C.us;
    const int n = C.n;
    const int my = ndx[i];
    double best = -100;
    const bool fwd = str[i] == 1;
    const double rs_i = 0.0, us_i = 0.0;        // (the stop's own RBS / upstream scores never enter: the start's do, when adjacent)
    // The reference walks the neighbours one by one (forward stop: j = i + 3 downwards; reverse stop: j = i - 3 upwards) until
    // it leaves the overlap window.  Which of the first OV_SPEC count is known (overlap_neighbours); only those are priced, in
    // the reference's order; a window that is not done by then (rare) goes on one by one.
    unsigned elig = neighbours & 0xffffu;
    int js = fwd ? i + 3 - OV_SPEC : i - 3 + OV_SPEC;        // where the one-by-one walk goes on if the window is not done by then
    bool more = !(neighbours >> 31);
    for (;;) {
        int j = -1;
        if (elig) {
            const int k = __builtin_ctz(elig);
            elig &= elig - 1u;
            j = fwd ? i + 3 - k : i - 3 + k;
        } else {
            while (more) {
                if (fwd ? js < 0 : js >= n) { more = false; break; }
                const int jq = js;
                js += fwd ? -1 : 1;
                if (jq < 0 || jq >= n) continue;
                const int nq = ndx[jq];
                if (fwd ? nq > my + 2 : nq < my - 2) continue;
                if (fwd ? nq + maxov < my : nq - maxov > my) { more = false; break; }
                if (str[jq] != (fwd ? 1 : -1) || typ[jq] == PGA_T_STOP) continue;
                if (fwd ? stv[jq] <= my : stv[jq] >= my) continue;
                j = jq;
                break;
            }
            if (j < 0) break;
        }
        const int nj = ndx[j];
        const double csj = C.css != nullptr ? C.css[j] : cs[j] + ss[j];
        // the RBS / upstream scores of the start only enter when the two nodes are adjacent (_connection.h:60-66): asked for then only
        const bool adj = fwd ? (my + 2 == nj || my == nj + 1) : (nj + 2 == my || nj == my + 1);
        const double rj = adj ? rs[j] : 0.0, uj = adj ? us[j] : 0.0;
        const double v = fwd ? csj + igm_same_dev(my, 1, rs_i, us_i, nj, rj, uj, mc->st_wt, mc->igm)
                             : csj + igm_same_dev(nj, -1, rj, uj, my, rs_i, us_i, mc->st_wt, mc->igm);
        if (v > best) { const int f = nj % 3; if (f == 0) sp0 = j; else if (f == 1) sp1 = j; else sp2 = j; best = v; }
    }
}

// the neighbours that count, once per stop node of a group (ga.ovl_topo, indexed like ga.stop_list)
__global__ void __launch_bounds__(256)
k_ovl_topo(GroupArrays ga, const int32_t* __restrict__ cbase, const int32_t* __restrict__ sbase, int n_contigs, int n_stops, int maxov) {
    __shared__ int s_c0;
    const int blk0 = blockIdx.x * blockDim.x, s = blk0 + threadIdx.x;
    int c = block_search_le([&](const int k) { return sbase[k]; }, n_contigs, blk0, &s_c0);
    if (s >= n_stops) return;
    while (c + 1 < n_contigs && sbase[c + 1] <= s) c++;
    const int b0 = cbase[c], n = cbase[c + 1] - b0;
    const int i = ga.stop_list[s] - b0;
    if (ga.srank != nullptr) ga.srank[b0 + i] = s - sbase[c];
    ga.ovl_topo[s] = ga.edge0[b0 + i] == 1 ? 0x80000000u
                                           : overlap_neighbours(ga.ndx + b0, ga.stop_val + b0, ga.type + b0, ga.strand + b0, n, i, maxov);
}

// one thread per chain node
__global__ void __launch_bounds__(256)
k_overlapping_starts(const ChainDesc* __restrict__ chains, int n_chains, int64_t node_begin, int64_t total,
                     GroupArrays ga, const ModelConst* __restrict__ mcs, ChainArrays ca, int maxov) {
    __shared__ int s_c0;
    const int64_t blk0 = node_begin + (int64_t)blockIdx.x * blockDim.x;
    const int64_t g = blk0 + threadIdx.x;
    const bool in_range = g < node_begin + total;
    const int c = block_chain(chains, n_chains, blk0, in_range ? g : blk0, &s_c0);
    if (!in_range) return;
    const ChainDesc ch = chains[c];
    const int i = (int)(g - ch.off);
    const int64_t tb = ch.topo_off;
    int sp0 = -1, sp1 = -1, sp2 = -1;
    if (ga.type[tb + i] == PGA_T_STOP && ga.edge0[tb + i] != 1) {
        const OvlChain C{ga.ndx + tb, ga.stop_val + tb, ga.type + tb, ga.strand + tb, ca.cscore + ch.off, ca.sscore + ch.off, ca.rscore + ch.off,
                         ca.uscore + ch.off, ch.n};
        overlapping_starts_of(C, i, &mcs[ch.model], maxov, sp0, sp1, sp2, overlap_neighbours(C.ndx, C.stv, C.typ, C.str, C.n, i, maxov));
    }
    ca.star_ptr[3 * g] = sp0; ca.star_ptr[3 * g + 1] = sp1; ca.star_ptr[3 * g + 2] = sp2;
}

// overlapping_starts_of for a kernel that does nothing but wait for memory: the first four neighbours that count are asked for AT ONCE
// (position, cscore + sscore; RBS and upstream score where k_ovl_topo found the pair adjacent; for the extras of a reverse stop the
// start's stop_val and first candidate as well) and priced in the reference's order afterwards; what is left of the window (a fifth
// neighbour: rare) goes on one by one as above.  Same values, same comparisons.  What the extras record needs of the three starts
// that are kept -- the price (cs + intergenic term: exactly DpwExt::x), position, stop_val, first candidate -- stays in registers.
constexpr int OV_AHEAD = 4;
struct OvlKept { double v[3]; int nd[3], sv[3], q2[3]; };
__device__ __forceinline__ void overlapping_starts_ahead(const OvlChain& C, const int i, const int my, const bool fwd, const ModelConst* __restrict__ mc,
                                                         const int maxov, int (&sp)[3], OvlKept& K, const unsigned neighbours,
                                                         const int32_t* __restrict__ q2 /* topology: first candidate of a reverse start, or nullptr (no extras wanted) */) {
    const int32_t* __restrict__ ndx = C.ndx; const int32_t* __restrict__ stv = C.stv;
    const uint8_t* __restrict__ typ = C.typ; const int8_t* __restrict__ str = C.str;
    const double* __restrict__ cs = C.cs; const double* __restrict__ ss = C.ss; const double* __restrict__ rs = C.rs; const double* __restrict__ us = C.us;
    const int n = C.n;
    const bool rext = !fwd && q2 != nullptr;                 // the extras of a reverse stop: stop_val and first candidate of the starts as well
    double best = -100;
    unsigned elig = neighbours & 0xffffu;
    const unsigned adjb = (neighbours >> 16) & 0x7fffu;
    int jn[OV_AHEAD], nq[OV_AHEAD], svq[OV_AHEAD], q2q[OV_AHEAD]; double cq[OV_AHEAD], rq[OV_AHEAD], uq[OV_AHEAD]; bool aq[OV_AHEAD];
#pragma unroll
    for (int q = 0; q < OV_AHEAD; q++) {
        jn[q] = -1; aq[q] = false;
        if (elig) { const int k = __builtin_ctz(elig); elig &= elig - 1u; jn[q] = fwd ? i + 3 - k : i - 3 + k; aq[q] = (adjb >> k) & 1u; }
    }
#pragma unroll
    for (int q = 0; q < OV_AHEAD; q++) {
        const int jj = jn[q] >= 0 ? jn[q] : i;              // (no neighbour: the stop node itself, read and dropped)
        nq[q] = ndx[jj]; cq[q] = C.css != nullptr ? C.css[jj] : cs[jj] + ss[jj];
        rq[q] = aq[q] ? rs[jj] : 0.0; uq[q] = aq[q] ? us[jj] : 0.0;          // (adjacent pairs are rare: k_ovl_topo says which, so nobody else reads these lines)
        svq[q] = rext ? stv[jj] : 0; q2q[q] = rext ? q2[jj] : 0;
    }
    auto price = [&](const int j, const int nj, const double csj, const double rj0, const double uj0, const int svj, const int q2j) {
        // the RBS / upstream scores of the start only enter when the two nodes are adjacent (_connection.h:60-66)
        const bool adj = fwd ? (my + 2 == nj || my == nj + 1) : (nj + 2 == my || nj == my + 1);
        const double rj = adj ? rj0 : 0.0, uj = adj ? uj0 : 0.0;
        const double v = fwd ? csj + igm_same_dev(my, 1, 0.0, 0.0, nj, rj, uj, mc->st_wt, mc->igm)
                             : csj + igm_same_dev(nj, -1, rj, uj, my, 0.0, 0.0, mc->st_wt, mc->igm);
        if (v > best) {
            const int f = nj % 3; best = v;
            if (f == 0) { sp[0] = j; K.v[0] = v; K.nd[0] = nj; K.sv[0] = svj; K.q2[0] = q2j; }
            else if (f == 1) { sp[1] = j; K.v[1] = v; K.nd[1] = nj; K.sv[1] = svj; K.q2[1] = q2j; }
            else { sp[2] = j; K.v[2] = v; K.nd[2] = nj; K.sv[2] = svj; K.q2[2] = q2j; }
        }
    };
#pragma unroll
    for (int q = 0; q < OV_AHEAD; q++) if (jn[q] >= 0) price(jn[q], nq[q], cq[q], rq[q], uq[q], svq[q], q2q[q]);
    bool more = !(neighbours >> 31);
    if (!elig && !more) return;
    int js = fwd ? i + 3 - OV_SPEC : i - 3 + OV_SPEC;        // where the one-by-one walk goes on if the window is not done by then
    for (;;) {
        int j = -1;
        if (elig) {
            const int k = __builtin_ctz(elig);
            elig &= elig - 1u;
            j = fwd ? i + 3 - k : i - 3 + k;
        } else {
            while (more) {
                if (fwd ? js < 0 : js >= n) { more = false; break; }
                const int jq = js;
                js += fwd ? -1 : 1;
                if (jq < 0 || jq >= n) continue;
                const int nqq = ndx[jq];
                if (fwd ? nqq > my + 2 : nqq < my - 2) continue;
                if (fwd ? nqq + maxov < my : nqq - maxov > my) { more = false; break; }
                if (str[jq] != (fwd ? 1 : -1) || typ[jq] == PGA_T_STOP) continue;
                if (fwd ? stv[jq] <= my : stv[jq] >= my) continue;
                j = jq;
                break;
            }
            if (j < 0) break;
        }
        price(j, ndx[j], C.css != nullptr ? C.css[j] : cs[j] + ss[j], rs[j], us[j], rext ? stv[j] : 0, rext ? q2[j] : 0);
    }
}

// One thread per (chain, stop node) pair: stop nodes are one node in five, and a wavefront of the kernel above waits for its
// few stop lanes.  The pairs of a chain are ChainDesc::soff .. ; the k-th stop of a contig is ga.stop_list[sbase[contig] + k].
// star_ptr of the other nodes is -1 (the launcher fills the range first).  With `ext` the 64-byte extras record of the
// wave-batch connection scorer is built from the three starts while they are at hand (dpw_core.h, dpw_chain_ext_sp: the same
// record from the same values -- a kept start's price IS the record's x, its position and stop_val give the candidate interval).
// The kernel waits for memory and nothing else -- 0.10 of the vector pipe, eight wavefronts per SIMD, its time the depth of its chain
// of dependent loads -- so every load is asked for as soon as its address is known: the chain of the pair (three descriptors ahead
// instead of a walk), the contig's bases, the stop node and its neighbour mask, then the node's own fields TOGETHER with those of
// its first four neighbours, and nothing at all for the extras (round 6, fourth session: fourteen round trips became five).
__global__ void __launch_bounds__(256)
k_ovl_stops(const ChainDesc* __restrict__ chains, int n_chains, int64_t soff_begin, int64_t n_pairs, GroupArrays ga,
            const int32_t* __restrict__ cbase, const int32_t* __restrict__ sbase, const ModelConst* __restrict__ mcs, ChainArrays ca, int maxov,
            const int32_t* __restrict__ topo_q2, DpwExt* __restrict__ ext, const double* __restrict__ css /* or nullptr: cscore + sscore per chain node */,
            const int32_t* __restrict__ blk_chain /* or nullptr: the chain of every workgroup's first pair (StopLaunch::blk_chain) */) {
    __shared__ int s_c0;
    const int64_t blk0 = soff_begin + (int64_t)blockIdx.x * blockDim.x;
    const int64_t p = blk0 + threadIdx.x;
    int c;
    if (blk_chain != nullptr) c = blk_chain[blockIdx.x];
    else c = block_search_le([&](const int k) { return chains[k].soff; }, n_chains, blk0, &s_c0);
    if (p >= soff_begin + n_pairs) return;
    {   // a workgroup's 256 pairs rarely reach beyond the third chain after its first
        const int64_t s1 = chains[min(c + 1, n_chains - 1)].soff, s2 = chains[min(c + 2, n_chains - 1)].soff, s3 = chains[min(c + 3, n_chains - 1)].soff;
        const int c_in = c;
        c += (c_in + 1 < n_chains && s1 <= p) + (c_in + 2 < n_chains && s2 <= p) + (c_in + 3 < n_chains && s3 <= p);
    }
    while (c + 1 < n_chains && chains[c + 1].soff <= p) c++;
    const ChainDesc ch = chains[c];
    const int64_t tb = ch.topo_off;
    const int sb = sbase[ch.contig], cb = cbase[ch.contig];
    const int sidx = sb + (int)(p - ch.soff);
    const int i = ga.stop_list[sidx] - cb;
    const unsigned neighbours = ga.ovl_topo[sidx];            // (an edge stop: nothing counts, k_ovl_topo)
    const int64_t g = ch.off + i;
    const ModelConst* __restrict__ mc = &mcs[ch.model];
    const int edge0 = ga.edge0[tb + i], my = ga.ndx[tb + i], my_str = ga.strand[tb + i];
    const bool rev = my_str != 1;
    int sp[3] = {-1, -1, -1};
    OvlKept K{{0.0, 0.0, 0.0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    const OvlChain C{ga.ndx + tb, ga.stop_val + tb, ga.type + tb, ga.strand + tb, ca.cscore + ch.off, ca.sscore + ch.off, ca.rscore + ch.off,
                     ca.uscore + ch.off, ch.n, css != nullptr ? css + ch.off : nullptr};
    overlapping_starts_ahead(C, i, my, !rev, mc, maxov, sp, K, edge0 != 1 ? neighbours : 0x80000000u, ext != nullptr ? topo_q2 + tb : nullptr);
    ca.star_ptr[3 * g] = sp[0]; ca.star_ptr[3 * g + 1] = sp[1]; ca.star_ptr[3 * g + 2] = sp[2];      // (an edge stop: -1, never what an earlier call left there)
    if (ext != nullptr) {
        // dpw_chain_ext_sp (dpw_core.h) from what the search kept: a start that counts lies on the stop's own strand, so its price
        // cs + igm IS x[k] (the same additions in the same order), and a reverse start carries its own first candidate
        DpwExt e;
        e.vm = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            e.x[k] = 0.0; e.dlo[k] = INT_MAX; e.dhi[k] = INT_MIN; e.cq[k] = DPW_NONE;
            if (sp[k] < 0) continue;
            e.vm |= 1 << k;
            e.x[k] = K.v[k];
            if (rev && e.x[k] > 0.0) {
                const int n3n = K.nd[k], n3s = K.sv[k];
                int hi = n3s + DPW_MAX_OPP_OVLP - 5;
                const int h2 = (n3n + n3s - 6) >> 1;
                if (h2 < hi) hi = h2;
                if (my - 4 < hi) hi = my - 4;
                e.dlo[k] = n3s - 5; e.dhi[k] = hi;
            }
            if (rev) e.cq[k] = K.q2[k];
        }
        ext[p] = e;             // dense: one record per (chain, stop node) pair, in pair order
    }
}

}  // namespace

// ------------------------------------------------------------------------------ launchers
static inline unsigned nblocks(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

int64_t pga_gc_blocks(int64_t total) { return (total + 1 + 4095) / 4096; }

void pga_launch_digitize(const char* d_seq, uint8_t* d_dig, int64_t total, const ContigDesc* d_ct, int n_contigs,
                         int32_t* d_gc, int32_t* d_unk, hipStream_t st) {
    if (total <= 0) return;
    hipLaunchKernelGGL(k_digitize, dim3(nblocks(total, DG_BLOCK)), dim3(256), 0, st, d_seq, d_dig, total, d_ct, n_contigs, d_gc, d_unk);
}

// Runs of unknown bases (ref: lib.pyx:699-713, Sequence._mask): the thread that sees the first N of a run walks to
// its end, 8 bytes at a time, and records it when it is long enough.  Runs are rare and mostly short.
__global__ void __launch_bounds__(256)
k_find_masks(const uint8_t* __restrict__ dig, const ContigDesc* __restrict__ ct, const TileDesc* __restrict__ tiles, int n_tiles, int n_contigs,
             int min_mask, MaskRun* __restrict__ runs, int32_t* __restrict__ count, int cap) {
    if ((int)blockIdx.x >= n_tiles) {
        // sequences of one or two bases have no extraction tile: one thread each
        const int c = ((int)blockIdx.x - n_tiles) * blockDim.x + threadIdx.x;
        if (c >= n_contigs) return;
        const ContigDesc cd = ct[c];
        if (cd.len < 1 || cd.len > 2) return;
        const uint8_t* __restrict__ d = dig + cd.base;
        for (int i = 0; i < cd.len; i++) {
            if (d[i] != NN || (i > 0 && d[i - 1] == NN)) continue;
            int e = i + 1;
            while (e < cd.len && d[e] == NN) e++;
            if (e - i >= min_mask || e == cd.len) { const int k = atomicAdd(count, 1); if (k < cap) runs[k] = MaskRun{c, i, e, 0}; }
        }
        return;
    }
    const TileDesc td = tiles[blockIdx.x];
    const ContigDesc cd = ct[td.contig];
    const int L = cd.len;
    const uint8_t* __restrict__ d = dig + cd.base;
    const int i0 = td.start + threadIdx.x * EX_PER_THREAD;
    for (int q = 0; q < EX_PER_THREAD; q++) {
        const int i = i0 + q;
        if (i >= L || d[i] != NN || (i > 0 && d[i - 1] == NN)) continue;
        int e = i + 1;
        while (e + 8 <= L) {
            uint64_t w; memcpy(&w, d + e, 8);
            if (w != 0x0606060606060606ull) break;
            e += 8;
        }
        while (e < L && d[e] == NN) e++;
        if (e - i >= min_mask || e == L) {          // a run that reaches the end of the sequence is masked whatever its length (ref: lib.pyx:711-712)
            const int k = atomicAdd(count, 1);
            if (k < cap) runs[k] = MaskRun{td.contig, i, e, 0};
        }
    }
}

void pga_launch_find_masks(const uint8_t* d_dig, const ContigDesc* d_ct, int n_contigs, const TileDesc* d_tiles, int n_tiles, int min_mask,
                           MaskRun* d_runs, int32_t* d_count, int cap, hipStream_t st) {
    (void)hipMemsetAsync(d_count, 0, sizeof(int32_t), st);
    hipLaunchKernelGGL(k_find_masks, dim3(n_tiles + (n_contigs + 255) / 256), dim3(256), 0, st, d_dig, d_ct, d_tiles, n_tiles, n_contigs, min_mask,
                       d_runs, d_count, cap);
}

void pga_launch_gc_prefix(const uint8_t* d_dig, int64_t total, int32_t* d_block_sum, int32_t* d_block_off, int32_t* d_p16, hipStream_t st) {
    if (total <= 0) return;
    const int nb = (int)pga_gc_blocks(total);
    hipLaunchKernelGGL(k_gcp_blocks, dim3(nb), dim3(256), 0, st, d_dig, total, d_block_sum);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, d_block_sum, nb, d_block_off, (const int32_t*)nullptr, (int32_t*)nullptr);
    hipLaunchKernelGGL(k_gcp_final, dim3(nb), dim3(256), 0, st, d_dig, total, d_block_off, d_p16);
}

void pga_launch_extract(const uint8_t* d_dig, int64_t total, const ContigDesc* d_ct, int n_contigs, int tt,
                        const pga_params& p, const GroupArrays& ga, const TileDesc* d_tiles, int n_tiles, const int32_t* d_tile0,
                        int32_t* d_tile_first, int32_t* d_tile_last, int32_t* d_tile_count, int32_t* d_tile_off, int32_t* d_cbase,
                        int32_t* d_tile_scount, int32_t* d_tile_soff, int32_t* d_sbase, MaskList masks, hipStream_t st, const uint8_t* d_enabled) {
    if (n_tiles > 0) {
        ExParams P{tt, p.closed, p.min_gene, p.min_edge_gene, 0ull, 0ull};
        for (int idx = 0; idx < 64; idx++) {
            const int b0 = idx & 3, b1 = (idx >> 2) & 3, b2 = (idx >> 4) & 3;
            if (codon_is_stop(b0, b1, b2, tt)) P.stop_codons |= 1ull << idx;
            if (codon_is_start(b0, b1, b2, tt)) P.start_codons |= 1ull << idx;
        }
        hipLaunchKernelGGL(k_tile_stops, dim3(n_tiles), dim3(64), 0, st, d_dig, total, d_ct, d_tiles, n_tiles, P.stop_codons, d_tile_first, d_tile_last, d_enabled);
        hipLaunchKernelGGL(k_extract_tile, dim3(n_tiles), dim3(256), 0, st, d_dig, total, d_ct, d_tiles, n_tiles, d_tile_first, d_tile_last, P, ga, masks,
                           d_enabled, d_tile_count, d_tile_scount);
    }
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, d_tile_count, n_tiles, d_tile_off, (const int32_t*)d_tile_scount, d_tile_soff);
    hipLaunchKernelGGL(k_contig_node_base, dim3((n_contigs + 1 + 255) / 256), dim3(256), 0, st, d_tile0, n_contigs, d_tile_off, d_cbase);
    hipLaunchKernelGGL(k_contig_node_base, dim3((n_contigs + 1 + 255) / 256), dim3(256), 0, st, d_tile0, n_contigs, d_tile_soff, d_sbase);
}

void pga_launch_place(const ContigDesc* d_ct, const TileDesc* d_tiles, int n_tiles, const int32_t* d_tile_off, const int32_t* d_tile_soff,
                      const GroupArrays& ga, hipStream_t st) {
    if (n_tiles <= 0) return;
    hipLaunchKernelGGL(k_place_nodes, dim3(n_tiles), dim3(128), 0, st, d_ct, d_tiles, d_tile_off, d_tile_soff, ga);
}

int pga_extract_tile_size() { return EX_TILE; }

// Tasks of k_coding_score_quads for one translation-table group: contigs whose models [m0, m0 + 4) are the same four columns
// of the group's interleaved table, in runs of about `task_nodes` nodes.  Returns false (and leaves the outputs empty) when
// some contig's models are not neighbours in the table: the caller then takes the global-memory form.
bool pga_cs_tasks(const int2* h_cc /* per contig: first chain, count */, int n_contigs, const ChainDesc* h_chains, const int32_t* h_cbase,
                  const int32_t* model_rank, int task_nodes, std::vector<int32_t>& tasks /* 4 per task */, std::vector<int32_t>& entries /* 4 per entry */) {
    tasks.clear(); entries.clear();
    std::vector<std::vector<int32_t>> bucket(64);
    for (int i = 0; i < n_contigs; i++) {
        const int2 cc = h_cc[i];
        if (cc.y <= 0 || h_cbase[i + 1] == h_cbase[i]) continue;
        const int r0 = model_rank[h_chains[cc.x].model];
        for (int m = 1; m < cc.y; m++) if (model_rank[h_chains[cc.x + m].model] != r0 + m) return false;
        for (int m0 = 0; m0 < cc.y; m0 += 4) {
            if (r0 + m0 >= 64) return false;
            bucket[(size_t)(r0 + m0)].push_back(i); bucket[(size_t)(r0 + m0)].push_back(m0);
        }
    }
    // columns of high-GC models first: their contigs have the longest ORFs, and a launch ends when its last task does
    for (int q = 63; q >= 0; q--) {
        const std::vector<int32_t>& b = bucket[(size_t)q];
        int first = (int)(entries.size() / 4), count = 0, nodes = 0, cuts = 0;
        auto flush = [&]() {
            if (count > 0) { tasks.push_back(q); tasks.push_back(first); tasks.push_back(count); tasks.push_back(0); }
            first += count; count = 0; nodes = 0; cuts = 0;
        };
        for (size_t k = 0; k < b.size(); k += 2) {
            // a task is one round of the kernel (at most task_nodes nodes): a contig with more is cut into pieces, a genome becomes
            // hundreds of tasks instead of one
            const int nc = h_cbase[b[k] + 1] - h_cbase[b[k]];
            for (int f0 = 0; f0 < nc; f0 += task_nodes) {
                const int piece = std::min(task_nodes, nc - f0);
                // (a piece of a contig may list a few stop nodes more than half its nodes: see CS_LIST)
                if (count > 0 && (nodes + piece > task_nodes || count == CS_TASK_MAX_ENTRIES || (piece != nc && cuts == CS_TASK_MAX_CUTS))) flush();
                entries.push_back(b[k]); entries.push_back(b[k + 1]); entries.push_back(f0); entries.push_back(piece);
                count++; nodes += piece; cuts += piece != nc;
            }
        }
        flush();
    }
    return true;
}

// Which translation-table groups a contig needs at all: a group whose models all lie outside the contig's GC window
// (ref: lib.pyx:5335-5336, the same two expressions the host evaluates when it plans the chains) is not extracted for it.
__global__ void __launch_bounds__(256)
k_group_enable(const ContigDesc* __restrict__ ct, int n_contigs, const int32_t* __restrict__ gc_count, const double* __restrict__ model_gc,
               const int32_t* __restrict__ model_group, int n_models, int n_groups, uint8_t* __restrict__ enabled /* [group][contig] */) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_contigs) return;
    const int L = ct[i].len;
    const double gc = L > 0 ? (double)gc_count[i] / (double)L : 0.0;
    const double low = fmin(0.65, 0.88495 * gc - 0.0102337), high = fmax(0.35, 0.86596 * gc + 0.1131991);
    unsigned need = 0;
    for (int m = 0; m < n_models; m++) if (!(model_gc[m] < low || model_gc[m] > high)) need |= 1u << model_group[m];
    for (int g = 0; g < n_groups; g++) enabled[(size_t)g * n_contigs + i] = (need >> g) & 1;
}
void pga_launch_group_enable(const ContigDesc* d_ct, int n_contigs, const int32_t* d_gc_count, const double* d_model_gc,
                             const int32_t* d_model_group, int n_models, int n_groups, uint8_t* d_enabled, hipStream_t st) {
    if (n_contigs <= 0) return;
    hipLaunchKernelGGL(k_group_enable, dim3(nblocks(n_contigs, 256)), dim3(256), 0, st, d_ct, n_contigs, d_gc_count, d_model_gc, d_model_group,
                       n_models, n_groups, d_enabled);
}



void pga_launch_orf_gc(const ContigDesc* d_ct, int n_contigs, const uint8_t* d_dig, const int32_t* d_p16, const GroupArrays& ga,
                       int n_nodes_total, const int32_t* d_node_contig_base, hipStream_t st) {
    if (n_nodes_total <= 0) return;
    hipLaunchKernelGGL(k_orf_gc, dim3(nblocks(n_nodes_total, 256)), dim3(256), 0, st, d_ct, n_contigs, d_dig, d_p16, ga,
                       n_nodes_total, d_node_contig_base);
}

void pga_launch_sd_lut(unsigned* d_lut, hipStream_t st) {
    hipLaunchKernelGGL(k_sd_lut, dim3((PGA_SD_LUT + 255) / 256), dim3(256), 0, st, d_lut);
}

void pga_launch_score(const ChainDesc* d_chains, int n_chains, int64_t node_begin, int64_t total, const uint8_t* d_dig,
                      const ContigDesc* d_ct, const GroupArrays& ga, const pga_training* d_models,
                      const ModelScoreConst* d_msc, const ModelConst* d_mc, const ChainArrays& ca, ScoreParams sp,
                      const ChainDesc* d_all_chains, const int2* d_contig_chains, const int32_t* d_node_contig_base, int n_contigs,
                      int group_nodes, const unsigned* d_sd_lut, hipStream_t st, int reuse_raw_cscore, const double* d_gil, int il_stride,
                      const int32_t* d_rank, const void* d_cs_tasks, int n_cs_tasks, const void* d_cs_entries, const StopLaunch* stops) {
    if (total <= 0 || n_chains <= 0) return;
    const dim3 grid(nblocks(total, 256)), blk(256);
    if (group_nodes > 0 && !reuse_raw_cscore && n_cs_tasks > 0) {
        // the ORF walks against hexamer tables in LDS (tasks built by the caller, pga_cs_tasks)
        // function attributes and allocations belong to ONE device: both are keyed by the device that is current
        // (contexts of several GPUs may live in one process)
        static std::atomic<bool> attr_set[64];
        static std::atomic<unsigned long long*> prof_of[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        dev &= 63;
        const size_t lds = sizeof(double) * 4096 * 4;
        int cs_wave = CS_WAVE;
        if (const char* e = getenv("PGA_CS_WAVE")) cs_wave = atoi(e) >= 64 ? atoi(e) : CS_WAVE;
        if (!attr_set[dev].load()) {
            hipFuncSetAttribute((const void*)k_coding_score_quads, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set[dev].store(true);
        }
        const bool profiling = getenv("PGA_CS_PROFILE") != nullptr;
        unsigned long long* d_prof = prof_of[dev].load();
        if (profiling) {
            if (!d_prof) { (void)hipMalloc((void**)&d_prof, 16 * sizeof(unsigned long long)); prof_of[dev].store(d_prof); }
            (void)hipMemsetAsync(d_prof, 0, 16 * sizeof(unsigned long long), st);
        }
        hipLaunchKernelGGL(k_coding_score_quads, dim3((unsigned)n_cs_tasks), dim3(CS_TASK_THREADS), lds, st, (const CsTask*)d_cs_tasks,
                           (const CsEntry*)d_cs_entries, d_all_chains, d_contig_chains, d_node_contig_base, d_dig, d_ct, ga, d_models, d_msc, ca,
                           d_gil, il_stride, d_rank, cs_wave, profiling ? d_prof : nullptr, (getenv("PGA_CS_QCLASSES") && atoi(getenv("PGA_CS_QCLASSES")) == 4) ? 4 : 3);
        if (profiling) {
            unsigned long long h[16];
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(h, d_prof, sizeof h, hipMemcpyDeviceToHost);
            fprintf(stderr, "[pga cs-profile] %d tasks (%llu workgroups, %llu batches of 64): wave-cycles  stage %.3g  list %.3g  one-wave ORFs %.3g  "
                            "four-to-a-wave %.3g  64-to-a-wave %.3g (lookup %.3g  first loads %.3g  walk %.3g  penalties %.3g)  end barrier %.3g\n",
                    n_cs_tasks, h[7], h[8], (double)h[0], (double)h[1], (double)h[2], (double)h[3], (double)h[4], (double)h[9], (double)h[10], (double)h[11],
                    (double)h[12], (double)h[5]);
        }
        reuse_raw_cscore = 1;           // done: skip the global-memory form below
    }
    // the raw coding score of a start depends on the model only, not on which model of the group was scored first on the
    // contig: the fresh re-score of a winning model (ref: lib.pyx:5380-5394) reads it where the winning pass left it
    // (ChainDesc::raw_off) instead of walking every ORF again
    if (group_nodes > 0 && !reuse_raw_cscore)
        hipLaunchKernelGGL(k_coding_score, dim3(nblocks(group_nodes, 256)), blk, 0, st, d_all_chains, d_contig_chains, d_node_contig_base,
                           n_contigs, 0, group_nodes, d_dig, d_ct, ga, d_models, d_msc, ca, d_gil, il_stride, d_rank);
    if (group_nodes > 0) {
        if (const char* e = getenv("PGA_SS_MODELS_PER_PASS")) { const int v = atoi(e); if (v >= 1 && v <= 512) sp.models_per_pass = v; }
        // PGA_SS_PROFILE=1: wave-cycles per phase of the start scorer (a synchronising debug aid)
        static std::atomic<unsigned long long*> ss_prof_of[64];
        const bool ss_profiling = getenv("PGA_SS_PROFILE") != nullptr;
        if (ss_profiling) {
            int dev = 0; (void)hipGetDevice(&dev); dev &= 63;
            unsigned long long* d_prof = ss_prof_of[dev].load();
            if (!d_prof) { (void)hipMalloc((void**)&d_prof, 16 * sizeof(unsigned long long)); ss_prof_of[dev].store(d_prof); }
            (void)hipMemsetAsync(d_prof, 0, 16 * sizeof(unsigned long long), st);
            sp.prof = d_prof;
        }
        // (a fifth wavefront per SIMD costs 96 bytes of scratch per lane and 10 % of the kernel's time: four)
        const bool listed = stops != nullptr && stops->starts_only && ga.start_list != nullptr;
        const int n_items = listed ? stops->n_starts : group_nodes;
        if (n_items > 0)
            hipLaunchKernelGGL(k_score_starts<4>, dim3(nblocks(n_items, 256)), blk, 0, st, d_all_chains, d_contig_chains, d_node_contig_base, n_contigs,
                               group_nodes, d_dig, d_ct, ga, d_models, ca, sp, d_sd_lut, listed ? (const int32_t*)ga.start_list : (const int32_t*)nullptr, n_items);
        if (ss_profiling) {
            unsigned long long h[16];
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(h, sp.prof, sizeof h, hipMemcpyDeviceToHost);
            fprintf(stderr, "[pga ss-profile] %d nodes: wave-cycles  locate %.3g  node + upstream window %.3g  barrier + edge scan %.3g | per model: next model %.3g  "
                            "chain %.3g  upstream + RBS / motif search %.3g  scores %.3g  finish + stores (slot 8) %.3g  stage %.3g  barrier %.3g (slot 7 unused %.3g)\n",
                    group_nodes, (double)h[0], (double)h[1], (double)h[2], (double)h[3], (double)h[4], (double)h[5], (double)h[6], (double)h[8],
                    (double)h[9], (double)h[10], (double)h[7]);
        }
    }
    if (stops != nullptr) {
        if (stops->fill_star_ptr) (void)hipMemsetAsync(ca.star_ptr + 3 * node_begin, 0xff, sizeof(int32_t) * 3 * (size_t)total, st);
        if (stops->n_pairs > 0 && stops->n_stops > 0)
            hipLaunchKernelGGL(k_ovl_topo, dim3(nblocks(stops->n_stops, 256)), blk, 0, st, ga, d_node_contig_base, stops->sbase, n_contigs, stops->n_stops,
                               sp.max_overlap);
        if (stops->n_pairs > 0)
            hipLaunchKernelGGL(k_ovl_stops, dim3(nblocks(stops->n_pairs, 256)), blk, 0, st, d_chains, n_chains, stops->soff_begin, stops->n_pairs, ga,
                               d_node_contig_base, stops->sbase, d_mc, ca, sp.max_overlap, stops->topo_q2, (DpwExt*)stops->ext,
                               stops->ext != nullptr ? (const double*)sp.cs_out : nullptr, getenv("PGA_OVL_SEARCH") ? nullptr : stops->blk_chain);
    } else hipLaunchKernelGGL(k_overlapping_starts, grid, blk, 0, st, d_chains, n_chains, node_begin, total, ga, d_mc, ca, sp.max_overlap);
}
