// FASTA ingest for the finder: multi-record, straight into batches of contigs.
//
// Restates what the reference feeds to GeneFinder.find_genes (ref: src/pyrodigal/tests/fasta.py:59-86 `parse`,
// src/pyrodigal/cli.py:32-61): a record starts at a line beginning with '>', its id is the first
// whitespace-delimited word of that line and its description the rest; every other non-blank line, stripped
// of surrounding whitespace, is appended to the sequence; text before the first header is dropped, and a
// file with sequence text but no header at all is not FASTA.  Records are handed out in batches bounded by a
// base budget so that a caller can keep one batch on the device while the next one is being read.
//
// Two sources:
//   * a plain file is mapped and parsed by several threads: a batch is a window of the file that ends at a record boundary, cut
//     into pieces at record boundaries; every thread first measures its piece (records, bases per record), then -- offsets
//     known -- copies the stripped sequence lines straight to their place in the batch's arena (pinned host memory in packed
//     mode: the finder uploads from there with one DMA).  No per-record buffers, no second copy;
//   * a compressed stream (gzip through zlib; anything else through a caller-supplied read function, pga_fasta_open_callback:
//     the Python layer hands in bz2 / xz decompressors, ref: tests/fasta.py:16-57 `zopen`) is read line by line on one thread.
#include <hip/hip_runtime.h>
#include <zlib.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pyrodigal_amd.h"

namespace {

struct Rec { const char* hdr; uint32_t hdr_len; int64_t bases; };      // one record of a mapped piece
struct Piece {
    const char* b; const char* e;        // [b, e): starts at a header line (or at the batch start)
    std::vector<Rec> recs;               // records that START in the piece
    int64_t head_bases = 0;              // sequence text before the piece's first header: belongs to the last record before it
    bool head_text = false;
};

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

}  // namespace

struct pga_fasta {
    // stream source
    gzFile gz = nullptr;
    pga_fasta_read_fn cb = nullptr; void* cb_user = nullptr;
    std::string err;
    std::vector<char> buf;               // read window
    size_t pos = 0, end = 0;
    bool eof = false;
    bool in_record = false, seen_header = false, seen_text = false;
    std::string cur_hdr, cur_seq;
    bool pending = false;                // cur_* holds a finished record that did not fit the previous batch
    // mapped source
    const char* map = nullptr; size_t map_len = 0, map_pos = 0; int fd = -1;
    int threads = 1;
    // the batch handed out last
    std::vector<std::string> hdrs, seqs;
    std::vector<const char*> p_hdr, p_seq;
    std::vector<int64_t> lens, offs;
    std::vector<char> plain_arena;       // mapped source, unpacked calls: the stripped sequences back to back
    // packed mode: staging arenas in pinned host memory, filled in turn
    struct Arena { char* p = nullptr; size_t cap = 0; };
    std::vector<Arena> arenas;
    size_t next_arena = 0;
};

// Pinning memory costs about as much as parsing into it (page by page, a few GB/s): the arenas of a closed reader go to a small
// process-wide pool and the next reader takes them from there, so that a caller working through file after file pins once.
// The pool is bounded by bytes (SPARE_MAX_BYTES of pinned memory at most wait there), its arenas are allocated portable -- pinned
// for every device, whichever was current when a reader made them -- and pga_fasta_release_spare() gives them back.
static std::mutex g_spare_mu;
static std::vector<pga_fasta::Arena> g_spare;
static size_t g_spare_bytes = 0;
static const size_t SPARE_MAX = 8;
static const size_t SPARE_MAX_BYTES = (size_t)512 << 20;

// (g_spare_mu held) an arena nobody uses goes to the pool, or back to the system when the pool is full
static void spare_put(const pga_fasta::Arena& a) {
    if (!a.p) return;
    if (g_spare.size() < SPARE_MAX && g_spare_bytes + a.cap <= SPARE_MAX_BYTES) { g_spare.push_back(a); g_spare_bytes += a.cap; }
    else hipHostFree(a.p);
}

extern "C" void pga_fasta_release_spare(void) {
    std::lock_guard<std::mutex> g(g_spare_mu);
    for (auto& a : g_spare) hipHostFree(a.p);
    g_spare.clear();
    g_spare_bytes = 0;
}

// ---- stream source ------------------------------------------------------------------------------------------------------
static bool fill(pga_fasta* f) {
    if (f->eof) return false;
    if (f->pos > 0) { memmove(f->buf.data(), f->buf.data() + f->pos, f->end - f->pos); f->end -= f->pos; f->pos = 0; }
    if (f->end == f->buf.size()) f->buf.resize(f->buf.size() * 2);
    int64_t n;
    if (f->cb) {
        n = f->cb(f->cb_user, f->buf.data() + f->end, (int64_t)(f->buf.size() - f->end));
        if (n < 0) { f->err = "the read function of the stream failed"; f->eof = true; return false; }
    } else {
        n = gzread(f->gz, f->buf.data() + f->end, (unsigned)(f->buf.size() - f->end));
        if (n < 0) { int e; f->err = gzerror(f->gz, &e); f->eof = true; return false; }
    }
    if (n == 0) { f->eof = true; return false; }
    f->end += (size_t)n;
    return true;
}

// next line without its terminator; false at end of input
static bool next_line(pga_fasta* f, const char** line, size_t* len) {
    for (;;) {
        const char* b = f->buf.data() + f->pos;
        const char* nl = (const char*)memchr(b, '\n', f->end - f->pos);
        if (nl) { *line = b; *len = (size_t)(nl - b); f->pos += *len + 1; return true; }
        if (!fill(f)) {
            if (f->pos < f->end) { *line = f->buf.data() + f->pos; *len = f->end - f->pos; f->pos = f->end; return true; }
            return false;
        }
    }
}

static int next_stream(pga_fasta* f, int64_t max_bases, int32_t max_records) {
    int64_t bases = 0;
    auto emit = [&]() {
        bases += (int64_t)f->cur_seq.size();
        f->hdrs.push_back(std::move(f->cur_hdr)); f->seqs.push_back(std::move(f->cur_seq));
        f->cur_hdr.clear(); f->cur_seq.clear();
    };
    auto full = [&]() { return (max_records > 0 && (int32_t)f->hdrs.size() >= max_records) || (max_bases > 0 && bases >= max_bases); };
    if (f->pending) { emit(); f->pending = false; }
    const char* line; size_t len;
    while (!full() && next_line(f, &line, &len)) {
        if (len > 0 && line[0] == '>') {
            if (f->in_record) emit();
            f->in_record = true; f->seen_header = true;
            size_t e = len;
            while (e > 1 && is_space(line[e - 1])) e--;
            f->cur_hdr.assign(line + 1, e - 1);
            f->cur_seq.clear();
            continue;
        }
        size_t a = 0, e = len;
        while (a < e && is_space(line[a])) a++;
        while (e > a && is_space(line[e - 1])) e--;
        if (e == a) continue;
        if (f->in_record) f->cur_seq.append(line + a, e - a);
        else f->seen_text = true;                  // text before the first header: dropped, unless no header ever comes
    }
    if (!f->err.empty()) return PGA_EINVAL;
    if (!full() && f->in_record && f->eof && f->pos >= f->end) { emit(); f->in_record = false; }
    if (f->hdrs.empty() && f->eof && !f->seen_header && f->seen_text) { f->err = "not in FASTA format"; return PGA_EINVAL; }
    return PGA_OK;
}

// ---- mapped source ------------------------------------------------------------------------------------------------------
// the first header line at or after p (a '>' at the start of a line), or e
static const char* next_header(const char* p, const char* e, const char* file_begin) {
    while (p < e) {
        if (*p == '>' && (p == file_begin || p[-1] == '\n')) return p;
        const char* q = (const char*)memchr(p, '\n', (size_t)(e - p));
        if (!q) return e;
        p = q + 1;
    }
    return e;
}

// pass 1: the records of a piece and their sizes
static void measure_piece(Piece& P) {
    const char* p = P.b;
    Rec* cur = nullptr;
    while (p < P.e) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(P.e - p));
        const char* le = nl ? nl : P.e;
        if (*p == '>') {
            const char* he = le;
            while (he > p + 1 && is_space(he[-1])) he--;
            P.recs.push_back(Rec{p + 1, (uint32_t)(he - p - 1), 0});
            cur = &P.recs.back();
        } else {
            const char* a = p; const char* e = le;
            while (a < e && is_space(*a)) a++;
            while (e > a && is_space(e[-1])) e--;
            if (e > a) { if (cur) cur->bases += e - a; else { P.head_bases += e - a; P.head_text = true; } }
        }
        p = nl ? nl + 1 : P.e;
    }
}
// pass 2: the stripped sequence lines of the piece's records [first, last) copied to dst (record r at dst_off[r]); the piece's
// head text (it continues the last record of an earlier piece) to head_dst when that record is part of the batch
static void copy_piece(const Piece& P, size_t first, size_t last, char* const* dst, char* head_dst) {
    const char* p = P.b;
    size_t r = (size_t)-1;                // the record being copied (index into P.recs), -1: the head
    char* out = head_dst;
    while (p < P.e) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(P.e - p));
        const char* le = nl ? nl : P.e;
        if (*p == '>') {
            r = r == (size_t)-1 ? 0 : r + 1;
            if (r >= last) return;
            out = r >= first ? dst[r] : nullptr;
        } else if (out) {
            const char* a = p; const char* e = le;
            while (a < e && is_space(*a)) a++;
            while (e > a && is_space(e[-1])) e--;
            if (e > a) { memcpy(out, a, (size_t)(e - a)); out += e - a; }
        }
        p = nl ? nl + 1 : P.e;
    }
}

template <class F> static void run_threads(int n, F&& fn) {
    if (n <= 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < n; t++) th.emplace_back([&fn, t] { fn(t); });
    fn(0);
    for (auto& t : th) t.join();
}

// One batch from the mapped file into `arena_of(total bytes)`; fills hdrs / lens / offs.
template <class ArenaOf>
static int next_mapped(pga_fasta* f, int64_t max_bases, int32_t max_records, ArenaOf&& arena_of, char** arena_out) {
    *arena_out = nullptr;
    const char* const fb = f->map;
    const char* const fe = f->map + f->map_len;
    if (!f->seen_header) {
        // text before the first header is dropped; a file with text and no header at all is not FASTA
        const char* h = next_header(fb + f->map_pos, fe, fb);
        for (const char* q = fb + f->map_pos; q < h; q++) if (!is_space(*q)) { f->seen_text = true; break; }
        f->map_pos = (size_t)(h - fb);
        if (h == fe) {
            if (f->seen_text) { f->err = "not in FASTA format"; return PGA_EINVAL; }
            return PGA_OK;
        }
        f->seen_header = true;
    }
    if (f->map_pos >= f->map_len) return PGA_OK;
    const char* const b0 = fb + f->map_pos;               // at a header line
    // the window that is measured: the budget plus room for line ends and header lines, doubled below when that was not enough (no
    // fixed floor: a caller asking for one short record at a time must not pay for a megabyte per call)
    size_t window = max_bases > 0 ? (size_t)max_bases + (size_t)max_bases / 32 + 4096 : f->map_len;
    if (max_records > 0 && max_bases <= 0) window = std::min(window, (size_t)max_records * (1u << 16));
    for (;;) {
        const char* we = (size_t)(fe - b0) <= window ? fe : next_header(b0 + window, fe, fb);
        // pieces at record boundaries
        const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)f->threads, (size_t)(we - b0) >> 20));
        std::vector<Piece> pieces((size_t)T);
        {
            const char* s = b0;
            for (int t = 0; t < T; t++) {
                const char* want = t + 1 == T ? we : b0 + (size_t)(we - b0) / (size_t)T * (size_t)(t + 1);
                const char* e = t + 1 == T ? we : next_header(want < s ? s : want, we, fb);
                pieces[(size_t)t].b = s; pieces[(size_t)t].e = e;
                s = e;
            }
        }
        run_threads(T, [&](int t) { measure_piece(pieces[(size_t)t]); });
        // the records of the window in order; a piece's head text belongs to the last record before it (only when a single
        // record is larger than a piece can that happen: pieces start at headers)
        size_t nrec = 0;
        for (auto& P : pieces) nrec += P.recs.size();
        std::vector<int64_t> bases; bases.reserve(nrec);
        for (auto& P : pieces) {
            if (P.head_bases && !bases.empty()) bases.back() += P.head_bases;
            for (auto& r : P.recs) bases.push_back(r.bases);
        }
        // the cut: the shortest run of records that reaches the base budget (or the record budget); everything if the file ends here
        size_t cut = 0; int64_t acc = 0;
        while (cut < nrec && !((max_records > 0 && (int32_t)cut >= max_records) || (max_bases > 0 && acc >= max_bases))) acc += bases[cut++];
        const bool budget_hit = (max_records > 0 && (int32_t)cut >= max_records) || (max_bases > 0 && acc >= max_bases);
        if (!budget_hit && we < fe) { window *= 2; continue; }      // the window ended before the budget was reached: a larger one
        // offsets
        f->lens.assign(bases.begin(), bases.begin() + (long)cut);
        f->offs.assign(cut + 1, 0);
        for (size_t r = 0; r < cut; r++) f->offs[r + 1] = f->offs[r] + f->lens[r];
        char* arena = arena_of((size_t)f->offs[cut] + 16);
        if (!arena) return PGA_ENOMEM;
        *arena_out = arena;
        // per piece: destination of each of its records, and of its head text
        std::vector<std::vector<char*>> dst((size_t)T);
        std::vector<char*> head_dst((size_t)T, nullptr);
        std::vector<size_t> first_rec((size_t)T), take((size_t)T);
        {
            size_t g = 0;
            std::vector<int64_t> written(cut, 0);             // bytes of record r written by earlier pieces (head text continues a record)
            for (int t = 0; t < T; t++) {
                Piece& P = pieces[(size_t)t];
                if (P.head_bases && g > 0 && g - 1 < cut) head_dst[(size_t)t] = arena + f->offs[g - 1] + written[g - 1];
                if (P.head_bases && g > 0 && g - 1 < cut) written[g - 1] += P.head_bases;
                first_rec[(size_t)t] = g;
                dst[(size_t)t].resize(P.recs.size(), nullptr);
                size_t k = 0;
                for (; k < P.recs.size() && g + k < cut; k++) {
                    dst[(size_t)t][k] = arena + f->offs[g + k];
                    written[g + k] += P.recs[k].bases;
                    f->hdrs.emplace_back(P.recs[k].hdr, P.recs[k].hdr_len);
                }
                take[(size_t)t] = k;
                g += P.recs.size();
            }
        }
        run_threads(T, [&](int t) {
            const Piece& P = pieces[(size_t)t];
            if (take[(size_t)t] == 0 && !head_dst[(size_t)t]) return;
            copy_piece(P, 0, take[(size_t)t], dst[(size_t)t].data(), head_dst[(size_t)t]);
        });
        // the next batch starts at the first record that was not taken
        if (cut == nrec) f->map_pos = (size_t)(we - fb);
        else {
            size_t g = 0;
            for (auto& P : pieces) {
                if (cut < g + P.recs.size()) { f->map_pos = (size_t)(P.recs[cut - g].hdr - 1 - fb); break; }
                g += P.recs.size();
            }
        }
        return PGA_OK;
    }
}

// ---- C-ABI --------------------------------------------------------------------------------------------------------------
static const char* compression_of(const unsigned char* m, size_t n) {
    if (n >= 3 && m[0] == 'B' && m[1] == 'Z' && m[2] == 'h') return "bzip2";
    if (n >= 5 && m[0] == 0xfd && m[1] == '7' && m[2] == 'z' && m[3] == 'X' && m[4] == 'Z') return "xz";
    if (n >= 4 && m[0] == 0x04 && m[1] == 0x22 && m[2] == 0x4d && m[3] == 0x18) return "lz4";
    if (n >= 4 && m[0] == 0x28 && m[1] == 0xb5 && m[2] == 0x2f && m[3] == 0xfd) return "zstd";
    return nullptr;
}

extern "C" int pga_fasta_open(const char* path, pga_fasta** out) {
    if (out) *out = nullptr;
    if (!path || !out) return PGA_EINVAL;
    pga_fasta* f = new (std::nothrow) pga_fasta();
    if (!f) return PGA_ENOMEM;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { delete f; return PGA_EINVAL; }
    unsigned char magic[8] = {0};
    const ssize_t got = pread(fd, magic, sizeof magic, 0);
    struct stat sb;
    const bool regular = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
    if (got >= 4 && compression_of(magic, (size_t)got)) {
        // the reference's reader sniffs these too (tests/fasta.py:16-57) and decompresses with Python modules: so does the
        // Python layer here, through pga_fasta_open_callback
        close(fd); delete f;
        return PGA_EINVAL;
    }
    const bool gz = got >= 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    if (!gz && regular && sb.st_size > 0 && !getenv("PGA_FASTA_NO_MMAP")) {
        void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
            (void)madvise(m, (size_t)sb.st_size, MADV_SEQUENTIAL);
            f->map = (const char*)m; f->map_len = (size_t)sb.st_size; f->fd = fd;
            const unsigned hw = std::thread::hardware_concurrency();
            f->threads = (int)std::max(1u, std::min(16u, hw / 4));
            if (const char* e = getenv("PGA_FASTA_THREADS")) f->threads = std::max(1, atoi(e));
            *out = f;
            return PGA_OK;
        }
    }
    close(fd);
    f->gz = gzopen(path, "rb");           // zlib reads plain files transparently (empty files, pipes)
    if (!f->gz) { delete f; return PGA_EINVAL; }
    gzbuffer(f->gz, 1 << 20);
    f->buf.resize(1 << 22);
    *out = f;
    return PGA_OK;
}

extern "C" int pga_fasta_open_callback(pga_fasta_read_fn read, void* user, pga_fasta** out) {
    if (out) *out = nullptr;
    if (!read || !out) return PGA_EINVAL;
    pga_fasta* f = new (std::nothrow) pga_fasta();
    if (!f) return PGA_ENOMEM;
    f->cb = read; f->cb_user = user;
    f->buf.resize(1 << 22);
    *out = f;
    return PGA_OK;
}

extern "C" void pga_fasta_close(pga_fasta* f) {
    if (!f) return;
    if (f->gz) gzclose(f->gz);
    if (f->map) munmap((void*)f->map, f->map_len);
    if (f->fd >= 0) close(f->fd);
    {
        std::lock_guard<std::mutex> g(g_spare_mu);
        for (auto& a : f->arenas) spare_put(a);
    }
    delete f;
}

extern "C" const char* pga_fasta_error(const pga_fasta* f) { return f ? f->err.c_str() : "null reader"; }

// Up to max_records records / about max_bases bases (at least one record if any is left).  The returned
// arrays stay valid until the next call on the same reader.  *n_records == 0 means end of file.
extern "C" int pga_fasta_next(pga_fasta* f, int64_t max_bases, int32_t max_records, int32_t* n_records,
                              const char* const** headers, const char* const** seqs, const int64_t** lens) {
    if (!f || !n_records) return PGA_EINVAL;
    *n_records = 0;
    f->hdrs.clear(); f->seqs.clear(); f->lens.clear(); f->p_hdr.clear(); f->p_seq.clear(); f->offs.clear();
    if (f->map) {
        char* arena = nullptr;
        const int rc = next_mapped(f, max_bases, max_records, [&](size_t bytes) { f->plain_arena.resize(bytes); return f->plain_arena.data(); }, &arena);
        if (rc != PGA_OK) return rc;
        for (size_t i = 0; i < f->hdrs.size(); i++) { f->p_hdr.push_back(f->hdrs[i].c_str()); f->p_seq.push_back(arena + f->offs[i]); }
    } else {
        const int rc = next_stream(f, max_bases, max_records);
        if (rc != PGA_OK) return rc;
        for (size_t i = 0; i < f->hdrs.size(); i++) {
            f->p_hdr.push_back(f->hdrs[i].c_str()); f->p_seq.push_back(f->seqs[i].data()); f->lens.push_back((int64_t)f->seqs[i].size());
        }
    }
    *n_records = (int32_t)f->hdrs.size();
    if (headers) *headers = f->p_hdr.data();
    if (seqs) *seqs = f->p_seq.data();
    if (lens) *lens = f->lens.data();
    return PGA_OK;
}

static char* take_arena(pga_fasta* f, size_t bytes) {
    pga_fasta::Arena& a = f->arenas[f->next_arena];
    f->next_arena = (f->next_arena + 1) % f->arenas.size();
    if (a.cap < bytes) {
        pga_fasta::Arena old = a;
        a = pga_fasta::Arena();
        {
            std::lock_guard<std::mutex> g(g_spare_mu);
            for (size_t k = 0; k < g_spare.size(); k++)
                if (g_spare[k].cap >= bytes) { a = g_spare[k]; g_spare_bytes -= a.cap; g_spare.erase(g_spare.begin() + (long)k); break; }
            spare_put(old);
        }
        if (!a.p) {
            const size_t want = bytes + bytes / 4 + 4096;
            if (hipHostMalloc((void**)&a.p, want, hipHostMallocPortable) != hipSuccess) { a.p = nullptr; a.cap = 0; f->err = "hipHostMalloc failed for a staging arena"; return nullptr; }
            a.cap = want;
        }
    }
    return a.p;
}

// The batch of pga_fasta_next with its sequences back to back in the next pinned staging arena.
extern "C" int pga_fasta_next_packed(pga_fasta* f, int64_t max_bases, int32_t max_records, int32_t n_arenas, int32_t* n_records,
                                     const char* const** headers, const char** packed, const int64_t** offs, const int64_t** lens) {
    if (!f || !n_records || !packed || !offs) return PGA_EINVAL;
    if (f->arenas.empty()) f->arenas.resize((size_t)(n_arenas < 2 ? 2 : (n_arenas > 8 ? 8 : n_arenas)));
    *packed = nullptr; *offs = nullptr; *n_records = 0;
    if (f->map) {
        f->hdrs.clear(); f->seqs.clear(); f->lens.clear(); f->p_hdr.clear(); f->p_seq.clear(); f->offs.clear();
        char* arena = nullptr;
        const int rc = next_mapped(f, max_bases, max_records, [&](size_t bytes) { return take_arena(f, bytes); }, &arena);
        if (rc != PGA_OK) return rc;
        for (size_t i = 0; i < f->hdrs.size(); i++) f->p_hdr.push_back(f->hdrs[i].c_str());
        *n_records = (int32_t)f->hdrs.size();
        if (headers) *headers = f->p_hdr.data();
        if (lens) *lens = f->lens.data();
        if (*n_records == 0) return PGA_OK;
        *packed = arena; *offs = f->offs.data();
        return PGA_OK;
    }
    const char* const* seqs = nullptr;
    const int64_t* ln = nullptr;
    const int rc = pga_fasta_next(f, max_bases, max_records, n_records, headers, &seqs, &ln);
    if (rc != PGA_OK) return rc;
    if (lens) *lens = ln;
    if (*n_records == 0) return PGA_OK;
    size_t total = 0;
    f->offs.assign((size_t)*n_records + 1, 0);
    for (int32_t i = 0; i < *n_records; i++) { f->offs[(size_t)i] = (int64_t)total; total += (size_t)ln[i]; }
    f->offs[(size_t)*n_records] = (int64_t)total;
    char* a = take_arena(f, total + 16);
    if (!a) return PGA_ENOMEM;
    for (int32_t i = 0; i < *n_records; i++) memcpy(a + f->offs[(size_t)i], seqs[i], (size_t)ln[i]);
    *packed = a; *offs = f->offs.data();
    return PGA_OK;
}
