// FASTA ingest for the finder: multi-record, plain or gzip, straight into batches of contigs.
//
// Restates what the reference feeds to GeneFinder.find_genes (ref: src/pyrodigal/tests/fasta.py:59-86 `parse`,
// src/pyrodigal/cli.py:32-61): a record starts at a line beginning with '>', its id is the first
// whitespace-delimited word of that line and its description the rest; every other non-blank line, stripped
// of surrounding whitespace, is appended to the sequence; text before the first header is dropped, and a
// file with sequence text but no header at all is not FASTA.  Records are handed out in batches bounded by a
// base budget so that a caller can keep one batch on the device while the next one is being read.
#include <hip/hip_runtime.h>
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/pyrodigal_amd.h"

struct pga_fasta {
    gzFile gz = nullptr;                 // zlib reads plain files transparently
    std::string err;
    std::vector<char> buf;               // read window
    size_t pos = 0, end = 0;
    bool eof = false;
    bool in_record = false, seen_header = false, seen_text = false;
    // the record being assembled and the batch handed out last
    std::string cur_hdr;
    std::string cur_seq;
    std::vector<std::string> hdrs, seqs;
    std::vector<const char*> p_hdr, p_seq;
    std::vector<int64_t> lens;
    bool pending = false;                // cur_* holds a finished record that did not fit the previous batch
    // packed mode: staging arenas in pinned host memory, filled in turn
    struct Arena { char* p = nullptr; size_t cap = 0; };
    std::vector<Arena> arenas;
    size_t next_arena = 0;
    std::vector<int64_t> offs;
};

static bool fill(pga_fasta* f) {
    if (f->eof) return false;
    if (f->pos > 0) { memmove(f->buf.data(), f->buf.data() + f->pos, f->end - f->pos); f->end -= f->pos; f->pos = 0; }
    if (f->end == f->buf.size()) f->buf.resize(f->buf.size() * 2);
    const int n = gzread(f->gz, f->buf.data() + f->end, (unsigned)(f->buf.size() - f->end));
    if (n < 0) { int e; f->err = gzerror(f->gz, &e); f->eof = true; return false; }
    if (n == 0) { f->eof = true; return false; }
    f->end += (size_t)n;
    return true;
}

static inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

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

extern "C" int pga_fasta_open(const char* path, pga_fasta** out) {
    if (out) *out = nullptr;
    if (!path || !out) return PGA_EINVAL;
    pga_fasta* f = new (std::nothrow) pga_fasta();
    if (!f) return PGA_ENOMEM;
    f->gz = gzopen(path, "rb");
    if (!f->gz) { delete f; return PGA_EINVAL; }
    gzbuffer(f->gz, 1 << 20);
    f->buf.resize(1 << 22);
    *out = f;
    return PGA_OK;
}

extern "C" void pga_fasta_close(pga_fasta* f) {
    if (!f) return;
    if (f->gz) gzclose(f->gz);
    for (auto& a : f->arenas) if (a.p) hipHostFree(a.p);
    delete f;
}

extern "C" const char* pga_fasta_error(const pga_fasta* f) { return f ? f->err.c_str() : "null reader"; }

// Up to max_records records / about max_bases bases (at least one record if any is left).  The returned
// arrays stay valid until the next call on the same reader.  *n_records == 0 means end of file.
extern "C" int pga_fasta_next(pga_fasta* f, int64_t max_bases, int32_t max_records, int32_t* n_records,
                              const char* const** headers, const char* const** seqs, const int64_t** lens) {
    if (!f || !n_records) return PGA_EINVAL;
    *n_records = 0;
    f->hdrs.clear(); f->seqs.clear(); f->lens.clear(); f->p_hdr.clear(); f->p_seq.clear();
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
    for (size_t i = 0; i < f->hdrs.size(); i++) {
        f->p_hdr.push_back(f->hdrs[i].c_str()); f->p_seq.push_back(f->seqs[i].data()); f->lens.push_back((int64_t)f->seqs[i].size());
    }
    *n_records = (int32_t)f->hdrs.size();
    if (headers) *headers = f->p_hdr.data();
    if (seqs) *seqs = f->p_seq.data();
    if (lens) *lens = f->lens.data();
    return PGA_OK;
}


// The batch of pga_fasta_next, sequences copied back to back into the next pinned staging arena.
extern "C" int pga_fasta_next_packed(pga_fasta* f, int64_t max_bases, int32_t max_records, int32_t n_arenas, int32_t* n_records,
                                     const char* const** headers, const char** packed, const int64_t** offs, const int64_t** lens) {
    if (!f || !n_records || !packed || !offs) return PGA_EINVAL;
    if (f->arenas.empty()) f->arenas.resize((size_t)(n_arenas < 2 ? 2 : (n_arenas > 8 ? 8 : n_arenas)));
    const char* const* seqs = nullptr;
    const int64_t* ln = nullptr;
    const int rc = pga_fasta_next(f, max_bases, max_records, n_records, headers, &seqs, &ln);
    if (rc != PGA_OK) return rc;
    *packed = nullptr; *offs = nullptr;
    if (lens) *lens = ln;
    if (*n_records == 0) return PGA_OK;
    size_t total = 0;
    f->offs.assign((size_t)*n_records + 1, 0);
    for (int32_t i = 0; i < *n_records; i++) { f->offs[(size_t)i] = (int64_t)total; total += (size_t)ln[i]; }
    f->offs[(size_t)*n_records] = (int64_t)total;
    pga_fasta::Arena& a = f->arenas[f->next_arena];
    f->next_arena = (f->next_arena + 1) % f->arenas.size();
    if (a.cap < total + 16) {
        if (a.p) { hipHostFree(a.p); a.p = nullptr; a.cap = 0; }
        const size_t want = total + total / 4 + 4096;
        if (hipHostMalloc((void**)&a.p, want, hipHostMallocDefault) != hipSuccess) { a.p = nullptr; f->err = "hipHostMalloc failed for a staging arena"; return PGA_ENOMEM; }
        a.cap = want;
    }
    for (int32_t i = 0; i < *n_records; i++) memcpy(a.p + f->offs[(size_t)i], seqs[i], (size_t)ln[i]);
    *packed = a.p; *offs = f->offs.data();
    return PGA_OK;
}
