/*
 * prodigal_oracle.c -- CPU restatement of the pyrodigal gene-finding path.
 * TEST INFRASTRUCTURE ONLY (see prodigal_oracle.h).  Plain C, IEEE doubles,
 * compiled with -ffp-contract=off so every operation rounds exactly like the
 * reference's x86-64 build.  All "ref:" citations are relative to
 * /root/reference/src/pyrodigal unless another root is given.
 */
#define _GNU_SOURCE          /* CPU_SET, pthread_setaffinity_np (the all-core baseline driver at the end of the file) */
#include "prodigal_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { NA = 0, NG = 1, NC = 2, NT = 3, NN = 6 };      /* ref: _sequence.h:8-14 */
enum { T_ATG = 0, T_GTG = 1, T_TTG = 2, T_STOP = 3 }; /* ref: prodigal/sequence.pxd:17-21 */

#define MAX_NODE_DIST 500   /* ref: _connection.h / dprog.h */
#define MAX_OPP_OVLP  200
#define OPER_DIST     60    /* ref: /root/reference/src/Prodigal/node.h:30-38 */
#define EDGE_BONUS    0.74
#define EDGE_UPS      (-1.00)
#define META_PEN      7.5
#define GC_WINDOW     120   /* ref: lib.pyx:171 */

struct po_ctx {
    int      slen;
    uint8_t* dig;
    double   gc, gc_known;
    int      unknown;
    int      nmask;
    int    (*mask)[2];
    po_node* nod;
    int      nn, ncap;
    po_gene* gen;
    int      ng, gcap;
    uint8_t *k_type, *k_strand, *k_frame, *k_skip;
    int      kcap;
    int      ipath;
    double   path_score;
};

/* ---------------------------------------------------------------- sequence */

/* ref: lib.pyx:664-697 (Sequence._build) and 699-713 (Sequence._mask) */
po_ctx* po_new(const char* ascii, int64_t len, int mask, int mask_size) {
    po_ctx* c = (po_ctx*)calloc(1, sizeof(po_ctx));
    if (!c) return NULL;
    c->slen = (int)len;
    c->dig = (uint8_t*)malloc(len > 0 ? (size_t)len : 1);
    long gcn = 0; int unk = 0;
    for (int64_t i = 0; i < len; i++) {
        switch (ascii[i]) {
            case 'A': case 'a': c->dig[i] = NA; break;
            case 'T': case 't': c->dig[i] = NT; break;
            case 'G': case 'g': c->dig[i] = NG; gcn++; break;
            case 'C': case 'c': c->dig[i] = NC; gcn++; break;
            default: c->dig[i] = NN; unk++;
        }
    }
    c->unknown = unk;
    if (len > 0) c->gc = (double)gcn / (double)len;
    if (len > unk) c->gc_known = (double)gcn / ((double)len - unk);
    c->ipath = -1;
    if (mask) {
        int cap = 8; c->mask = malloc(sizeof(int[2]) * (cap + 1));
        int mb = -1;
        for (int i = 0; i <= c->slen; i++) {
            int isn = (i < c->slen) && c->dig[i] == NN;
            if (isn) { if (mb == -1) mb = i; continue; }
            if (mb != -1) {
                if (i == c->slen || i >= mask_size + mb) {
                    if (c->nmask == cap) { cap *= 2; c->mask = realloc(c->mask, sizeof(int[2]) * (cap + 1)); }
                    c->mask[c->nmask][0] = mb; c->mask[c->nmask][1] = i; c->nmask++;
                }
                mb = -1;
            }
        }
        c->mask[c->nmask][0] = 0; c->mask[c->nmask][1] = 0; /* sentinel, see walk below */
    }
    return c;
}

/* masked regions as [begin, end) pairs; returns their number (ref: lib.pyx:699-713, Sequence.masks) */
int po_masks(const po_ctx* c, int32_t* out, int cap) {
    for (int k = 0; k < c->nmask && k < cap; k++) { out[2 * k] = c->mask[k][0]; out[2 * k + 1] = c->mask[k][1]; }
    return c->nmask;
}

void po_free(po_ctx* c) {
    if (!c) return;
    free(c->dig); free(c->mask); free(c->nod); free(c->gen);
    free(c->k_type); free(c->k_strand); free(c->k_frame); free(c->k_skip);
    free(c);
}

int po_slen(const po_ctx* c) { return c->slen; }
double po_gc(const po_ctx* c) { return c->gc; }
int po_unknown(const po_ctx* c) { return c->unknown; }
const uint8_t* po_digits(const po_ctx* c) { return c->dig; }
int po_node_size(void) { return (int)sizeof(po_node); }
int po_num_nodes(const po_ctx* c) { return c->nn; }
po_node* po_nodes(po_ctx* c) { return c->nod; }
int po_num_genes(const po_ctx* c) { return c->ng; }
po_gene* po_genes(po_ctx* c) { return c->gen; }
double po_last_path_score(const po_ctx* c) { return c->path_score; }
int po_last_ipath(const po_ctx* c) { return c->ipath; }

/* base `i` of strand-local coordinates; reverse strand is virtual.
 * ref: _sequence.h:45-55 */
static inline int base(const po_ctx* c, int i, int strand) {
    return strand == 1 ? c->dig[i] : (c->dig[c->slen - 1 - i] ^ 3);
}
/* ref: _sequence.h:35-43 (unknown bases count as GC) */
static inline int is_gc_fwd(const po_ctx* c, int i) {
    int d = c->dig[i];
    return d != NA && d != NT;
}
static inline int comp2(int d) { return d <= 3 ? (d ^ 3) : NN; }  /* ref: _sequence.h:16 */

/* ref: _sequence.h:207-220 */
static inline int mer_ndx(const po_ctx* c, int i, int len, int strand) {
    int v = 0;
    if (strand == 1) {
        for (int j = 0; j < len; j++) v |= (c->dig[i + j] & 3) << (2 * j);
    } else {
        int k = c->slen - 1 - i;
        for (int j = 0; j < len; j++) v |= (comp2(c->dig[k - j]) & 3) << (2 * j);
    }
    return v;
}

static uint64_t tt_set(const int* l, int n) { uint64_t m = 0; for (int i = 0; i < n; i++) m |= 1ull << l[i]; return m; }

/* ref: _sequence.h:117-157 */
static int is_stop(const po_ctx* c, int i, int tt, int strand) {
    static uint64_t taa = 0, tag = 0, tga = 0;
    if (!taa) {
        static const int a[] = {1,2,3,4,5,9,10,11,12,13,15,16,21,22,23,24,25,26,32};
        static const int g[] = {1,2,3,4,5,9,10,11,12,13,14,21,23,24,25,26,33};
        static const int o[] = {1,6,11,12,15,16,22,23,26,29,30,32};
        tag = tt_set(g, sizeof g / sizeof *g); tga = tt_set(o, sizeof o / sizeof *o);
        taa = tt_set(a, sizeof a / sizeof *a);
    }
    int x0 = base(c, i, strand), x1 = base(c, i + 1, strand), x2 = base(c, i + 2, strand);
    if (x0 == NT && x1 == NA && x2 == NG) return (tag >> tt) & 1;
    if (x0 == NT && x1 == NG && x2 == NA) return (tga >> tt) & 1;
    if (x0 == NT && x1 == NA && x2 == NA) return (taa >> tt) & 1;
    if (tt == 2)  return x0 == NA && x1 == NG && (x2 == NA || x2 == NG);
    if (tt == 22) return x0 == NT && x1 == NC && x2 == NA;
    if (tt == 23) return x0 == NT && x1 == NT && x2 == NA;
    return 0;
}

/* ref: _sequence.h:45-73 */
static int is_start(const po_ctx* c, int i, int tt, int strand) {
    int x0 = base(c, i, strand), x1 = base(c, i + 1, strand), x2 = base(c, i + 2, strand);
    if (x1 != NT || x2 != NG) return 0;
    if (x0 == NA) return 1;
    if (tt == 6 || tt == 10 || tt == 14 || tt == 15 || tt == 16 || tt == 2) return 0;
    if (x0 == NG) return !(tt == 1 || tt == 3 || tt == 12 || tt == 2);
    if (x0 == NT) return !(tt < 4 || tt == 9 || (tt >= 21 && tt < 25));
    return 0;
}

/* ------------------------------------------------------------------- nodes */

static po_node* push_node(po_ctx* c, int ndx, int type, int strand, int stop_val, int edge) {
    if (c->nn == c->ncap) {
        int ncap = c->ncap ? c->ncap + (c->ncap >> 1) + 16 : 1024;
        c->nod = (po_node*)realloc(c->nod, sizeof(po_node) * ncap);
        memset(c->nod + c->ncap, 0, sizeof(po_node) * (ncap - c->ncap));
        c->ncap = ncap;
    }
    po_node* n = &c->nod[c->nn++];
    n->ndx = ndx; n->type = (uint8_t)type; n->strand = (int8_t)strand;
    n->stop_val = stop_val; n->edge = (uint8_t)edge;
    return n;
}

static inline int mask_hits(const int m[2], int b, int e) { return m[0] < e && b < m[1]; } /* ref: lib.pyx:336-340 */

/* One right-to-left sweep over one strand, three frame automata.
 * ref: lib.pyx:1905-2117 (Nodes._extract; forward half 1930-2020, reverse half 2022-2115) */
static void sweep_strand(po_ctx* c, int strand, int tt, const po_params* p) {
    const int slen = c->slen, slmod = slen % 3;
    int last[3], saw[3], mind[3], mk[3];
    for (int f = 0; f < 3; f++) {
        int fr = (f + slmod) % 3;
        last[fr] = slen + f;
        saw[f] = 0; mind[f] = p->min_edge_gene;
        if (!p->closed) while (last[fr] + 3 > slen) last[fr] -= 3;
        mk[f] = c->nmask > 0 ? (strand == 1 ? c->nmask - 1 : 0) : -1;
    }
    for (int i = slen - 3; i >= 0; i--) {
        const int fr = i % 3;
        if (is_stop(c, i, tt, strand)) {
            if (saw[fr]) {
                int edge = !is_stop(c, last[fr], tt, strand);
                if (strand == 1) push_node(c, last[fr], T_STOP, 1, i, edge);
                else push_node(c, slen - last[fr] - 1, T_STOP, -1, slen - i - 1, edge);
            }
            mind[fr] = p->min_gene; last[fr] = i; saw[fr] = 0;
            continue;
        }
        if (last[fr] >= slen) continue;
        /* region masks (ref: lib.pyx:1959-1966 / 2053-2061).  The reference's reverse walk can
         * step one entry past its mask array; the sentinel entry {0,0} makes that step defined. */
        if (strand == 1) {
            while (mk[fr] != -1 && last[fr] < c->mask[mk[fr]][0]) mk[fr] = (mk[fr] == 0) ? -1 : mk[fr] - 1;
            if (mk[fr] != -1 && mask_hits(c->mask[mk[fr]], i, last[fr])) continue;
        } else {
            while (mk[fr] != -1 && slen - last[fr] - 1 > c->mask[mk[fr]][1]) mk[fr] = (mk[fr] == c->nmask) ? -1 : mk[fr] + 1;
            if (mk[fr] != -1 && mask_hits(c->mask[mk[fr]], slen - last[fr] - 1, slen - i - 1)) continue;
        }
        if (last[fr] - i + 3 >= mind[fr] && is_start(c, i, tt, strand)) {
            int b = base(c, i, strand);
            int type = b == NA ? T_ATG : (b == NT ? T_TTG : T_GTG);
            saw[fr] = 1;
            if (strand == 1) push_node(c, i, type, 1, last[fr], 0);
            else push_node(c, slen - i - 1, type, -1, slen - last[fr] - 1, 0);
        } else if (i <= 2 && !p->closed && last[fr] - i > p->min_edge_gene) {
            saw[fr] = 1;
            if (strand == 1) push_node(c, i, T_ATG, 1, last[fr], 1);
            else push_node(c, slen - i - 1, T_ATG, -1, slen - last[fr] - 1, 1);
        }
    }
    for (int f = 0; f < 3; f++) {
        if (!saw[f]) continue;
        int edge = !is_stop(c, last[f], tt, strand);
        if (strand == 1) push_node(c, last[f], T_STOP, 1, f - 6, edge);
        else push_node(c, slen - last[f] - 1, T_STOP, -1, slen - f + 5, edge);
    }
}

int po_extract(po_ctx* c, int tt, const po_params* p) {
    if (c->nod) memset(c->nod, 0, sizeof(po_node) * c->nn);   /* Nodes._clear, ref: lib.pyx:1898-1903 */
    c->nn = 0;
    if (c->slen < 3) return 0;
    sweep_strand(c, 1, tt, p);
    sweep_strand(c, -1, tt, p);
    return c->nn;
}

/* Prodigal node.c compare_nodes (absent from checkout): ndx ascending, strand descending.
 * call site ref: lib.pyx:2489-2493 */
static int node_order(const void* a, const void* b) {
    const po_node* x = (const po_node*)a; const po_node* y = (const po_node*)b;
    if (x->ndx != y->ndx) return x->ndx < y->ndx ? -1 : 1;
    if (x->strand != y->strand) return x->strand > y->strand ? -1 : 1;
    return 0;
}
void po_sort(po_ctx* c) { qsort(c->nod, c->nn, sizeof(po_node), node_order); }

/* Prodigal node.c reset_node_scores (absent). call site ref: lib.pyx:2495-2497 */
void po_reset_scores(po_ctx* c) {
    for (int i = 0; i < c->nn; i++) {
        po_node* n = &c->nod[i];
        for (int j = 0; j < 3; j++) { n->star_ptr[j] = 0; n->gc_score[j] = 0.0; }
        n->rbs[0] = n->rbs[1] = 0;
        n->score = n->cscore = n->sscore = n->rscore = n->tscore = n->uscore = 0.0;
        n->traceb = n->tracef = -1; n->ov_mark = -1;
        n->elim = 0; n->gc_bias = 0;
        n->mot_score = 0.0; n->mot_ndx = 0; n->mot_len = n->mot_spacer = n->mot_spacendx = 0;
    }
}

/* ref: lib.pyx:1846-1896 (Nodes._calc_orf_gc), including the reverse-strand k-range quirk */
static void calc_orf_gc(po_ctx* c) {
    const int slen = c->slen; po_node* nod = c->nod;
    int last[3] = {0, 0, 0}; double gc[3] = {0, 0, 0};
    for (int i = c->nn - 1; i >= 0; i--) {
        if (nod[i].strand != 1) continue;
        int ph = nod[i].ndx % 3;
        if (nod[i].type == T_STOP) {
            int j = last[ph] = nod[i].ndx; gc[ph] = 0.0;
            for (int k = j; k < j + 3; k++) if (k >= 0 && k < slen) gc[ph] += is_gc_fwd(c, k);
        } else {
            for (int j = last[ph] - 3; j > nod[i].ndx - 1; j -= 3)
                for (int k = j; k < j + 3; k++) if (k >= 0 && k < slen) gc[ph] += is_gc_fwd(c, k);
            double gsize = abs(nod[i].stop_val - nod[i].ndx) + 3.0;
            nod[i].gc_cont = (float)(gc[ph] / gsize);
            last[ph] = nod[i].ndx;
        }
    }
    gc[0] = gc[1] = gc[2] = 0.0;
    for (int i = 0; i < c->nn; i++) {
        if (nod[i].strand != -1) continue;
        int ph = nod[i].ndx % 3;
        if (nod[i].type == T_STOP) {
            int j = last[ph] = nod[i].ndx; gc[ph] = 0.0;
            for (int k = j; k > j - 3; k--) if (k >= 0 && k < slen) gc[ph] += is_gc_fwd(c, k);
        } else {
            for (int j = last[ph] + 3; j < nod[i].ndx + 1; j += 3)
                for (int k = j; k < j + 3; k++) if (k >= 0 && k < slen) gc[ph] += is_gc_fwd(c, k);
            double gsize = abs(nod[i].stop_val - nod[i].ndx) + 3.0;
            nod[i].gc_cont = (float)(gc[ph] / gsize);
            last[ph] = nod[i].ndx;
        }
    }
}

/* ref: lib.pyx:2119-2239 (Nodes._raw_coding_score) */
static void raw_coding_score(po_ctx* c, const po_training* t) {
    const int nn = c->nn, slen = c->slen; po_node* nod = c->nod;
    double sc[3], no_stop, lfac, lmin, lmax, gsize, tmp;
    long last[3] = {0, 0, 0};
    if (t->trans_table != 11) {
        no_stop  = ((1 - t->gc) * (1 - t->gc) * t->gc) / 8.0;
        no_stop += ((1 - t->gc) * (1 - t->gc) * (1 - t->gc)) / 8.0;
    } else {
        no_stop  = ((1 - t->gc) * (1 - t->gc) * t->gc) / 4.0;
        no_stop += ((1 - t->gc) * (1 - t->gc) * (1 - t->gc)) / 8.0;
    }
    no_stop = 1 - no_stop;
    lmax = log((1 - pow(no_stop, 1000.0)) / pow(no_stop, 1000.0));
    lmin = log((1 - pow(no_stop, 80)) / pow(no_stop, 80));

    /* pass 1: hexamer log-odds accumulated from each stop outward */
    sc[0] = sc[1] = sc[2] = 0.0;
    for (int i = nn - 1; i >= 0; i--) {
        if (nod[i].strand != 1) continue;
        int ph = nod[i].ndx % 3;
        if (nod[i].type == T_STOP) { last[ph] = nod[i].ndx; sc[ph] = 0.0; continue; }
        for (long j = last[ph] - 3; j > nod[i].ndx - 1; j -= 3) sc[ph] += t->gene_dc[mer_ndx(c, (int)j, 6, 1)];
        nod[i].cscore = sc[ph]; last[ph] = nod[i].ndx;
    }
    sc[0] = sc[1] = sc[2] = 0.0;
    for (int i = 0; i < nn; i++) {
        if (nod[i].strand != -1) continue;
        int ph = nod[i].ndx % 3;
        if (nod[i].type == T_STOP) { last[ph] = nod[i].ndx; sc[ph] = 0.0; continue; }
        for (long j = last[ph] + 3; j < nod[i].ndx + 1; j += 3) sc[ph] += t->gene_dc[mer_ndx(c, slen - 1 - (int)j, 6, -1)];
        nod[i].cscore = sc[ph]; last[ph] = nod[i].ndx;
    }
    /* pass 2: penalise starts whose coding score does not exceed an upstream start's */
    for (int dir = 0; dir < 2; dir++) {
        sc[0] = sc[1] = sc[2] = -10000.0;
        for (int k = 0; k < nn; k++) {
            int i = dir == 0 ? k : nn - 1 - k;
            if (nod[i].strand != (dir == 0 ? 1 : -1)) continue;
            int ph = nod[i].ndx % 3;
            if (nod[i].type == T_STOP) sc[ph] = -10000.0;
            else if (nod[i].cscore > sc[ph]) sc[ph] = nod[i].cscore;
            else nod[i].cscore -= (sc[ph] - nod[i].cscore);
        }
    }
    /* pass 3: length factor.  NB: sc[] carries over from pass 2 (reference does not reset it). */
    for (int dir = 0; dir < 2; dir++) {
        for (int k = 0; k < nn; k++) {
            int i = dir == 0 ? k : nn - 1 - k;
            if (nod[i].strand != (dir == 0 ? 1 : -1)) continue;
            int ph = nod[i].ndx % 3;
            if (nod[i].type == T_STOP) { sc[ph] = -10000.0; continue; }
            if (dir == 0) gsize = (((double)nod[i].stop_val - nod[i].ndx) + 3.0) / 3.0;
            else          gsize = (((double)nod[i].ndx - nod[i].stop_val) + 3.0) / 3.0;
            if (gsize > 1000.0) lfac = (lmax - lmin) * (gsize - 80) / 920.0;
            else { tmp = pow(no_stop, gsize); lfac = log((1 - tmp) / tmp) - lmin; }
            if (lfac > sc[ph]) sc[ph] = lfac;
            else lfac -= fmax(fmin(sc[ph] - lfac, lfac), 0);
            if (lfac > 3.0 && nod[i].cscore < 0.5 * lfac) nod[i].cscore = 0.5 * lfac;
            nod[i].cscore += lfac;
        }
    }
}

/* ref: lib.pyx:791-881 (exact) and 883-979 (one mismatch); `mm` selects the variant.
 * The mismatch variant has no "else cur_val = 0": the previous value persists (kept). */
static int shine_dalgarno(const po_ctx* c, int pos, int start, const double* w, int strand, int mm) {
    int match[6], limit, maxv = 0, cur = 0;
    for (int i = 0; i < 6; i++) match[i] = -10;
    limit = start - 4 - pos; if (limit > 6) limit = 6;
    for (int i = 0; i < limit; i++) {
        int in = pos + i >= 0 && pos + i < c->slen;
        if (!mm) {
            if (!in) continue;
            if (i % 3 == 0) { if (base(c, pos + i, strand) == NA) match[i] = 2; }
            else            { if (base(c, pos + i, strand) == NG) match[i] = 3; }
        } else {
            if (i % 3 == 0) match[i] = (in && base(c, pos + i, strand) == NA) ? 2 : -3;
            else            match[i] = (in && base(c, pos + i, strand) == NG) ? 3 : -2;
        }
    }
    for (int i = limit; i > (mm ? 4 : 2); i--) {
        for (int j = 0; j < limit + 1 - i; j++) {
            int ctr = -2, mism = 0, flag, rdis;
            for (int k = j; k < j + i; k++) {
                ctr += match[k];
                if (mm && match[k] < 0) { mism++; if (k <= j + 1 || k >= j + i - 2) ctr -= 10; }
            }
            if (mm ? (mism != 1 || ctr < 6) : (ctr < 6)) continue;
            rdis = start - (pos + j + i);
            if (!mm) {
                if (rdis < 5) flag = i < 5 ? 2 : 1;
                else if (rdis < 11) flag = 0;
                else if (rdis < 13) flag = i < 5 ? 1 : 2;
                else if (rdis < 16) flag = 3;
                else continue;
                static const int v6[4] = {13, 6, 1, 2}, v8[4] = {15, 12, 11, 3}, v9[4] = {16, 12, 11, 3},
                                 v11[4] = {22, 21, 20, 10}, v12[4] = {24, 23, 20, 10}, v14[4] = {27, 26, 25, 10};
                switch (ctr) {
                    case 6: cur = v6[flag]; break;   case 8: cur = v8[flag]; break;
                    case 9: cur = v9[flag]; break;   case 11: cur = v11[flag]; break;
                    case 12: cur = v12[flag]; break; case 14: cur = v14[flag]; break;
                    default: cur = 0;
                }
            } else {
                if (rdis < 5) flag = 1;
                else if (rdis < 11) flag = 0;
                else if (rdis < 13) flag = 2;
                else if (rdis < 16) flag = 3;
                else continue;
                static const int m6[4] = {9, 5, 4, 2}, m7[4] = {14, 8, 7, 2}, m9[4] = {19, 18, 17, 3};
                switch (ctr) {
                    case 6: cur = m6[flag]; break; case 7: cur = m7[flag]; break; case 9: cur = m9[flag]; break;
                    default: break;
                }
            }
            if (w[cur] < w[maxv]) continue;
            if (w[cur] == w[maxv] && cur < maxv) continue;
            maxv = cur;
        }
    }
    return maxv;
}

/* ref: lib.pyx:2241-2277 (Nodes._rbs_score) */
static void rbs_score(po_ctx* c, const po_training* t) {
    const int slen = c->slen;
    for (int i = 0; i < c->nn; i++) {
        po_node* n = &c->nod[i];
        if (n->type == T_STOP || n->edge) continue;
        n->rbs[0] = n->rbs[1] = 0;
        int lo, hi, st;
        if (n->strand == 1) { lo = n->ndx - 20; hi = n->ndx - 5; st = n->ndx; }
        else { lo = slen - n->ndx - 21; hi = slen - n->ndx - 6; st = slen - 1 - n->ndx; }
        for (int j = lo; j < hi; j++) {
            if (n->strand == 1 ? j < 0 : j >= slen) continue;
            int a = shine_dalgarno(c, j, st, t->rbs_wt, n->strand, 0);
            int b = shine_dalgarno(c, j, st, t->rbs_wt, n->strand, 1);
            if (a > n->rbs[0]) n->rbs[0] = (uint8_t)a;
            if (b > n->rbs[1]) n->rbs[1] = (uint8_t)b;
        }
    }
}

/* ref: lib.pyx:1556-1616 (Node._find_best_upstream_motif) */
static void best_upstream_motif(const po_ctx* c, po_node* n, const po_training* t, int stage) {
    if (n->type == T_STOP || n->edge) return;
    int start = n->strand == 1 ? n->ndx : c->slen - 1 - n->ndx;
    int bsp = 0, bsi = 0, blen = 0, bndx = 0; double bsc = -100.0;
    for (int i = 3; i >= 0; i--) {
        for (int j = start - 18 - i; j < start - 5 - i; j++) {
            if (j < 0) continue;
            int si;
            if (j <= start - 16 - i) si = 3;
            else if (j <= start - 14 - i) si = 2;
            else if (j >= start - 7 - i) si = 1;
            else si = 0;
            int idx = mer_ndx(c, j, i + 3, n->strand);
            double s = t->mot_wt[i][si][idx];
            if (s > bsc) { bsc = s; bsi = si; bsp = start - j - i - 3; bndx = idx; blen = i + 3; }
        }
    }
    if (stage == 2 && (bsc == -4.0 || bsc < t->no_mot + 0.69)) {
        n->mot_ndx = 0; n->mot_len = 0; n->mot_spacendx = 0; n->mot_spacer = 0; n->mot_score = t->no_mot;
    } else {
        n->mot_ndx = bndx; n->mot_len = (uint8_t)blen; n->mot_spacendx = (uint8_t)bsi;
        n->mot_spacer = (uint8_t)(bsp & 15); n->mot_score = bsc;   /* 4-bit field in the reference struct */
    }
}

/* ref: lib.pyx:1618-1650 (Node._score_upstream_composition) */
static void upstream_composition(const po_ctx* c, po_node* n, const po_training* t) {
    int start = n->strand == 1 ? n->ndx : c->slen - 1 - n->ndx;
    int cnt = 0; double u = 0.0;
    for (int i = 1; i < 3; i++) {
        if (i > start) break;
        u += 0.4 * t->st_wt * t->ups_comp[cnt][mer_ndx(c, start - i, 1, n->strand)]; cnt++;
    }
    for (int i = 15; i < 45; i++) {
        if (i > start) break;
        u += 0.4 * t->st_wt * t->ups_comp[cnt][mer_ndx(c, start - i, 1, n->strand)]; cnt++;
    }
    n->uscore = u;
}

/* ref: lib.pyx:2331-2487 (Nodes._score) */
void po_score_nodes(po_ctx* c, const po_training* t, int closed, int is_meta) {
    const int nn = c->nn, slen = c->slen; po_node* nod = c->nod;
    calc_orf_gc(c);
    raw_coding_score(c, t);
    if (t->uses_sd) rbs_score(c, t);
    else for (int i = 0; i < nn; i++) { if (nod[i].type == T_STOP || nod[i].edge) continue; best_upstream_motif(c, &nod[i], t, 2); }

    for (int i = 0; i < nn; i++) {
        po_node* n = &nod[i];
        if (n->type == T_STOP) continue;
        long orf = n->ndx > n->stop_val ? n->ndx - n->stop_val : n->stop_val - n->ndx;
        double edge_gene = 0;
        if (n->edge) edge_gene += 1;
        if ((n->strand == 1 && !is_stop(c, n->stop_val, t->trans_table, 1)) ||
            (n->strand == -1 && !is_stop(c, slen - 1 - n->stop_val, t->trans_table, -1))) edge_gene += 1;

        if (n->edge) {
            n->tscore = EDGE_BONUS * t->st_wt / edge_gene; n->uscore = 0.0; n->rscore = 0.0;
        } else {
            n->tscore = t->type_wt[n->type] * t->st_wt;
            double r1 = t->rbs_wt[n->rbs[0]], r2 = t->rbs_wt[n->rbs[1]];
            double sd = fmax(r1, r2) * t->st_wt;
            if (t->uses_sd) n->rscore = sd;
            else {
                n->rscore = t->st_wt * n->mot_score;
                if (n->rscore < sd && t->no_mot > -0.5) n->rscore = sd;
            }
            upstream_composition(c, n, t);
            if (!closed && n->ndx <= 2 && n->strand == 1) n->uscore += EDGE_UPS * t->st_wt;
            else if (!closed && n->ndx >= slen - 3 && n->strand == -1) n->uscore += EDGE_UPS * t->st_wt;
            else if (i < 500 && n->strand == 1) {
                for (int j = i - 1; j >= 0; j--)
                    if (nod[j].edge && n->stop_val == nod[j].stop_val) { n->uscore += EDGE_UPS * t->st_wt; break; }
            } else if (i + 500 >= nn && n->strand == -1) {
                for (int j = i + 1; j < nn; j++)
                    if (nod[j].edge && n->stop_val == nod[j].stop_val) { n->uscore += EDGE_UPS * t->st_wt; break; }
            }
        }
        if (!closed && !n->edge && ((n->ndx <= 2 && n->strand == 1) || (n->ndx >= slen - 3 && n->strand == -1))) {
            edge_gene += 1; n->edge = 1; n->tscore = 0.0;
            n->uscore = EDGE_BONUS * t->st_wt / edge_gene; n->rscore = 0.0;
        }
        if (!n->edge && edge_gene == 1) n->uscore -= 0.5 * EDGE_BONUS * t->st_wt;
        if (edge_gene == 0 && orf < 250) {
            double negf = 250.0 / (float)orf, posf = (float)orf / 250.0;
            n->rscore *= n->rscore < 0 ? negf : posf;
            n->uscore *= n->uscore < 0 ? negf : posf;
            n->tscore *= n->tscore < 0 ? negf : posf;
        }
        if (is_meta && slen < 3000 && edge_gene == 0 && (n->cscore < 5.0 || orf < 120))
            n->cscore -= META_PEN * fmax(0, (3000.0 - slen) / 2700.0);
        n->sscore = n->tscore + n->rscore + n->uscore;
        if (n->cscore < 0.0) {
            if (edge_gene > 0 && !n->edge) {
                if (!is_meta || slen > 1500) n->sscore -= t->st_wt;
                else n->sscore -= 10.31 - 0.004 * slen;
            } else if (is_meta && slen < 3000 && n->edge) {
                double mml = sqrt((double)slen) * 5.0;
                if (orf >= mml) {
                    if (n->cscore >= 0) n->cscore = -1.0;
                    n->sscore = 0.0; n->uscore = 0.0;
                }
            } else n->sscore -= 0.5;
        } else if (is_meta && n->cscore < 5.0 && orf < 120 && n->sscore < 0.0) n->sscore -= t->st_wt;
    }
}

/* ---------------------------------------------------- connection scoring */

/* ref: _connection.h:52-78 */
static double igm_same(const po_node* a, const po_node* b, double st_wt) {
    int dist = abs(a->ndx - b->ndx);
    int ovl = a->ndx + 2 * a->strand >= b->ndx;
    double r = 0.0;
    if (a->ndx + 2 == b->ndx || a->ndx == b->ndx + 1) {
        if (a->strand == 1) { if (b->rscore < 0) r -= b->rscore; if (b->uscore < 0) r -= b->uscore; }
        else                { if (a->rscore < 0) r -= a->rscore; if (a->uscore < 0) r -= a->uscore; }
    }
    if (dist > 3 * OPER_DIST) r -= 0.15 * st_wt;
    else if ((dist <= OPER_DIST && !ovl) || dist * 4 < OPER_DIST) r += (2.0 - (double)dist / OPER_DIST) * 0.15 * st_wt;
    return r;
}
static inline double igm_diff(double st_wt) { return -0.15 * st_wt; }   /* ref: _connection.h:43-49 */
static double igm(const po_node* a, const po_node* b, double st_wt) {  /* ref: _connection.h:81-91 */
    return a->strand == b->strand ? igm_same(a, b, st_wt) : igm_diff(st_wt);
}

/* ref: lib.pyx:2279-2329 (Nodes._record_overlapping_starts) */
void po_overlapping_starts(po_ctx* c, const po_training* t, int flag, int maxov) {
    const int nn = c->nn; po_node* nod = c->nod;
    for (int i = 0; i < nn; i++) {
        po_node* s = &nod[i];
        s->star_ptr[0] = s->star_ptr[1] = s->star_ptr[2] = -1;
        if (s->type != T_STOP || s->edge == 1) continue;
        double best = -100;
        if (s->strand == 1) {
            for (int j = i + 3; j >= 0; j--) {
                if (j >= nn || nod[j].ndx > s->ndx + 2) continue;
                if (nod[j].ndx + maxov < s->ndx) break;
                if (nod[j].strand != 1 || nod[j].type == T_STOP) continue;
                if (nod[j].stop_val <= s->ndx) continue;
                int f = nod[j].ndx % 3;
                if (flag == 0) { if (s->star_ptr[f] == -1) s->star_ptr[f] = j; }
                else {
                    double v = nod[j].cscore + nod[j].sscore + igm_same(s, &nod[j], t->st_wt);
                    if (v > best) { s->star_ptr[f] = j; best = v; }
                }
            }
        } else {
            for (int j = i - 3; j < nn; j++) {
                if (j < 0 || nod[j].ndx < s->ndx - 2) continue;
                if (nod[j].ndx - maxov > s->ndx) break;
                if (nod[j].strand != -1 || nod[j].type == T_STOP) continue;
                if (nod[j].stop_val >= s->ndx) continue;
                int f = nod[j].ndx % 3;
                if (flag == 0) { if (s->star_ptr[f] == -1) s->star_ptr[f] = j; }
                else {
                    double v = nod[j].cscore + nod[j].sscore + igm_same(&nod[j], s, t->st_wt);
                    if (v > best) { s->star_ptr[f] = j; best = v; }
                }
            }
        }
    }
}

static inline double gc_bias_score(const po_node* n, const po_training* t) {
    return t->bias[0] * n->gc_score[0] + t->bias[1] * n->gc_score[1] + t->bias[2] * n->gc_score[2];
}

/* One candidate connection j -> i.  kind of i: 0 fwd start, 1 fwd stop, 2 rev start, 3 rev stop.
 * ref: _connection.h:94-140, 143-202, 205-267, 270-367 (the four split scorers) */
static void connect(po_node* nod, int j, int i, int kind, const po_training* t, int final) {
    const po_node* a = &nod[j]; po_node* b = &nod[i]; const po_node* n3;
    const int a_fs = a->strand == 1 && a->type == T_STOP;      /* forward stop  */
    const int a_rb = a->strand != 1 && a->type != T_STOP;      /* reverse start */
    const int a_fb = a->strand == 1 && a->type != T_STOP;      /* forward start */
    const int a_rs = a->strand != 1 && a->type == T_STOP;      /* reverse stop  */
    int left = a->ndx, right = b->ndx, ovlp = 0, maxfr = -1;
    double score = 0.0, mod = 0.0;

    if (a->traceb == -1 && (a_fs || a_rb)) return;             /* edge artifacts */

    switch (kind) {
    case 0:
        if (a_fs) { left += 2; if (left >= right) return; if (final) score = igm_same(a, b, t->st_wt); }
        else if (a_rb) { if (left >= right) return; if (final) score = igm_diff(t->st_wt); }
        break;
    case 1:
        if (a_fb) {
            if (b->stop_val >= a->ndx) return;
            right += 2;
            if (final) score = a->cscore + a->sscore; else mod = gc_bias_score(a, t);
        } else if (a_fs) {
            if (b->stop_val >= a->ndx) return;
            if (a->star_ptr[b->ndx % 3] == -1) return;
            n3 = &nod[a->star_ptr[b->ndx % 3]];
            left = n3->ndx; right += 2;
            if (final) score = n3->cscore + n3->sscore + igm(a, n3, t->st_wt); else mod = gc_bias_score(n3, t);
        }
        break;
    case 2:
        if (a_rs) {
            if (a->stop_val <= b->ndx) return;
            left -= 2;
            if (final) score = b->cscore + b->sscore; else mod = gc_bias_score(b, t);
        } else if (a_fs) {
            if (b->stop_val - 2 >= a->ndx + 2) return;
            ovlp = (a->ndx + 2) - (b->stop_val - 2) + 1;
            if (ovlp >= MAX_OPP_OVLP) return;
            if ((a->ndx - b->stop_val) >= (b->ndx - a->ndx + 3)) return;
            int bnd = a->traceb == -1 ? 0 : nod[a->traceb].ndx;
            if ((a->ndx - b->stop_val) >= (b->stop_val - 3 - bnd)) return;
            left = b->stop_val - 2;
            if (final) score = b->cscore + b->sscore + igm_diff(t->st_wt); else mod = gc_bias_score(b, t);
        }
        break;
    default:
        if (a_fs) {
            left += 2; right -= 2;
            if (left >= right) return;
            double maxval = 0.0, cur;
            for (int k = 0; k < 3; k++) {
                if (b->star_ptr[k] == -1) continue;
                n3 = &nod[b->star_ptr[k]];
                ovlp = left - n3->stop_val + 3;
                if (ovlp <= 0 || ovlp >= MAX_OPP_OVLP) continue;
                if (ovlp >= n3->ndx - left) continue;
                if (a->traceb == -1) continue;
                if (ovlp >= n3->stop_val - nod[a->traceb].ndx - 2) continue;
                cur = n3->cscore + n3->sscore + igm(n3, b, t->st_wt);
                if ((final && cur > maxval) || (!final && gc_bias_score(n3, t) > maxval)) { maxfr = k; maxval = cur; }
            }
            if (maxfr != -1) {
                n3 = &nod[b->star_ptr[maxfr]];
                if (final) score = n3->cscore + n3->sscore + igm(n3, b, t->st_wt); else mod = gc_bias_score(n3, t);
            } else if (final) score = igm_diff(t->st_wt);
        } else if (a_rb) {
            right -= 2;
            if (left >= right) return;
            if (final) score = igm_same(a, b, t->st_wt);
        } else if (a_rs) {
            if (a->stop_val <= b->ndx) return;
            if (b->star_ptr[a->ndx % 3] == -1) return;
            n3 = &nod[b->star_ptr[a->ndx % 3]];
            left -= 2; right = n3->ndx;
            if (final) score = n3->cscore + n3->sscore + igm(n3, b, t->st_wt); else mod = gc_bias_score(n3, t);
        }
    }
    if (!final) score = ((double)(right - left + 1 - ovlp * 2)) * mod;
    if (a->score + score >= b->score) { b->score = a->score + score; b->traceb = j; b->ov_mark = (int8_t)maxfr; }
}

/* ref: lib.pyx:1126-1162 (BaseConnectionScorer._index) */
static void index_nodes(po_ctx* c) {
    if (c->kcap < c->nn) {
        c->kcap = c->nn + 64;
        c->k_type = realloc(c->k_type, c->kcap); c->k_strand = realloc(c->k_strand, c->kcap);
        c->k_frame = realloc(c->k_frame, c->kcap); c->k_skip = realloc(c->k_skip, c->kcap);
    }
    for (int i = 0; i < c->nn; i++) {
        c->k_type[i] = c->nod[i].type;
        c->k_strand[i] = c->nod[i].strand == 1 ? 1 : 2;
        c->k_frame[i] = (uint8_t)(c->nod[i].ndx % 3);
        c->k_skip[i] = 0;
    }
}

/* ref: impl/generic.h:13-49 -- the six skip conditions, byte-wise (auto-vectorised by the compiler,
 * standing in for the SSE2/AVX2 instantiations of impl/template.h) */
static void skippable(const uint8_t* restrict st, const uint8_t* restrict ty, const uint8_t* restrict fr,
                      int lo, int i, uint8_t* restrict skip) {
    const uint8_t s2 = st[i], t2 = ty[i], f2 = fr[i];
    for (int j = lo; j < i; j++) {
        const uint8_t s1 = st[j], t1 = ty[j], f1 = fr[j];
        skip[j] = (uint8_t)(
              ((t1 != T_STOP) & (t2 != T_STOP) & (s1 == s2))
            | ((s1 == 1) & (t1 != T_STOP) & (s2 != 1))
            | ((s1 != 1) & (t1 == T_STOP) & (s2 == 1))
            | ((s1 != 1) & (t1 != T_STOP) & (s2 == 1) & (t2 == T_STOP))
            | ((s1 == s2) & (s1 == 1) & (t1 != T_STOP) & (t2 == T_STOP) & (f1 != f2))
            | ((s1 == s2) & (s1 != 1) & (t1 == T_STOP) & (t2 != T_STOP) & (f1 != f2)));
    }
}

/* ref: lib.pyx:1205-1311 (_score_connections, _find_max_index, _disentangle_overlaps,
 * _max_forward_pointers, _dynamic_programming) and _connection.h:386-408 */
/* the connection loop alone: node score / traceb / ov_mark before any traceback fix-up
 * (ref: lib.pyx:1205-1237 `_score_connections`) */
void po_dprog_raw(po_ctx* c, const po_training* t, int final) {
    const int nn = c->nn; po_node* nod = c->nod;
    if (nn == 0) return;
    index_nodes(c);
    for (int i = 0; i < nn; i++) { nod[i].score = 0; nod[i].traceb = -1; nod[i].tracef = -1; }
    for (int i = 0; i < nn; i++) {
        int lo = i < MAX_NODE_DIST ? 0 : i - MAX_NODE_DIST;
        int kind = 2 * (nod[i].strand != 1) + (nod[i].type == T_STOP);
        if ((kind == 2 || kind == 1) && nod[lo].ndx > nod[i].stop_val)
            while (lo > 0 && nod[lo].ndx != nod[i].stop_val) lo--;
        lo = lo < MAX_NODE_DIST ? 0 : lo - MAX_NODE_DIST;
        skippable(c->k_strand, c->k_type, c->k_frame, lo, i, c->k_skip);
        for (int j = lo; j < i; j++) if (!c->k_skip[j]) connect(nod, j, i, kind, t, final);
    }
}

/* ref: lib.pyx:1239-1251 (_find_max_index) */
int po_find_max_index(const po_ctx* c) {
    const po_node* nod = c->nod;
    int mx = -1; double best = -1.0;
    for (int i = c->nn - 1; i >= 0; i--) {
        if (nod[i].strand == 1 && nod[i].type != T_STOP) continue;
        if (nod[i].strand == -1 && nod[i].type == T_STOP) continue;
        if (nod[i].score > best) { best = nod[i].score; mx = i; }
    }
    return mx;
}

int po_dprog(po_ctx* c, const po_training* t, int final, int use_filter) {
    (void)use_filter;   /* the split scorers are only defined on filtered pairs */
    const int nn = c->nn; po_node* nod = c->nod;
    if (nn == 0) return -1;
    po_dprog_raw(c, t, final);
    int mx = po_find_max_index(c);
    if (mx < 0) return -1;   /* guarded; the reference would read nodes[-1] here */
    /* first pass: triple overlaps (ov_mark) */
    for (int p = mx; nod[p].traceb != -1; p = nod[p].traceb) {
        int nx = nod[p].traceb;
        if (nod[p].strand == -1 && nod[p].type == T_STOP && nod[nx].strand == 1 && nod[nx].type == T_STOP &&
            nod[p].ov_mark != -1 && nod[p].ndx > nod[nx].ndx) {
            int tmp = nod[p].star_ptr[nod[p].ov_mark], k = tmp;
            while (nod[k].ndx != nod[tmp].stop_val) k--;
            nod[p].traceb = tmp; nod[tmp].traceb = k; nod[k].ov_mark = -1; nod[k].traceb = nx;
        }
    }
    /* second pass: simple overlaps */
    for (int p = mx; nod[p].traceb != -1; p = nod[p].traceb) {
        int nx = nod[p].traceb;
        int p_rb = nod[p].strand == -1 && nod[p].type != T_STOP, p_fs = nod[p].strand == 1 && nod[p].type == T_STOP;
        int p_rs = nod[p].strand == -1 && nod[p].type == T_STOP;
        int n_fs = nod[nx].strand == 1 && nod[nx].type == T_STOP, n_rs = nod[nx].strand == -1 && nod[nx].type == T_STOP;
        if (p_rb && n_fs) {
            int k = p;
            while (nod[k].ndx != nod[p].stop_val) k--;
            nod[p].traceb = k; nod[k].traceb = nx;
        }
        if (p_fs && n_fs) { nod[p].traceb = nod[nx].star_ptr[nod[p].ndx % 3]; nod[nod[p].traceb].traceb = nx; }
        if (p_rs && n_rs) { nod[p].traceb = nod[p].star_ptr[nod[nx].ndx % 3]; nod[nod[p].traceb].traceb = nx; }
    }
    for (int p = mx; nod[p].traceb != -1; p = nod[p].traceb) nod[nod[p].traceb].tracef = p;
    return nod[mx].traceb == -1 ? -1 : mx;
}

/* Prodigal dprog.c eliminate_bad_genes (absent from checkout).
 * call sites ref: lib.pyx:5308, 5369 */
void po_eliminate_bad_genes(po_ctx* c, int ipath, const po_training* t) {
    po_node* nod = c->nod;
    if (ipath == -1) return;
    int p = ipath;
    while (nod[p].traceb != -1) p = nod[p].traceb;
    int head = p;
    for (; nod[p].tracef != -1; p = nod[p].tracef) {
        int f = nod[p].tracef;
        if (nod[p].strand == 1 && nod[p].type == T_STOP) nod[f].sscore += igm(&nod[p], &nod[f], t->st_wt);
        if (nod[p].strand == -1 && nod[p].type != T_STOP) nod[p].sscore += igm(&nod[p], &nod[f], t->st_wt);
    }
    for (p = head; nod[p].tracef != -1; p = nod[p].tracef) {
        int f = nod[p].tracef;
        if (nod[p].strand == 1 && nod[p].type != T_STOP && nod[p].cscore + nod[p].sscore < 0) { nod[p].elim = 1; nod[f].elim = 1; }
        if (nod[p].strand == -1 && nod[p].type == T_STOP && nod[f].cscore + nod[f].sscore < 0) { nod[p].elim = 1; nod[f].elim = 1; }
    }
}

/* ------------------------------------------------------------------- genes */

static void push_gene(po_ctx* c, int b, int e, int s, int t) {
    if (c->ng == c->gcap) { c->gcap = c->gcap ? c->gcap * 2 : 64; c->gen = realloc(c->gen, sizeof(po_gene) * c->gcap); }
    po_gene* g = &c->gen[c->ng++]; g->begin = b; g->end = e; g->start_ndx = s; g->stop_ndx = t;
}

/* ref: lib.pyx:3231-3270 (Genes._extract) */
int po_extract_genes(po_ctx* c, int ipath) {
    po_node* nod = c->nod; int p = ipath, b = 0, e = 0, s = 0, t = 0;
    c->ng = 0; c->ipath = ipath;
    if (p == -1) return 0;
    while (nod[p].traceb != -1) p = nod[p].traceb;
    for (; p != -1; p = nod[p].tracef) {
        if (nod[p].elim == 1) continue;
        if (nod[p].strand == 1) {
            if (nod[p].type != T_STOP) { b = nod[p].ndx + 1; s = p; }
            else { e = nod[p].ndx + 3; t = p; push_gene(c, b, e, s, t); }
        } else {
            if (nod[p].type != T_STOP) { e = nod[p].ndx + 1; s = p; push_gene(c, b, e, s, t); }
            else { b = nod[p].ndx - 1; t = p; }
        }
    }
    return c->ng;
}

/* ref: lib.pyx:3272-3401 (Genes._tweak_final_starts) */
void po_tweak_final_starts(po_ctx* c, const po_training* t, int maxov) {
    po_node* nod = c->nod; po_gene* g = c->gen; const int nn = c->nn, ng = c->ng; const double w = t->st_wt;
    for (int i = 0; i < ng; i++) {
        int ndx = g[i].start_ndx;
        double sc = nod[ndx].sscore + nod[ndx].cscore, ig = 0.0;
        int prev_fwd = i > 0 && nod[g[i - 1].start_ndx].strand == 1, prev_rev = i > 0 && nod[g[i - 1].start_ndx].strand == -1;
        int next_fwd = i < ng - 1 && nod[g[i + 1].start_ndx].strand == 1, next_rev = i < ng - 1 && nod[g[i + 1].start_ndx].strand == -1;
        if (nod[ndx].strand == 1 && prev_fwd) ig = igm_same(&nod[g[i - 1].stop_ndx], &nod[ndx], w);
        if (nod[ndx].strand == 1 && prev_rev) ig = igm_diff(w);
        if (nod[ndx].strand == -1 && next_fwd) ig = igm_diff(w);
        if (nod[ndx].strand == -1 && next_rev) ig = igm_same(&nod[ndx], &nod[g[i + 1].stop_ndx], w);

        int mi[2] = {-1, -1}; double ms[2] = {0, 0}, mg[2] = {0, 0};
        for (int j = ndx - 100; j < ndx + 100; j++) {
            if (j < 0 || j >= nn || j == ndx) continue;
            if (nod[j].type == T_STOP || nod[j].stop_val != nod[ndx].stop_val) continue;
            double tg = 0.0;
            if (nod[j].strand == 1 && prev_fwd) {
                if (nod[g[i - 1].stop_ndx].ndx - nod[j].ndx > maxov) continue;
                tg = igm_same(&nod[g[i - 1].stop_ndx], &nod[j], w);
            }
            if (nod[j].strand == 1 && prev_rev) {
                if (nod[g[i - 1].start_ndx].ndx - nod[j].ndx >= 0) continue;
                tg = igm_diff(w);
            }
            if (nod[j].strand == -1 && next_fwd) {
                if (nod[j].ndx - nod[g[i + 1].start_ndx].ndx >= 0) continue;
                tg = igm_diff(w);
            }
            if (nod[j].strand == -1 && next_rev) {
                if (nod[j].ndx - nod[g[i + 1].stop_ndx].ndx > maxov) continue;
                tg = igm_same(&nod[j], &nod[g[i + 1].stop_ndx], w);
            }
            double cs = nod[j].cscore + nod[j].sscore;
            if (mi[0] == -1) { mi[0] = j; ms[0] = cs; mg[0] = tg; }
            else if (cs + tg > ms[0]) { mi[1] = mi[0]; ms[1] = ms[0]; mg[1] = mg[0]; mi[0] = j; ms[0] = cs; mg[0] = tg; }
            else if (mi[1] == -1 || cs + tg > ms[1]) { mi[1] = j; ms[1] = cs; mg[1] = tg; }
        }
        for (int k = 0; k < 2; k++) {
            int m = mi[k];
            if (m == -1) continue;
            if (nod[m].tscore < nod[ndx].tscore && ms[k] - nod[m].tscore >= sc - nod[ndx].tscore + w &&
                nod[m].rscore > nod[ndx].rscore && nod[m].uscore > nod[ndx].uscore &&
                nod[m].cscore > nod[ndx].cscore && abs(nod[m].ndx - nod[ndx].ndx) > 15) {
                ms[k] += nod[ndx].tscore - nod[m].tscore;
            } else if (abs(nod[m].ndx - nod[ndx].ndx) <= 15 &&
                       nod[m].rscore + nod[m].tscore > nod[ndx].rscore + nod[ndx].tscore &&
                       nod[ndx].edge == 0 && nod[m].edge == 0) {
                if (nod[ndx].cscore > nod[m].cscore) ms[k] += nod[ndx].cscore - nod[m].cscore;
                if (nod[ndx].uscore > nod[m].uscore) ms[k] += nod[ndx].uscore - nod[m].uscore;
                if (ig > mg[k]) ms[k] += ig - mg[k];
            } else ms[k] = -1000.0;
        }
        int pick = -1;
        for (int k = 0; k < 2; k++) {
            if (mi[k] == -1) continue;
            if (pick == -1 && ms[k] + mg[k] > sc + ig) pick = k;
            else if (pick >= 0 && ms[k] + mg[k] > ms[pick] + mg[pick]) pick = k;
        }
        if (pick != -1 && nod[mi[pick]].strand == 1) { g[i].start_ndx = mi[pick]; g[i].begin = nod[mi[pick]].ndx + 1; }
        else if (pick != -1 && nod[mi[pick]].strand == -1) { g[i].start_ndx = mi[pick]; g[i].end = nod[mi[pick]].ndx + 1; }
    }
}

/* ----------------------------------------------------------------- drivers */

/* ref: lib.pyx:5281-5315 (GeneFinder._find_genes_single) */
int po_find_genes_single(po_ctx* c, const po_training* t, const po_params* p) {
    po_extract(c, t->trans_table, p);
    po_sort(c);
    po_reset_scores(c);
    po_score_nodes(c, t, p->closed, 0);
    po_overlapping_starts(c, t, 1, p->max_overlap);
    int ipath = po_dprog(c, t, 1, 1);
    if (c->nn > 0) po_eliminate_bad_genes(c, ipath, t);
    c->path_score = (ipath >= 0) ? c->nod[ipath].score : 0.0;
    po_extract_genes(c, ipath);
    po_tweak_final_starts(c, t, p->max_overlap);
    return c->ng;
}

/* ref: lib.pyx:5317-5396 (GeneFinder._find_genes_meta); returns the winning bin or -1 */
int po_find_genes_meta(po_ctx* c, const po_training* const* bins, int nbins, const po_params* p) {
    int tt = -1, phase = -1; double best = -100.0;
    double low = fmin(0.65, 0.88495 * c->gc - 0.0102337);
    double high = fmax(0.35, 0.86596 * c->gc + 0.1131991);
    c->ng = 0; c->ipath = -1;
    for (int b = 0; b < nbins; b++) {
        const po_training* t = bins[b];
        if (t->gc < low || t->gc > high) continue;
        if (t->trans_table != tt) { tt = t->trans_table; po_extract(c, tt, p); po_sort(c); }
        po_reset_scores(c);
        po_score_nodes(c, t, p->closed, 1);
        po_overlapping_starts(c, t, 1, p->max_overlap);
        int ipath = po_dprog(c, t, 1, 1);
        if (c->nn > 0 && ipath >= 0 && c->nod[ipath].score > best) {
            phase = b; best = c->nod[ipath].score;
            po_eliminate_bad_genes(c, ipath, t);
            po_extract_genes(c, ipath);
            po_tweak_final_starts(c, t, p->max_overlap);
        }
    }
    c->path_score = best;
    if (phase >= 0) {
        const po_training* t = bins[phase];
        po_extract(c, t->trans_table, p); po_sort(c);
        po_reset_scores(c);
        po_score_nodes(c, t, p->closed, 1);
    }
    return phase;
}

/* ---------------------------------------------------------------- training */

static int max_fr(int a, int b, int c3) { return a > b ? (a > c3 ? 0 : 2) : (b > c3 ? 1 : 2); } /* Prodigal sequence.c */

/* ref: lib.pyx:724-768 (Sequence._max_gc_frame_plot) */
static int* gc_frame_plot(const po_ctx* c) {
    const int n = c->slen, half = GC_WINDOW / 2;
    int* fwd = calloc(n, sizeof(int)); int* bwd = calloc(n, sizeof(int));
    int* tot = calloc(n, sizeof(int)); int* gp = malloc(sizeof(int) * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) gp[i] = -1;
    for (int i = 0; i < n; i++) {
        int g = is_gc_fwd(c, i), gr = is_gc_fwd(c, n - 1 - i);
        fwd[i] = (i >= 3 ? fwd[i - 3] : 0) + g;
        bwd[n - i - 1] = (i >= 3 ? bwd[n - i + 2] : 0) + gr;
    }
    for (int i = 0; i < n; i++) {
        tot[i] = fwd[i] + bwd[i] - is_gc_fwd(c, i);
        if (i >= half) tot[i] -= fwd[i - half];
        if (i + half < n) tot[i] -= bwd[i + half];
    }
    for (int i = 0; i < n - 2; i += 3) {
        int w = max_fr(tot[i], tot[i + 1], tot[i + 2]);
        gp[i] = gp[i + 1] = gp[i + 2] = w;
    }
    free(fwd); free(bwd); free(tot);
    return gp;
}

/* Prodigal node.c record_gc_bias (absent). call site ref: lib.pyx:5261 */
static void record_gc_bias(const int* gc, po_ctx* c, po_training* t) {
    const int nn = c->nn; po_node* nod = c->nod;
    int ctr[3][3], last[3] = {0, 0, 0};
    if (nn == 0) return;
    memset(ctr, 0, sizeof ctr);
    for (int i = nn - 1; i >= 0; i--) {
        if (nod[i].strand != 1) continue;
        int fr = nod[i].ndx % 3, fm = 3 - fr;
        if (nod[i].type == T_STOP) {
            ctr[fr][0] = ctr[fr][1] = ctr[fr][2] = 0;
            last[fr] = nod[i].ndx;
            ctr[fr][(gc[nod[i].ndx] + fm) % 3] = 1;
        } else {
            for (int j = last[fr] - 3; j >= nod[i].ndx; j -= 3) ctr[fr][(gc[j] + fm) % 3]++;
            nod[i].gc_bias = (uint8_t)max_fr(ctr[fr][0], ctr[fr][1], ctr[fr][2]);
            for (int j = 0; j < 3; j++) {
                nod[i].gc_score[j] = 3.0 * ctr[fr][j];
                nod[i].gc_score[j] /= 1.0 * (nod[i].stop_val - nod[i].ndx + 3);
            }
            last[fr] = nod[i].ndx;
        }
    }
    memset(ctr, 0, sizeof ctr);
    for (int i = 0; i < nn; i++) {
        if (nod[i].strand != -1) continue;
        int fr = nod[i].ndx % 3, fm = fr;
        if (nod[i].type == T_STOP) {
            ctr[fr][0] = ctr[fr][1] = ctr[fr][2] = 0;
            last[fr] = nod[i].ndx;
            ctr[fr][((3 - gc[nod[i].ndx]) + fm) % 3] = 1;
        } else {
            for (int j = last[fr] + 3; j <= nod[i].ndx; j += 3) ctr[fr][((3 - gc[j]) + fm) % 3]++;
            nod[i].gc_bias = (uint8_t)max_fr(ctr[fr][0], ctr[fr][1], ctr[fr][2]);
            for (int j = 0; j < 3; j++) {
                nod[i].gc_score[j] = 3.0 * ctr[fr][j];
                nod[i].gc_score[j] /= 1.0 * (nod[i].ndx - nod[i].stop_val + 3);
            }
            last[fr] = nod[i].ndx;
        }
    }
    t->bias[0] = t->bias[1] = t->bias[2] = 0.0;
    for (int i = 0; i < nn; i++) {
        if (nod[i].type == T_STOP) continue;
        int len = abs(nod[i].stop_val - nod[i].ndx) + 1;
        t->bias[nod[i].gc_bias] += (nod[i].gc_score[nod[i].gc_bias] * len) / 1000.0;
    }
    double tot = t->bias[0] + t->bias[1] + t->bias[2];
    for (int i = 0; i < 3; i++) t->bias[i] *= (3.0 / tot);
}

/* ref: lib.pyx:4284-4358 (TrainingInfo._calc_dicodon_gene) */
static void calc_dicodon_gene(const po_ctx* c, po_training* t, int ipath) {
    static int counts[4096]; static double prob[4096], bg[4096];
    const po_node* nod = c->nod; const int slen = c->slen;
    int glob = 0, in_gene = 0, left = -1, right = -1;
    memset(counts, 0, sizeof counts);
    for (int i = 0; i < slen - 5; i++) { counts[mer_ndx(c, i, 6, 1)]++; counts[mer_ndx(c, i, 6, -1)]++; glob += 2; }
    for (int i = 0; i < 4096; i++) bg[i] = (double)counts[i] / (double)glob;
    glob = 0; memset(counts, 0, sizeof counts);
    for (int p = ipath; p != -1; p = nod[p].traceb) {
        if (nod[p].strand == 1) {
            if (nod[p].type == T_STOP) { in_gene = 1; right = nod[p].ndx + 2; }
            else if (in_gene == 1) {
                left = nod[p].ndx;
                for (int i = left; i < right - 5; i += 3) { counts[mer_ndx(c, i, 6, 1)]++; glob++; }
                in_gene = 0;
            }
        } else {
            if (nod[p].type != T_STOP) { in_gene = -1; left = slen - nod[p].ndx - 1; }
            else if (in_gene == -1) {
                right = slen - nod[p].ndx + 1;
                for (int i = left; i < right - 5; i += 3) { counts[mer_ndx(c, i, 6, -1)]++; glob++; }
                in_gene = 0;
            }
        }
    }
    for (int i = 0; i < 4096; i++) {
        prob[i] = (double)counts[i] / (double)glob;
        if (prob[i] == 0 && bg[i] != 0) t->gene_dc[i] = -5.0;
        else if (bg[i] == 0) t->gene_dc[i] = 0.0;
        else t->gene_dc[i] = log(prob[i] / bg[i]);
        if (t->gene_dc[i] > 5.0) t->gene_dc[i] = 5.0;
        else if (t->gene_dc[i] < -5.0) t->gene_dc[i] = -5.0;
    }
}

/* ref: lib.pyx:4360-4389 (TrainingInfo._count_upstream_composition) */
static void count_upstream(const po_ctx* c, po_training* t, int pos, int strand) {
    int k = 0;
    for (int pass = 0; pass < 2; pass++) {
        int lo = pass ? 15 : 1, hi = pass ? 45 : 3;
        for (int j = lo; j < hi; j++, k++) {
            if (strand == 1) { if (pos >= j) t->ups_comp[k][c->dig[pos - j] & 3] += 1; }
            else { if (pos + j < c->slen) t->ups_comp[k][comp2(c->dig[pos + j]) & 3] += 1; }
        }
    }
}

static void ups_to_log(po_training* t) {   /* ref: lib.pyx:4571-4599 / 4797-4827 */
    for (int i = 0; i < 32; i++) {
        double sum = 0.0;
        for (int j = 0; j < 4; j++) sum += t->ups_comp[i][j];
        if (sum == 0.0) { for (int j = 0; j < 4; j++) t->ups_comp[i][j] = 0.0; continue; }
        for (int j = 0; j < 4; j++) {
            double* u = &t->ups_comp[i][j];
            *u /= sum;
            int at = (j == 0 || j == 3);
            if (t->gc <= 0.1) *u = log(*u * 2.0 / (at ? 0.90 : 0.10));
            else if (t->gc >= 0.9) *u = log(*u * 2.0 / (at ? 0.10 : 0.90));
            else *u = at ? log(*u * 2.0 / (1.0 - t->gc)) : log(*u * 2.0 / t->gc);
            if (*u > 4.0) *u = 4.0;
            if (*u < -4.0) *u = -4.0;
        }
    }
}

static void log_odds3(double* real, const double* bgv, double* out, int n) {
    double sum = 0.0;
    for (int j = 0; j < n; j++) sum += real[j];
    if (sum == 0.0) { for (int j = 0; j < n; j++) out[j] = 0.0; return; }
    for (int j = 0; j < n; j++) {
        real[j] /= sum;
        out[j] = bgv[j] != 0 ? log(real[j] / bgv[j]) : -4.0;
        if (out[j] > 4.0) out[j] = 4.0; else if (out[j] < -4.0) out[j] = -4.0;
    }
}

static int pick_rbs(const po_training* t, const po_node* n) {   /* ref: lib.pyx:4441-4448 */
    double w0 = t->rbs_wt[n->rbs[0]], w1 = t->rbs_wt[n->rbs[1]];
    if (w0 > w1 + 1.0 || n->rbs[1] == 0) return n->rbs[0];
    if (w0 < w1 - 1.0 || n->rbs[0] == 0) return n->rbs[1];
    return n->rbs[0] > n->rbs[1] ? n->rbs[0] : n->rbs[1];
}

/* ref: lib.pyx:4391-4599 (TrainingInfo._train_starts_sd) */
static void train_starts_sd(po_ctx* c, po_training* t) {
    const int nn = c->nn; const po_node* nod = c->nod; const double wt = t->st_wt;
    double rbg[28], rreal[28], best[3], tbg[3] = {0, 0, 0}, treal[3], sum, sthresh = 35.0;
    int rbs[3], type[3], bndx[3];
    memset(t->type_wt, 0, sizeof t->type_wt); memset(t->rbs_wt, 0, sizeof t->rbs_wt);
    memset(t->ups_comp, 0, sizeof t->ups_comp);
    for (int i = 0; i < nn; i++) if (nod[i].type != T_STOP) tbg[nod[i].type] += 1.0;
    sum = 0.0; for (int i = 0; i < 3; i++) sum += tbg[i];
    for (int i = 0; i < 3; i++) tbg[i] /= sum;

    for (int it = 0; it < 10; it++) {
        memset(rbg, 0, sizeof rbg);
        for (int j = 0; j < nn; j++) {
            if (nod[j].type == T_STOP || nod[j].edge) continue;
            rbg[pick_rbs(t, &nod[j])] += 1.0;
        }
        sum = 0.0; for (int j = 0; j < 28; j++) sum += rbg[j];
        for (int j = 0; j < 28; j++) rbg[j] /= sum;
        memset(rreal, 0, sizeof rreal); treal[0] = treal[1] = treal[2] = 0.0;
        for (int dir = 0; dir < 2; dir++) {
            const int strand = dir == 0 ? 1 : -1;
            for (int j = 0; j < 3; j++) { best[j] = 0.0; bndx[j] = -1; rbs[j] = 0; type[j] = 0; }
            for (int k = 0; k < nn; k++) {
                int j = dir == 0 ? k : nn - 1 - k;
                if (nod[j].type != T_STOP && nod[j].edge) continue;
                if (nod[j].strand != strand) continue;
                int ph = nod[j].ndx % 3;
                if (nod[j].type == T_STOP) {
                    if (best[ph] >= sthresh && nod[bndx[ph]].ndx % 3 == ph) {
                        rreal[rbs[ph]] += 1.0; treal[type[ph]] += 1.0;
                        if (it == 9) count_upstream(c, t, nod[bndx[ph]].ndx, strand);
                    }
                    best[ph] = 0.0; bndx[ph] = -1; rbs[ph] = 0; type[ph] = 0;
                } else {
                    int mr = pick_rbs(t, &nod[j]);
                    double v = nod[j].cscore + wt * t->rbs_wt[mr] + wt * t->type_wt[nod[j].type];
                    if (v >= best[ph]) { best[ph] = v; bndx[ph] = j; type[ph] = nod[j].type; rbs[ph] = mr; }
                }
            }
        }
        log_odds3(rreal, rbg, t->rbs_wt, 28);
        sum = 0.0; for (int j = 0; j < 3; j++) sum += treal[j];
        log_odds3(treal, tbg, t->type_wt, 3);
        if (sum * 2000.0 <= nn) sthresh /= 2.0;
    }
    ups_to_log(t);
}

/* Prodigal node.c determine_sd_usage (absent). call site ref: lib.pyx:5276 */
static void determine_sd_usage(po_training* t) {
    t->uses_sd = 1;
    if (t->rbs_wt[0] >= 0.0) t->uses_sd = 0;
    if (t->rbs_wt[16] < 1.0 && t->rbs_wt[13] < 1.0 && t->rbs_wt[15] < 1.0 &&
        (t->rbs_wt[0] >= -0.5 || (t->rbs_wt[22] < 2.0 && t->rbs_wt[24] < 2.0 && t->rbs_wt[27] < 2.0))) t->uses_sd = 0;
}

typedef double mot_tab[4][4096];

/* ref: lib.pyx:4225-4282 (TrainingInfo._update_motif_counts) */
static void update_motif_counts(const po_ctx* c, mot_tab* cnt, double* zero, const po_node* n, int stage) {
    if (n->type == T_STOP || n->edge == 1) return;
    if (n->mot_len == 0) { *zero += 1.0; return; }
    int start = n->strand == 1 ? n->ndx : c->slen - 1 - n->ndx;
    if (stage == 0) {
        for (int i = 3; i >= 0; i--)
            for (int j = start - 18 - i; j < start - 5 - i; j++) {
                if (j < 0) continue;
                int mer = mer_ndx(c, j, i + 3, n->strand);
                for (int k = 0; k < 4; k++) cnt[i][k][mer] += 1.0;
            }
    } else if (stage == 1) {
        cnt[n->mot_len - 3][n->mot_spacendx][n->mot_ndx] += 1.0;
        for (int i = 0; i < n->mot_len - 3; i++)
            for (int j = start - n->mot_spacer - n->mot_len; j < start - n->mot_spacer - i - 2; j++) {
                if (j < 0) continue;
                int si;
                if (j <= start - 16 - i) si = 3;
                else if (j <= start - 14 - i) si = 2;
                else if (j >= start - 7 - i) si = 1;
                else si = 0;
                cnt[i][si][mer_ndx(c, j, i + 3, n->strand)] += 1.0;
            }
    } else if (stage == 2) cnt[n->mot_len - 3][n->mot_spacendx][n->mot_ndx] += 1.0;
}

/* Prodigal node.c build_coverage_map (absent). call site ref: lib.pyx:4735 */
static void build_coverage_map(mot_tab* real, int (*good)[4][4096], double ng) {
    const double thresh = 0.2;
    memset(good, 0, sizeof(int) * 4 * 4 * 4096);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 64; j++)
        if (real[0][i][j] / ng >= thresh) for (int k = 0; k < 4; k++) good[0][k][j] = 1;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 256; j++) {
        int d0 = (j & 252) >> 2, d1 = j & 63;
        if (good[0][i][d0] == 0 || good[0][i][d1] == 0) continue;
        good[1][i][j] = 1;
    }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 1024; j++) {
        int d0 = (j & 1008) >> 4, d1 = (j & 252) >> 2, d2 = j & 63;
        if (good[0][i][d0] == 0 || good[0][i][d1] == 0 || good[0][i][d2] == 0) continue;
        good[2][i][j] = 1;
        int tmp = j;
        for (int k = 0; k <= 16; k += 16) {
            tmp ^= k;
            for (int l = 0; l <= 32; l += 32) { tmp ^= l; if (good[2][i][tmp] == 0) good[2][i][tmp] = 2; }
        }
    }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4096; j++) {
        int d0 = (j & 4092) >> 2, d1 = j & 1023;
        if (good[2][i][d0] == 0 || good[2][i][d1] == 0) continue;
        good[3][i][j] = (good[2][i][d0] == 1 && good[2][i][d1] == 1) ? 1 : 2;
    }
}

/* ref: lib.pyx:4601-4827 (TrainingInfo._train_starts_nonsd) */
static void train_starts_nonsd(po_ctx* c, po_training* t) {
    const int nn = c->nn; po_node* nod = c->nod; const double wt = t->st_wt;
    mot_tab* mbg = calloc(4, sizeof(mot_tab)); mot_tab* mreal = calloc(4, sizeof(mot_tab));
    int (*mgood)[4][4096] = calloc(4, sizeof(int[4][4096]));
    double best[3], tbg[3] = {0, 0, 0}, treal[3], sum, ngenes, zbg, zreal, sthresh = 35.0;
    int bndx[3];
    memset(t->ups_comp, 0, sizeof t->ups_comp);
    memset(t->type_wt, 0, sizeof t->type_wt);
    for (int i = 0; i < nn; i++) if (nod[i].type != T_STOP) tbg[nod[i].type] += 1.0;
    sum = 0.0; for (int i = 0; i < 3; i++) sum += tbg[i];
    for (int i = 0; i < 3; i++) tbg[i] /= sum;

    for (int it = 0; it < 20; it++) {
        int stage = it < 4 ? 0 : (it < 12 ? 1 : 2);
        memset(mbg, 0, 4 * sizeof(mot_tab)); zbg = 0.0;
        for (int j = 0; j < nn; j++) {
            if (nod[j].type == T_STOP || nod[j].edge) continue;
            best_upstream_motif(c, &nod[j], t, stage);
            update_motif_counts(c, mbg, &zbg, &nod[j], stage);
        }
        sum = 0.0;
        for (int j = 0; j < 4; j++) for (int k = 0; k < 4; k++) for (int l = 0; l < 4096; l++) sum += mbg[j][k][l];
        sum += zbg;
        for (int j = 0; j < 4; j++) for (int k = 0; k < 4; k++) for (int l = 0; l < 4096; l++) mbg[j][k][l] /= sum;
        zbg /= sum;
        memset(mreal, 0, 4 * sizeof(mot_tab)); zreal = 0.0;
        treal[0] = treal[1] = treal[2] = 0.0; ngenes = 0.0;
        for (int dir = 0; dir < 2; dir++) {
            const int strand = dir == 0 ? 1 : -1;
            for (int j = 0; j < 3; j++) { best[j] = 0.0; bndx[j] = -1; }
            for (int k = 0; k < nn; k++) {
                int j = dir == 0 ? k : nn - 1 - k;
                if (nod[j].type != T_STOP && nod[j].edge) continue;
                if (nod[j].strand != strand) continue;
                int fr = nod[j].ndx % 3;
                if (nod[j].type == T_STOP) {
                    if (best[fr] >= sthresh) {
                        ngenes += 1.0; treal[nod[bndx[fr]].type] += 1.0;
                        update_motif_counts(c, mreal, &zreal, &nod[bndx[fr]], stage);
                        if (it == 19) count_upstream(c, t, nod[bndx[fr]].ndx, strand);
                    }
                    best[fr] = 0.0; bndx[fr] = -1;
                } else {
                    double v = nod[j].cscore + wt * nod[j].mot_score + wt * t->type_wt[nod[j].type];
                    if (v >= best[fr]) { best[fr] = v; bndx[fr] = j; }
                }
            }
        }
        if (stage < 2) build_coverage_map(mreal, mgood, ngenes);
        sum = 0.0;
        for (int j = 0; j < 4; j++) for (int k = 0; k < 4; k++) for (int l = 0; l < 4096; l++) sum += mreal[j][k][l];
        sum += zreal;
        if (sum == 0.0) {
            memset(t->mot_wt, 0, sizeof t->mot_wt); t->no_mot = 0.0;
        } else {
            for (int j = 0; j < 4; j++) for (int k = 0; k < 4; k++) for (int l = 0; l < 4096; l++) {
                if (mgood[j][k][l] == 0) { zreal += mreal[j][k][l]; zbg += mreal[j][k][l]; mreal[j][k][l] = 0.0; mbg[j][k][l] = 0.0; }
                mreal[j][k][l] /= sum;
                double* w = &t->mot_wt[j][k][l];
                *w = mbg[j][k][l] != 0 ? log(mreal[j][k][l] / mbg[j][k][l]) : -4.0;
                if (*w > 4.0) *w = 4.0; else if (*w < -4.0) *w = -4.0;
            }
        }
        zreal /= sum;
        t->no_mot = zbg != 0 ? log(zreal / zbg) : -4.0;
        if (t->no_mot > 4.0) t->no_mot = 4.0; else if (t->no_mot < -4.0) t->no_mot = -4.0;
        sum = 0.0; for (int j = 0; j < 3; j++) sum += treal[j];
        log_odds3(treal, tbg, t->type_wt, 3);
        if (sum * 2000.0 <= nn) sthresh /= 2.0;
    }
    ups_to_log(t);
    free(mbg); free(mreal); free(mgood);
}

/* ref: lib.pyx:5236-5279 (GeneFinder._train) and 3955-4003 (TrainingInfo.__init__) */
/* GC frame plot + record_gc_bias on the current nodes (fills gc_score / gc_bias of the nodes and t->bias) */
void po_record_gc_bias(po_ctx* c, po_training* t) {
    int* gcf = gc_frame_plot(c);
    record_gc_bias(gcf, c, t);
    free(gcf);
}

/* `upto` stops the training early, for step-by-step checks of other implementations:
 * 1 after the GC frame bias, 2 after the hexamer statistics, 3 after the Shine-Dalgarno start training, else all of it */
int po_train_upto(po_ctx* c, po_training* t, const po_params* p, int force_nonsd, double start_weight, int tt, int upto) {
    memset(t, 0, sizeof *t);
    t->gc = c->gc; t->trans_table = tt; t->st_wt = start_weight; t->uses_sd = 1;
    po_extract(c, tt, p);
    po_sort(c);
    int* gcf = gc_frame_plot(c);
    record_gc_bias(gcf, c, t);
    free(gcf);
    if (upto == 1) return 0;
    po_overlapping_starts(c, t, 0, p->max_overlap);
    int ipath = po_dprog(c, t, 0, 1);
    calc_dicodon_gene(c, t, ipath);
    if (upto == 2) return 0;
    raw_coding_score(c, t);
    rbs_score(c, t);
    train_starts_sd(c, t);
    if (force_nonsd) t->uses_sd = 0; else determine_sd_usage(t);
    if (upto == 3) return 0;
    if (!t->uses_sd) train_starts_nonsd(c, t);
    return 0;
}
int po_train(po_ctx* c, po_training* t, const po_params* p, int force_nonsd, double start_weight, int tt) {
    return po_train_upto(c, t, p, force_nonsd, start_weight, tt, 0);
}

/* ---- all-core baseline driver (bench.py `cpu_baseline.all_cores`) ---------------------------------------------------------
 * pyrodigal's own way of using every core is a pool of threads mapping find_genes over the records (ref: cli.py:289-302): the
 * threads share the models (one copy of the 9 MB of tables in the caches) and every call builds private state.  This is that
 * pool in C -- no interpreter lock, no per-call mmap / munmap (glibc is told to keep freed blocks: a process-wide munmap per
 * contig costs a TLB shootdown on every core) -- so the figure says what the CPU path can do, not what a Python harness costs. */
#include <malloc.h>
#include <pthread.h>
#include <sched.h>
#include <time.h>

typedef struct {
    const char* const* seqs; const int64_t* lens; int n;
    const po_training* const* bins; int nbins; const po_params* p;
    int next; int64_t genes; pthread_mutex_t mu;
    int total;                  /* calls to make: the list is gone over as often as that takes (or only its head) */
    const int* cpus; int ncpus; /* thread t runs on logical CPU cpus[t % ncpus] (or anywhere) */
    int started;                /* threads that took their number */
    double cpu_seconds;         /* CPU time the threads spent, summed */
} po_pool_job;

static void* po_pool_worker(void* arg) {
    po_pool_job* J = (po_pool_job*)arg;
    const int t = __atomic_fetch_add(&J->started, 1, __ATOMIC_RELAXED);
    if (J->cpus != NULL && J->ncpus > 0) {
        cpu_set_t set; CPU_ZERO(&set); CPU_SET(J->cpus[t % J->ncpus], &set);
        pthread_setaffinity_np(pthread_self(), sizeof set, &set);
    }
    int64_t genes = 0;
    for (;;) {
        const int k = __atomic_fetch_add(&J->next, 1, __ATOMIC_RELAXED);
        if (k >= J->total) break;
        po_ctx* c = po_new(J->seqs[k % J->n], J->lens[k % J->n], 0, 50);
        po_find_genes_meta(c, J->bins, J->nbins, J->p);
        genes += po_num_genes(c);
        po_free(c);
    }
    struct timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    pthread_mutex_lock(&J->mu); J->genes += genes; J->cpu_seconds += (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; pthread_mutex_unlock(&J->mu);
    return NULL;
}

/* `total` calls (the n sequences, gone over as often as that takes) on `threads` threads, thread t pinned to cpus[t % ncpus] when a
 * list is given; *cpu_seconds receives the CPU time the threads used.  Returns the number of genes found, -1 on failure. */
int64_t po_find_genes_meta_pool_pinned(const char* const* seqs, const int64_t* lens, int n, const po_training* const* bins, int nbins,
                                       const po_params* p, int threads, int total, const int* cpus, int ncpus, double* cpu_seconds) {
    if (threads < 1) threads = 1;
    if (n < 1) return 0;
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    po_pool_job J = {seqs, lens, n, bins, nbins, p, 0, 0, PTHREAD_MUTEX_INITIALIZER, total > 0 ? total : n, cpus, ncpus, 0, 0.0};
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    if (!th) return -1;
    int started = 0;
    for (int t = 0; t < threads; t++) { if (pthread_create(&th[t], NULL, po_pool_worker, &J) != 0) break; started++; }
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    free(th);
    if (cpu_seconds) *cpu_seconds = J.cpu_seconds;
    return started > 0 ? J.genes : -1;
}

/* meta-mode gene finding of n sequences on `threads` threads; returns the number of genes found, -1 on failure */
int64_t po_find_genes_meta_pool(const char* const* seqs, const int64_t* lens, int n, const po_training* const* bins, int nbins,
                                const po_params* p, int threads) {
    return po_find_genes_meta_pool_pinned(seqs, lens, n, bins, nbins, p, threads, n, NULL, 0, NULL);
}
