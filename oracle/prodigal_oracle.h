/*
 * prodigal_oracle.h -- CPU restatement of the pyrodigal / Prodigal gene-finding
 * path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (pyrodigal_amd/, include/) never links,
 * imports or calls it.
 *
 * Every function cites the reference location it restates (paths relative to
 * /root/reference).  The vendored Prodigal C sources are absent from the
 * reference checkout (vendor/Prodigal is an un-vendored submodule pinned at
 * hyattpd/Prodigal v2.6.3+c1e2d36), so helpers that live only there
 * (eliminate_bad_genes, reset_node_scores, compare_nodes, record_gc_bias,
 * determine_sd_usage, build_coverage_map) restate the published v2.6.3
 * algorithm and are pinned through the reference's own golden fixtures
 * (tests/golden/, see tests/test_oracle_golden.py).
 */
#ifndef PRODIGAL_ORACLE_H
#define PRODIGAL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/pyrodigal/prodigal/training.pxd:3-14 -- byte layout of the reference's
 * `struct _training` (558 392 bytes), so its TrainingInfo dumps load as-is. */
typedef struct po_training {
    double  gc;
    int32_t trans_table;
    int32_t _pad0;
    double  st_wt;
    double  bias[3];
    double  type_wt[3];
    int32_t uses_sd;
    int32_t _pad1;
    double  rbs_wt[28];
    double  ups_comp[32][4];
    double  mot_wt[4][4][4096];
    double  no_mot;
    double  gene_dc[4096];
} po_training;

/* Same information as src/Prodigal/node.h:40-76 (own field order). */
typedef struct po_node {
    double  cscore, uscore, tscore, rscore, sscore, score;
    double  gc_score[3];
    double  mot_score;
    float   gc_cont;
    int32_t star_ptr[3];
    int32_t traceb, tracef, ndx, stop_val;
    int32_t mot_ndx;
    int8_t  ov_mark, strand;
    uint8_t rbs[2];
    uint8_t edge, elim, gc_bias, type;
    uint8_t mot_len, mot_spacer, mot_spacendx, _pad;
} po_node;

/* src/pyrodigal/lib.pxd:274-278 */
typedef struct po_gene {
    int32_t begin, end, start_ndx, stop_ndx;
} po_gene;

typedef struct po_params {
    int32_t closed;
    int32_t min_gene;
    int32_t min_edge_gene;
    int32_t max_overlap;
} po_params;

typedef struct po_ctx po_ctx;

po_ctx*  po_new(const char* ascii, int64_t len, int mask, int mask_size);
int      po_masks(const po_ctx* c, int32_t* out, int cap);
void     po_free(po_ctx*);

int      po_slen(const po_ctx*);
double   po_gc(const po_ctx*);
int      po_unknown(const po_ctx*);
const uint8_t* po_digits(const po_ctx*);

int      po_node_size(void);
int      po_num_nodes(const po_ctx*);
po_node* po_nodes(po_ctx*);
int      po_num_genes(const po_ctx*);
po_gene* po_genes(po_ctx*);

/* individual stages (mirror Nodes.extract/sort/score, ConnectionScorer...) */
int  po_extract(po_ctx*, int tt, const po_params*);
void po_sort(po_ctx*);
void po_reset_scores(po_ctx*);
void po_score_nodes(po_ctx*, const po_training*, int closed, int is_meta);
void po_overlapping_starts(po_ctx*, const po_training*, int flag, int max_overlap);
int  po_dprog(po_ctx*, const po_training*, int final, int use_filter);
void po_record_gc_bias(po_ctx*, po_training*);   /* training: GC frame plot + frame bias of every start */
void po_dprog_raw(po_ctx*, const po_training*, int final);   /* connection loop only, no fix-ups */
int  po_find_max_index(const po_ctx*);
void po_eliminate_bad_genes(po_ctx*, int ipath, const po_training*);
int  po_extract_genes(po_ctx*, int ipath);
void po_tweak_final_starts(po_ctx*, const po_training*, int max_overlap);

/* whole-path drivers (GeneFinder._find_genes_single/_meta/_train) */
int  po_find_genes_single(po_ctx*, const po_training*, const po_params*);
int  po_find_genes_meta(po_ctx*, const po_training* const* bins, int nbins, const po_params*);
/* the same for n sequences on a pool of threads sharing the models (bench.py's all-core CPU baseline); returns the genes found */
int64_t po_find_genes_meta_pool(const char* const* seqs, const int64_t* lens, int n, const po_training* const* bins, int nbins,
                                const po_params* p, int threads);
int  po_train(po_ctx*, po_training* out, const po_params*, int force_nonsd,
              double start_weight, int tt);
int  po_train_upto(po_ctx*, po_training* out, const po_params*, int force_nonsd,
              double start_weight, int tt, int upto);

double po_last_path_score(const po_ctx*);  /* nodes[ipath].score of last winning DP */
int    po_last_ipath(const po_ctx*);

#ifdef __cplusplus
}
#endif
#endif
