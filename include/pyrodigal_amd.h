/*
 * pyrodigal_amd.h -- C-ABI of the MI355X-native Prodigal gene-finding core.
 *
 * Plain C linkage, plain pointers and sizes, no exceptions cross this boundary.
 * Every entry point returns 0 on success or a negative PGA_E* code; the message
 * for the last failure on a context is available through pga_last_error().
 * All "ref:" citations are relative to /root/reference/src/pyrodigal.
 *
 * Two levels, both replacing reference plug points:
 *
 *  1. scorer level  -- pga_score_connections(): whole-array drop-in for
 *     ConnectionScorer.index() + score_connections()  (ref: lib.pyx:1126-1237,
 *     1336-1357; _connection.h:386-408; impl/generic.h:13-49).
 *  2. finder level  -- pga_find_genes_batch(): whole-batch drop-in for
 *     GeneFinder.find_genes() in meta or single mode  (ref: lib.pyx:5281-5469).
 */
#ifndef PYRODIGAL_AMD_H
#define PYRODIGAL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGA_OK          0
#define PGA_EINVAL     (-1)   /* bad argument (maps to ValueError)  */
#define PGA_ENOMEM     (-2)   /* host or device allocation failed (MemoryError) */
#define PGA_EDEVICE    (-3)   /* HIP runtime error (RuntimeError) */
#define PGA_ENODEVICE  (-4)   /* no gfx950 device visible (RuntimeError) */

typedef struct pga_ctx pga_ctx;

/* Byte layout of the reference's `struct _training` (ref: prodigal/training.pxd:3-14),
 * 558 392 bytes; TrainingInfo.dump() files can be passed as-is. */
typedef struct pga_training {
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
} pga_training;

/* GeneFinder constructor options (ref: lib.pyx:5102-5115). */
typedef struct pga_params {
    int32_t closed;         /* default 0  */
    int32_t min_gene;       /* default 90 */
    int32_t min_edge_gene;  /* default 60 */
    int32_t max_overlap;    /* default 60 */
    int32_t meta;           /* 1: meta mode over all loaded models; 0: single mode with model 0 */
    int32_t want_nodes;     /* 1: also return the winning model's full node arrays */
    int32_t mask;           /* 1: no gene may run across a masked region (runs of unknown bases), default 0 */
    int32_t min_mask;       /* shortest run of unknown bases that is masked, default 50 (ref: lib.pyx:5102-5115, 699-713) */
} pga_params;

/* One predicted gene (ref: lib.pxd:274-278 `_gene` + the start/stop node fields Gene reads,
 * lib.pyx:2644-2830). Coordinates are 1-based inclusive like the reference. */
typedef struct pga_gene {
    int32_t contig;        /* index into the batch */
    int32_t begin, end;
    int32_t start_ndx, stop_ndx;
    int8_t  strand;
    uint8_t partial_begin, partial_end;
    uint8_t start_type;    /* 0 ATG, 1 GTG, 2 TTG, 3 Edge */
    uint8_t rbs[2];
    uint8_t mot_len, mot_spacer;
    int32_t mot_ndx;
    float   gc_cont;
    double  cscore, sscore, rscore, uscore, tscore, mot_score;
} pga_gene;

/* Node arrays of one contig for its winning model (SoA; ref: src/Prodigal/node.h:40-76). */
typedef struct pga_nodes {
    int32_t  n;
    int32_t *ndx, *stop_val, *traceb, *tracef, *star_ptr /* [n][3] */;
    uint8_t *type, *edge, *elim, *rbs /* [n][2] */;
    int8_t  *strand, *ov_mark;
    float   *gc_cont;
    double  *cscore, *sscore, *rscore, *uscore, *tscore, *score, *mot_score;
    int32_t *mot_ndx;
    uint8_t *mot_len, *mot_spacer, *mot_spacendx;
} pga_nodes;

typedef struct pga_contig_result {
    int32_t model;        /* winning model index, -1 if none (ref: lib.pyx:5317-5396) */
    int32_t n_nodes;
    int64_t gene_begin;   /* genes[gene_begin .. gene_begin + n_genes) */
    int32_t n_genes;
    int32_t n_unknown;    /* bases that are not A, C, G or T (ref: lib.pyx:664-697) */
    double  gc;
    double  score;        /* nodes[ipath].score of the winning DP pass */
} pga_contig_result;

typedef struct pga_result {
    int32_t            n_contigs;
    int64_t            n_genes;
    pga_contig_result* contigs;
    pga_gene*          genes;
    pga_nodes*         nodes;     /* NULL unless want_nodes */
    double             t_total_ms, t_dp_ms;   /* device time of the whole batch / of the DP kernel */
    int64_t            node_passes;           /* sum over (contig, model) DP passes of node count */
    int32_t            n_chains;              /* number of (contig, model) DP passes */
    int32_t            _pad;
    /* masked regions when params.mask is set, else NULL: contig i owns masks[2*k], masks[2*k+1] = [begin, end)
     * for k in [mask_off[i], mask_off[i+1])   (ref: lib.pyx:699-713, `Sequence.masks`) */
    int32_t*           mask_off;
    int32_t*           masks;
} pga_result;

/* ---- context ---------------------------------------------------------- */
int         pga_create(int device, pga_ctx** out);
void        pga_destroy(pga_ctx*);
const char* pga_last_error(const pga_ctx*);
int         pga_device_info(const pga_ctx*, char* name, int name_len, int* cus, int64_t* hbm_bytes);
/* How the connection scoring of the last pga_find_genes / pga_find_genes_batch / pga_score_connections call on this
 * context ran (diagnostics; no counterpart in the reference, whose dynamic programme is one serial loop):
 *   out[0] chains that were cut into segments (0: every chain was walked serially)   out[1] segments
 *   out[2..4] nodes whose speculative result the verification rounds 1..3 rejected
 *   out[5] chains that were walked serially in the end because they never verified clean
 *   out[6] 1 when the wave-batch scorer ran from a step schedule   out[7] 64-node batches whose schedule did not fit its
 *          buffer (> 0: the launch was repeated by the kernel that works the lane masks out per chain) */
int         pga_dp_stats(const pga_ctx*, int32_t out[8]);
/* Device time (HIP events on the context's stream, milliseconds) of the connection scoring of the last pga_find_genes /
 * pga_find_genes_batch call on this context, by part (diagnostics; the reference's counterpart of all three is ConnectionScorer.index +
 * score_connections, src/pyrodigal/lib.pyx:1126-1176, 1205-1237):
 *   out[0] the connection-scoring launch(es) themselves (= pga_result.t_dp_ms; both launches when a schedule miss repeated the launch)
 *   out[1] the topology kernels of every translation-table group (windows, near-zone starts, candidate links)
 *   out[2] the step-schedule kernels of every group   out[3] reserved (0) */
int         pga_dp_timings(const pga_ctx*, double out[4]);
/* How the node extraction of the last pga_find_genes / pga_find_genes_batch / pga_nodes_stage call on this context ran (diagnostics):
 *   out[0] extraction passes (1; 2 when a tile of the batch did not fit the half-density staging and the batch was extracted again)
 *   out[1] reserved (0) */
int         pga_extract_stats(const pga_ctx*, int32_t out[2]);
/* How a connection-scoring launch over chains of these node counts would be cut (host arithmetic only, no device needed;
 * the PGA_DP_SEG* environment variables of INTEGRATION.md apply):
 *   out[0] chains cut into segments   out[1] segments   out[2] nodes of the longest sub-chain (segment + warm-up)
 *   out[3] scratch elements behind the real chains in every per-node array */
int         pga_dp_plan_summary(int32_t n_chains, const int32_t* nodes_per_chain, int64_t out[4]);
/* The order in which the wave-batch connection scorer starts the chains of a launch: longest first by walk batches of 64 nodes,
 * launch order among equals (host arithmetic only).  order[k] = index of the chain started k-th. */
int         pga_dp_start_order(int32_t n_chains, const int32_t* nodes_per_chain, int32_t* order);
/* ... and that order made XCD-aware (workgroup b runs on XCD b % 8): chains with the same key -- a contig under one translation
 * table, i.e. the same topology arrays -- all go to one XCD, the one with the fewest nodes so far; out[8 k + x] is the k-th chain of
 * XCD x, -1 where a queue has ended.  Returns the entries written (a multiple of 8) or a negative PGA_E* code. */
int64_t     pga_dp_xcd_order(int32_t n_chains, const int32_t* order, const int32_t* nodes_per_chain, const int32_t* key_of_chain,
                             int32_t n_keys, int32_t* out, int64_t out_cap);
/* How the ORF walks of the coding score (LDS-table form) of one translation-table group are cut into tasks (host arithmetic only):
 * contig i has nodes_per_contig[i] nodes and is scored for models_per_contig[i] models whose table columns start at
 * first_column[i] (columns of a contig are neighbours; every four of them are one walk).
 *   out[0] tasks   out[1] entries (pieces of contigs)   out[2] nodes of the largest task   out[3] nodes over all tasks
 *   out[4] 1 if the tasks of higher columns come first; returns PGA_EINVAL when the LDS form does not apply */
int         pga_cs_task_summary(int32_t n_contigs, const int32_t* nodes_per_contig, const int32_t* first_column,
                                const int32_t* models_per_contig, int32_t task_nodes, int64_t out[5]);

/* ---- models (MetagenomicBins / TrainingInfo, ref: lib.pyx:4888-5069, 3898-3953) ---- */
int pga_set_models(pga_ctx*, const pga_training* const* models, int n_models);

/* ---- scorer level ------------------------------------------------------ */
/* Whole-array connection scoring of one sorted node list (the gene prediction pass, final = 1).
 * Inputs are the node fields _score_connections reads; outputs are the fields it writes. */
int pga_score_connections(pga_ctx*, int32_t n,
                          const int32_t* ndx, const int32_t* stop_val,
                          const uint8_t* type, const int8_t* strand,
                          const double* cscore, const double* sscore,
                          const double* rscore, const double* uscore,
                          const int32_t* star_ptr /* [n][3] */,
                          double st_wt, int final /* must be 1: see pga_score_connections_training for the training pass */,
                          double* score, int32_t* traceb, int8_t* ov_mark,
                          int32_t* max_index /* _find_max_index, may be NULL */,
                          double* kernel_ms /* may be NULL */);

/* The training pass of the same scorer (final = 0 in the reference, _connection.h:94-367): a connection is worth its length
 * times bias . gc_score of one of its nodes.  gc_score is the [n][3] array of the nodes after Prodigal's record_gc_bias
 * (call site lib.pyx:5261), bias = TrainingInfo.bias, star_ptr as left by _record_overlapping_starts(flag = 0). */
int pga_score_connections_training(pga_ctx*, int32_t n,
                                   const int32_t* ndx, const int32_t* stop_val,
                                   const uint8_t* type, const int8_t* strand,
                                   const double* gc_score /* [n][3] */, const double* bias /* [3] */,
                                   const int32_t* star_ptr /* [n][3] */, double st_wt,
                                   double* score, int32_t* traceb, int8_t* ov_mark,
                                   int32_t* max_index /* may be NULL */, double* kernel_ms /* may be NULL */);

/* ---- finder level ------------------------------------------------------ */
/* `seqs[c]` points at `lens[c]` ASCII nucleotides (any case, non-ACGT = unknown).
 * One call = GeneFinder.find_genes() on every contig of the batch (ref: lib.pyx:5400-5469). */
int  pga_find_genes_batch(pga_ctx*, int32_t n_contigs, const char* const* seqs, const int64_t* lens,
                          const pga_params*, pga_result** out);
void pga_result_free(pga_result*);

/* The same in two steps, for callers that keep a batch resident in HBM (repeated passes over the
 * same contigs with different options/models, benchmarking without the PCIe upload):
 * pga_batch_create packs and uploads the contigs once, pga_find_genes runs the whole path on it.
 * A context runs ONE pga_find_genes / pga_find_genes_batch at a time; pga_batch_create / pga_batch_create_packed / pga_batch_free may be
 * called from another thread while it does (the upload has a stream, a pinned staging area and a worker pool of its own: the next batch
 * of a context can be on its way while the current one is worked on), one upload at a time per context. */
typedef struct pga_batch pga_batch;
int  pga_batch_create(pga_ctx*, int32_t n_contigs, const char* const* seqs, const int64_t* lens, pga_batch** out);
/* The same from contigs that already lie back to back in one host buffer (offs[i + 1] == offs[i] + lens[i]), ideally pinned
 * (pga_fasta_next_packed): no host-side packing, one DMA of the whole batch.  The buffer may be reused when the call returns. */
int  pga_batch_create_packed(pga_ctx*, int32_t n_contigs, const char* packed, const int64_t* offs, const int64_t* lens, pga_batch** out);
void pga_batch_free(pga_batch*);
int  pga_find_genes(pga_ctx*, const pga_batch*, const pga_params*, pga_result** out);

/* ---- stage level --------------------------------------------------------- */
/* The node arrays as the reference's Nodes methods leave them, one pga_nodes per contig of the batch
 * (pga_result.nodes; no genes).  Later stages include the earlier ones and use model 0 of the context;
 * `params.meta` then selects the is_meta behaviour of Nodes.score, `translation_table` is only read
 * by PGA_STAGE_EXTRACT. */
#define PGA_STAGE_EXTRACT 1   /* Nodes.extract() + Nodes.sort()             (ref: lib.pyx:2501-2541, 2489-2493) */
#define PGA_STAGE_SCORE   2   /* + Nodes.reset_scores() + Nodes.score()     (ref: lib.pyx:2543-2595)            */
#define PGA_STAGE_OVERLAP 3   /* + _record_overlapping_starts(flag = 1)      (ref: lib.pyx:2279-2329, 5302)      */
#define PGA_STAGE_SEQUENCE 4  /* Sequence.__init__ only: gc, n_unknown and masks per contig, no nodes (ref: lib.pyx:664-713) */
int pga_nodes_stage(pga_ctx*, const pga_batch*, const pga_params*, int stage, int translation_table, pga_result** out);

/* ---- translation -------------------------------------------------------------- */
/* Protein translations of gene records, computed on the device from a resident batch: one thread per codon
 * (ref: lib.pyx:2932-3047 `Gene.translate`, 770-789 `Sequence._amino`, _translation.h:4-42 the genetic codes).
 *   genes[g]         records of a pga_result of THIS batch (contig, begin, end, strand, partial flags are read)
 *   table_of_contig  translation table of every contig of the batch (what the winning model was trained with, or the caller's
 *                    choice); one of the NCBI tables the reference knows (1-6, 9-16, 21-26, 29, 30, 32, 33)
 *   unknown_residue  the letter of a codon with an unknown base ('X' in the reference)
 *   include_stop     0: a complete gene loses its final `*`
 *   strict           0: a codon with one unknown base in second or third position reads the residue all four completions agree on
 *   offsets[g]       first letter of gene g in `out`: offsets[g + 1] - offsets[g] must be (end - begin + 1) / 3, minus 1 for a
 *                    gene whose stop is not at an edge when include_stop == 0; offsets[n_genes] letters are written to `out`
 * As in the reference the first codon of a gene that does not start at an edge reads M when it is a start codon of the table,
 * and a stop codon of the table reads `*`. */
int pga_translate_genes(pga_ctx*, const pga_batch*, int64_t n_genes, const pga_gene* genes, const int32_t* table_of_contig,
                        int unknown_residue, int include_stop, int strict, const int64_t* offsets, char* out);

/* ---- training --------------------------------------------------------------- */
/* Single-genome training (ref: lib.pyx:5236-5279 `GeneFinder._train`): `batch` holds exactly ONE sequence (several
 * training sequences are joined by the caller with the reference's TTAATTAATTAA spacer, lib.pyx:5510-5532); closed,
 * min_gene, min_edge_gene, max_overlap and mask come from `params`.  On success `*out` is the complete TrainingInfo.
 * `upto` = 0 trains completely; 1 / 2 / 3 stop after the GC frame bias / the hexamer statistics / the Shine-Dalgarno
 * start training (partial structs, for validation). */
int pga_train(pga_ctx*, const pga_batch*, const pga_params*, int translation_table, double start_weight, int force_nonsd,
              int upto, pga_training* out);

/* ---- FASTA ingest (host side) ---------------------------------------------- */
/* Multi-record FASTA, plain or gzip, read in batches ready for pga_find_genes_batch / pga_batch_create
 * (ref: src/pyrodigal/tests/fasta.py:59-86 `parse`, src/pyrodigal/cli.py:32-61).  headers[i] is the header line
 * without '>' (id = first word, description = the rest); seqs[i] holds lens[i] letters with every blank
 * removed, not NUL-terminated.  The arrays are valid until the next call on the same reader; a batch ends
 * after max_records records or once max_bases bases are exceeded (0 = no limit); *n_records == 0 at end of file. */
typedef struct pga_fasta pga_fasta;
/* plain files are mapped and parsed by several threads; gzip files are inflated with zlib on one thread; a file in another
 * compression format is rejected (PGA_EINVAL): decompress it into pga_fasta_open_callback.  A mapped file must not be truncated by
 * another process while it is read: the kernel answers a read behind the new end with SIGBUS, which no return code can stand for
 * (feed files that may shrink through pga_fasta_open_callback and read() instead). */
int         pga_fasta_open(const char* path, pga_fasta** out);
/* The same reader over a byte stream the caller produces: read(user, buf, cap) fills up to cap bytes and returns their number,
 * 0 at the end of the stream, negative on failure (ref: tests/fasta.py:16-57 `zopen` -- the reference sniffs bz2 / xz / lz4 / zstd
 * and decompresses with Python modules; the Python layer here hands those decompressors in through this entry point). */
typedef int64_t (*pga_fasta_read_fn)(void* user, char* buf, int64_t cap);
int         pga_fasta_open_callback(pga_fasta_read_fn read, void* user, pga_fasta** out);
int         pga_fasta_next(pga_fasta*, int64_t max_bases, int32_t max_records, int32_t* n_records,
                           const char* const** headers, const char* const** seqs, const int64_t** lens);
const char* pga_fasta_error(const pga_fasta*);
void        pga_fasta_close(pga_fasta*);
/* The pinned staging arenas of closed readers wait in a process-wide pool for the next reader (at most eight arenas and 512 MB,
 * allocated hipHostMallocPortable, so a reader on any device can take them); this gives them back to the system. */
void        pga_fasta_release_spare(void);
/* Device and pinned buffers of destroyed contexts wait in a process-wide cache for the next context (at most PGA_CACHE_GB, default
 * 16 GB, of device memory and 2 GB of pinned memory; 0 turns the cache off); this gives them back to the system. */
void        pga_release_cached(void);
/* The same records with their sequences packed back to back in PINNED host memory (hipHostMalloc): `*packed` holds the
 * letters of the batch, record i at offs[i], lens[i] long, offs[i + 1] == offs[i] + lens[i].  The reader owns `n_arenas`
 * staging arenas (2 .. 8, fixed at the first call) and fills them in turn: the LETTERS of a call (`*packed`) stay valid until
 * that arena comes up again, i.e. for the next n_arenas - 1 calls -- batch k can be on its way to the device
 * (pga_batch_create_packed) while batch k + 1 is being parsed (ref: the reader the reference's CLI feeds its thread pool
 * with, cli.py:287-302).  `headers`, `offs` and `lens` belong to the reader and are only valid until the NEXT call: copy them
 * (or hand them to pga_batch_create_packed, which copies) before asking for the next batch. */
int         pga_fasta_next_packed(pga_fasta*, int64_t max_bases, int32_t max_records, int32_t n_arenas, int32_t* n_records,
                                  const char* const** headers, const char** packed, const int64_t** offs, const int64_t** lens);

#ifdef __cplusplus
}
#endif
#endif
