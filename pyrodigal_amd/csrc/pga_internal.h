// Internal types shared by the HIP kernels and the C-ABI host code.
// All "ref:" citations are relative to /root/reference/src/pyrodigal.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/pyrodigal_amd.h"

#define PGA_MAX_NODE_DIST 500   // ref: _connection.h:5 / dprog.h
#define PGA_MAX_OPP_OVLP  200
#define PGA_OPER_DIST     60    // ref: /root/reference/src/Prodigal/node.h:30-38
#define PGA_T_STOP        3

// node kinds used by the connection-scoring kernels: bit0 = stop codon, bit1 = reverse strand
// 0 = F5 (forward start), 1 = F3 (forward stop), 2 = R5 (reverse start), 3 = R3 (reverse stop)
#define PGA_KIND(meta)      ((meta) & 3)
#define PGA_FRAME(meta)     (((meta) >> 2) & 3)
#define PGA_SPVALID(meta,k) (((meta) >> (4 + (k))) & 1)

// Static per-node record read (wave-uniformly) for every candidate source j. 64 bytes.
struct __attribute__((aligned(64))) DpSrc {
    int32_t ndx, stop_val, meta, _pad;
    double  cs;      // cscore + sscore
    double  x[3];    // F3: cs(n3_k) + igm(j, n3_k);  R3: cs(n3_k) + igm(n3_k, i)   (n3_k = nodes[star_ptr[k]])
    int32_t n3src[3];   // training pass only, F3 nodes: ndx of the overlapping start of each frame (star_ptr)
    int32_t _pad2;
};
// Per-node record only needed when the node is the target i. 80 bytes.
// The index ranges are static (positions only) and found by binary search in dp_prepare.
struct __attribute__((aligned(16))) DpTgt {
    int32_t n3ndx[3];
    int32_t n3stop[3];
    int32_t lo;      // first candidate index of the window (ref: lib.pyx:1221-1233)
    int32_t p_near;  // F5 / R3 targets: first index whose ndx >= ndx - 3*OPER_DIST (closer sources need the exact intergenic term)
    int32_t a[3];    // F3: a[0] = first index with ndx > stop_val (own ORF).  R5: a[0] = index of its own stop node,
                     //     a[1], a[2] = index range of forward stops that can overlap its 3' end.  R3: a[k] = begin of zone k
    int32_t b[3];    // R3: b[k] = end of the index range of forward stops that can overlap overlapping-start k
    int32_t c[3];    // R3: c[f] = the one reverse stop of frame f whose ORF covers this node (operon candidate) or -1
    int32_t _pad[3];
};
// Per-model constants of the connection scorer.
struct ModelConst {
    double st_wt;
    double negc;        // -0.15 * st_wt                       (ref: _connection.h:43-49)
    double igm[64];     // (2 - d/60) * 0.15 * st_wt, d = 0..60  (ref: _connection.h:73-75), host-computed
};
// One DP chain = one (contig, model) pass.
struct ChainDesc {
    int64_t off;        // first element of the chain in the per-chain arrays (scores, DP records)
    int64_t topo_off;   // first element of the contig's nodes in the topology arrays of its group
    int32_t n;          // node count
    int32_t model;
    int32_t contig;
    int32_t first;      // 1: first model scored after (re-)extraction (edge flags not yet converted;
                        //    ref: lib.pyx:2424-2434 mutates node.edge, which persists to the next bin)
    // connection scoring of a SEGMENT of a chain (dp.hip "segmented chains"): the sub-chain reads the records of the
    // chain it belongs to, from node `rebase` on, and keeps its own results at `off`
    int64_t rec_off = -1;   // first DpSrc / DpTgt record of the sub-chain; -1: the chain's own, at `off`
    int32_t rebase = 0;     // chain index of the sub-chain's node 0: index fields of DpTgt are shifted down by it
    int32_t group = 0;      // translation-table group of the chain's model: whose topology arrays `topo_off` indexes (dp_wave.hip)
    int64_t raw_off = -1;   // node scoring: read the raw coding scores of the chain at this offset (same contig, same model)
                            // instead of the chain's own; -1: its own
    int64_t soff = 0;       // stop nodes of the chains before this one (in launch order): its first slot in a launch over (chain, stop) pairs
    int32_t sched_b0 = 0;   // wave-batch scorer: first 64-node batch of the chain's contig in its group's step schedule (dpw_core.h)
    int32_t _pad0 = 0;
};

// Node fields in device memory (struct of arrays; each pointer covers the whole batch).
struct NodeArrays {
    const int32_t* ndx; const int32_t* stop_val; const uint8_t* type; const int8_t* strand;   // indexed by topo_off + i
    const double* cscore; const double* sscore; const double* rscore; const double* uscore;   // indexed by off + i
    const int32_t* star_ptr;   // [n][3], indexed by off + i
    const double* gcb;         // training pass (final = 0) only: bias . gc_score of each node, indexed by off + i
};

struct DpBuffers {
    DpSrc* src; DpTgt* tgt;
    double* score; int32_t* traceb; int32_t* tbn; int8_t* ov_mark;
    int32_t* max_index; double* max_score;    // per chain: _find_max_index and its score
    int32_t* ipath;                           // per chain: max_index, or -1 when that node has no traceb (ref: lib.pyx:1311)
    // far-field candidate values of finalized nodes (see dp.hip)
    double* A;                                // gene ends:  score + igm_diff                       (-inf when unusable)
    double* V[3];                             // per target frame f: forward start of frame f: score + cs; forward stop: score + x[f]
    double* hv; int32_t* hi;                  // 8-ary max tree over A: (value, index) of every complete block
    unsigned long long* prof;                 // optional [8] phase cycle counters of chain 0 (PGA_DP_PROFILE), else NULL
};

// Segmented connection scoring of long chains (see dp.hip "segmented chains").
// One segment = nodes [s, e) of a chain, walked speculatively as a sub-chain that starts `s - a` nodes early.
struct DpSeg {
    int32_t chain;      // index into the chain list of the launch
    int32_t a, s, e;    // sub-chain [a, e) of the chain; its results are kept for [s, e)
    int64_t off;        // first element of the sub-chain in the DP arrays (scratch region behind the real chains)
};
#define PGA_SEG_ROUNDS 3
struct DpSegPlan {       // host side
    std::vector<DpSeg> segs;
    std::vector<ChainDesc> p1_chains;    // speculative launch: the sub-chains, then the chains that are not segmented
    std::vector<int32_t> p1_slot;        // where each of them publishes its _find_max_index result
    std::vector<int32_t> big;            // the segmented chains
    int64_t extra = 0;                   // scratch elements needed behind the real chains in every per-node DP array
    int32_t max_seg_nodes = 0, max_seg_len = 0, max_big_n = 0;
    mutable std::vector<char> stage;     // pga_dp_seg_bind: the head of the device workspace as it is uploaded (plan + cleared flags), one copy
};
struct DpSegDev {        // device copies + workspace; all owned by the caller
    const DpSeg* segs; const ChainDesc* p1_chains; const int32_t* p1_slot; const int32_t* big;
    int32_t* flags;      // [PGA_SEG_ROUNDS][n_chains] mismatches found by each verification round
    int32_t* first_bad;  // [PGA_SEG_ROUNDS][n_chains] lowest node index each round rejected
    int32_t* ctb;        // claimed traceb of every node (chain indexing)
    double* cw;          // weight of the claimed connection
    uint32_t* hb;        // bit k: the node's height in the claimed traceb forest is >= k
    double* tv; int32_t* ti;   // per 64-node tile: best gene-end score and its index (n_nodes / 64 + n_chains + 1 entries)
    unsigned long long* tmask; int32_t* toff;   // per tile: its spine nodes, and how many come before them in the chain
    int32_t* nsp;              // [n_chains] spine nodes of the chain
    int32_t* sp_idx; int32_t* sp_tb; int32_t* sp_pp; double* sp_w;   // the spine lists (one element per node at most)
    int64_t n_nodes;     // elements of the real chains in the per-node arrays
    int32_t n_segs, n_p1, n_big, max_seg_nodes, max_seg_len, max_big_n;
    int32_t* h_round = nullptr;   // or: pinned host memory, n_chains ints -- the launcher reads a round's verdict back and stops when every chain passed
    // Round 6: the segments' speculative walk by the wave-batch kernel (dp_wave.hip, k_dp_wave<6, true>), a wavefront per segment and up to
    // eight times as many segments as the chain kernel's one workgroup per compute unit allows; nullptr: k_dp_tree_mw walks them
    const struct DpwGroupPtrs* wave_groups = nullptr;
    const struct DpwBuffers* wave_buf = nullptr;
};
// false: nothing to segment (plan left empty).  wave_walk: the segments are walked by the wave-batch kernel (a wavefront each: PGA_DP_SEG_WSLOTS,
// 2048, of them at once) instead of the chain kernel (a workgroup each: PGA_DP_SEG_SLOTS, 256)
bool pga_dp_plan(const ChainDesc* h_chains, int n_chains, int64_t tot_nodes, DpSegPlan& plan, bool wave_walk = false);
// device workspace of a segmented launch: its size, and its layout inside `arena` + the upload of the plan (the plan must
// stay alive until the copies on `st` are done)
size_t pga_dp_seg_bytes(const DpSegPlan& plan, int n_chains, int64_t tot_nodes);
hipError_t pga_dp_seg_bind(const DpSegPlan& plan, int n_chains, int64_t tot_nodes, void* arena, hipStream_t st, DpSegDev* out);

// ---- wave-batch connection scoring for launches with many chains (dp_wave.hip, dpw_core.h) ----
struct DpwExt;
// topology of one translation-table group: what depends on positions and kinds only, shared by every model of a contig
// srank: per node (stop nodes only) its rank among the stop nodes of its contig, or nullptr: then the extras record of node i of a
// chain is ext[off + i]; with it, ext[soff + srank[i]] (one 64-byte record per (chain, stop node) pair, dense)
// shdr / sent / scur (optional): the step schedule of the group (dpw_core.h "Step schedule"): one header per 64-node batch of every
// contig, 32-byte slots (DPW_SCHED_STRIDE per batch), scur[1] = batches whose entries did not fit
struct DpwSchedHdr;
struct DpwTopoArrays { const int32_t* ndx; const int32_t* stop_val; uint8_t* kf; int32_t* lo; int32_t* q1; int32_t* q2; const int32_t* srank = nullptr;
                       DpwSchedHdr* shdr = nullptr; uint4* sent = nullptr; uint32_t* scur = nullptr; };
struct DpwGroupPtrs { DpwTopoArrays g[4]; };
// per chain node: cs = cscore + sscore, suffix maxima of finished blocks; per stop node a 64-byte record of extras (indexed by
// node, or dense by (chain, stop) pair: DpwTopoArrays::srank)
struct DpwBuffers { double* cs; DpwExt* ext; double* sfxv; int32_t* sfxi; };
// which connection scorer a final-pass launch over n_chains chains uses (PGA_DP_KERNEL overrides: wave | tree1 | tree3 | scan)
bool pga_dp_use_wave(int n_chains);
// max_contig_nodes (0: unknown): the most nodes any contig of the group holds -- contigs that fit are staged in LDS, a workgroup each
void pga_launch_dpw_topo(const DpwTopoArrays& ta, const uint8_t* type, const int8_t* strand, const int32_t* d_cbase, int n_contigs, int n_nodes,
                         hipStream_t st, int max_contig_nodes = 0);
// the step schedule of a group (after pga_launch_dpw_topo): d_bbase[c] = 64-node batches of the contigs before contig c; max_batches: of
// the contig with the most nodes; ta.sent holds DPW_SCHED_STRIDE slots per batch; clears ta.scur first
void pga_launch_dpw_sched(const DpwTopoArrays& ta, const int32_t* d_cbase, const int32_t* d_bbase, int n_contigs, int max_batches, hipStream_t st);
// chains[0..n_chains) of ONE group, contiguous in `off` from node_begin
void pga_launch_dpw_chain(const ChainDesc* d_chains, int n_chains, int64_t node_begin, int64_t total_nodes, const NodeArrays& nodes,
                          const DpwTopoArrays& ta, const ModelConst* d_models, const DpwBuffers& wb, hipStream_t st);
// d_order (optional): the order in which the chains are started -- a launch ends when its last chain does, so long chains go first
// scheduled: every group carries a step schedule (k_dp_wave); else the steps' lane masks are worked out by every chain (k_dpw_dyn:
// the fallback when a schedule did not fit its buffer, and a cross-check: PGA_DPW_SCHED=0)
void pga_launch_dp_wave(const ChainDesc* d_chains, int n_chains, const DpwGroupPtrs& groups, const ModelConst* d_models, DpBuffers buf,
                        const DpwBuffers& wb, hipStream_t st, const int32_t* d_order = nullptr, int n_blocks = 0 /* entries of d_order; < 0 = filler */,
                        bool scheduled = false);
bool pga_dpw_use_sched();
// the sub-chains of a segmented launch (DpSegPlan::p1_chains, the first n_segs of them) by the scheduled kernel; results to d_slot[k]
void pga_launch_dp_wave_sub(const ChainDesc* d_subs, int n_subs, const DpwGroupPtrs& groups, const ModelConst* d_models, DpBuffers buf,
                            const DpwBuffers& wb, hipStream_t st, const int32_t* d_slot);

// kernel launchers (dp.hip)
// chains[0..n_chains) must be contiguous in `off`; node_begin = chains[0].off, total_nodes = their node count
void pga_launch_dp_prepare(const ChainDesc* d_chains, int n_chains, int64_t node_begin, int64_t total_nodes,
                           const NodeArrays& nodes, const ModelConst* d_models, DpBuffers buf, hipStream_t st, int final = 1);
// seg != nullptr (from a non-empty plan): max_index / max_score / ipath need n_chains + n_segs entries and every per-node
// array plan.extra more elements
void pga_launch_dp(const ChainDesc* d_chains, int n_chains, const ModelConst* d_models, DpBuffers buf,
                   int final, hipStream_t st, const DpSegDev* seg = nullptr);

struct FinderState;   // finder.hip

struct pga_ctx {
    int device = 0;
    FinderState* finder = nullptr;
    std::string err;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<pga_training> models;
    void* d_models_raw = nullptr;      // device copy of the pga_training structs
    ModelConst* d_model_const = nullptr;
    int n_models = 0;
    int32_t dp_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};    // pga_dp_stats
    double dp_timings[4] = {0, 0, 0, 0};               // pga_dp_timings
    int32_t extract_passes = 0;                        // pga_extract_stats: extraction passes of the last call (2: a tile overflowed the half-density staging)
};
// summary of a segmented launch's flags (host copy, [PGA_SEG_ROUNDS][stride]) into pga_ctx::dp_stats
void pga_dp_note_stats(pga_ctx* c, const DpSegPlan* plan, const int32_t* h_flags, int stride);

// finder.hip
void pga_finder_release(pga_ctx*);
int  pga_finder_models_changed(pga_ctx*);
void pga_fill_model_const(ModelConst* mc, double st_wt);
int  pga_hip_try_(pga_ctx* c, hipError_t e, const char* what);
