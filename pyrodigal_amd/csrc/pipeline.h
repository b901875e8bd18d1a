// Device-side data layout of the finder pipeline (see DESIGN.md "Data layout in HBM").
#pragma once
#include "pga_internal.h"

// One contig of the batch: `len` bases starting at byte `base` of the concatenated batch buffers.
struct ContigDesc {
    int64_t base;
    int32_t len;
    int32_t _pad;
};

// One 3072-position tile of one contig (forward coordinates; every position of a contig of three bases or more lies in one) for the extraction kernels.
#define PGA_STAGE_SLACK 8     // extra staging slots of every tile under GroupArrays::st_half
struct TileDesc {
    int32_t contig;
    int32_t start;
};

// Per translation-table group: per-position scratch of the extraction and the node topology
// (fields that do not depend on the model), indexed by global node number.
struct GroupArrays {
    // per position (whole batch)
    uint8_t* df;                             // the digit of the position (bits 0-2) | 16 = a forward node has its ndx here | 32 = a reverse node has
                                             // (written densely by the tile that owns the position): all an ORF walk reads of a position is one byte
    // the index of a position's first node (the forward one when both strands have one there: the reverse node then is the next, ref:
    // lib.pyx:2489-2493) is a COUNT: c16[192 tile + g] = index of the first node at or after position 16 g of the tile (k_place_nodes; a
    // contig's tiles are consecutive from tile0[contig], so that is entry position >> 4 of the contig), and the sixteen bytes of df from
    // there on carry the node flags of the positions in between.  (Until the end of round 6 an int32 per POSITION: 500 MB per group and
    // call, written four bytes at a time wherever a node sat -- most of what k_place_nodes moved.)
    int32_t* c16; const int32_t* tile0 = nullptr;
    // staging: the nodes of the tile that starts at global position g, packed in order from slot 2 g (two slots per position: every
    // position can hold a node on either strand) -- or, st_half = s > 0, from slot g >> s with room for one node per 2^s positions
    // of the tile (s = 1 in production: sequence has a node every 25 positions or so, periodic worst cases reach one in two); a
    // tile that does not fit says so in *st_overflow and the caller extracts again with the full staging
    int32_t* st_ndx; int32_t* st_sv;         // ndx, stop_val
    uint8_t* st_info;                        // type | edge << 2 | reverse << 3
    int32_t st_half = 0; int32_t* st_overflow = nullptr;
    int32_t* stop_list;                      // the stop nodes of the group in node order (node indices); contig c owns [sbase[c], sbase[c + 1])
    int32_t* start_list = nullptr;           // the other nodes of the group in node order (k_place_nodes; nullptr: not wanted): the start scorer's work list
    uint32_t* ovl_topo;                      // per entry of stop_list: which of its first 16 neighbours can be overlapping starts (k_ovl_topo)
    int32_t* srank = nullptr;                // per node, written for stop nodes only: its rank among the stop nodes of its contig (k_ovl_topo):
                                             // the k-th stop of a chain owns extras record ChainDesc::soff + k of the wave-batch scorer
    // per node, in (contig, ndx, strand) order
    int32_t* ndx; int32_t* stop_val; uint8_t* type; int8_t* strand; uint8_t* edge0; float* gc_cont;
    int32_t* contig_of = nullptr;            // the node's contig (k_place_nodes): kernels over nodes read it instead of searching the contigs' first nodes
};

// Per (contig, model) chain node fields (SoA over all chains of the batch).
struct ChainArrays {
    double* cscore; double* sscore; double* rscore; double* uscore; double* tscore; double* mot_score;
    double* cscore_raw;    // the coding score as the ORF walk leaves it (k_coding_score); `cscore` is what Nodes._score makes of it
    int32_t* star_ptr;     // [n][3]
    int32_t* mot_ndx;
    uint8_t* rbs;          // [n][2]
    uint8_t* edge;         // edge flag after Nodes._score's conversion
    uint8_t* mot_len; uint8_t* mot_spacer; uint8_t* mot_spacendx;
    double* cs_sink = nullptr;   // as many doubles as the largest contig has nodes: where the ORF walks of k_coding_score_quads store the sums of
                                 // table columns a contig has no model for (unconditional stores instead of a divergent branch per column)
};

// Host-computed per-model constants of the node scorer (libm log/pow stay on the host).
struct ModelScoreConst {
    double lfac_min, lfac_max;          // ref: lib.pyx:2146-2147
    double lfac_tab[1001];              // log((1-p^n)/p^n) - lfac_min for n = 0..1000 codons (ref: lib.pyx:2209-2210)
};

// Masked regions (ref: lib.pyx:699-713): per contig a sorted run of [begin, end) intervals.
struct MaskList {
    const int32_t* off;     // [n_contigs + 1] into iv, or nullptr when masking is off
    const int2* iv;         // begin, end (contig coordinates)
};
struct MaskRun { int32_t contig, begin, end, _pad; };

struct ScoreParams {
    int32_t closed, is_meta, max_overlap, n_models;       // n_models: entries of the model array the chains index
    double* cs_out;          // or nullptr: cscore + sscore of every node, indexed like the chain arrays (what the wave-batch connection scorer reads)
    uint8_t* conv_flag;      // per contig (of the group being scored), or nullptr: set when the contig holds a start node that the
                             // reference turns into an edge node while scoring (lib.pyx:2424-2434) -- only such contigs are scored
                             // differently by the first and by a later model of a run
    unsigned long long* prof = nullptr;   // PGA_SS_PROFILE=1: wave-cycles per phase of k_score_starts (16 slots), or nullptr
    int32_t models_per_pass = 512;        // k_score_starts walks a workgroup's models in sets of this many (<= 512; PGA_SS_MODELS_PER_PASS: tests)
    int32_t lean_stops = 0;               // 1: of a stop node's per-chain fields only `edge` is written (round 6).  A stop node carries no start
                                          // scores (reset_node_scores: zeros); the wave-batch connection scorer, the overlapping-start search and the
                                          // device tail read a stop node's `edge` and nothing else of them, so the path proper -- no node arrays asked
                                          // for -- leaves the other twelve fields unwritten: 66 bytes per (stop node, model) pair, 0.6 GB per 125 Mbp call
};

void pga_launch_digitize(const char* d_seq, uint8_t* d_dig, int64_t total, const ContigDesc* d_ct, int n_contigs,
                         int32_t* d_gc, int32_t* d_unk, hipStream_t st);
// GC-or-unknown bases before every 16th position of the batch (d_p16: total / 16 + 2 entries); scratch of pga_gc_blocks(total) + 1 entries each
int64_t pga_gc_blocks(int64_t total);
void pga_launch_gc_prefix(const uint8_t* d_dig, int64_t total, int32_t* d_block_sum, int32_t* d_block_off, int32_t* d_p16, hipStream_t st);
// nodes of one translation-table group: staged per tile, counted (d_tile_count[n_tiles], d_tile_off[n_tiles + 1]), first node of
// every contig in d_cbase[n_contigs + 1]; d_tile0[c] = first tile of contig c (n_contigs + 1 entries).  pga_launch_place moves the
// staged nodes to ga.ndx .. ga.edge0 once those are allocated.
void pga_launch_extract(const uint8_t* d_dig, int64_t total, const ContigDesc* d_ct, int n_contigs, int tt,
                        const pga_params& p, const GroupArrays& ga, const TileDesc* d_tiles, int n_tiles, const int32_t* d_tile0,
                        int32_t* d_tile_first, int32_t* d_tile_last, int32_t* d_tile_count, int32_t* d_tile_off, int32_t* d_cbase,
                        int32_t* d_tile_scount, int32_t* d_tile_soff, int32_t* d_sbase /* the same three for stop nodes only */,
                        MaskList masks, hipStream_t st,
                        const uint8_t* d_enabled = nullptr /* per contig: extract it in this group?  nullptr = every contig */);
void pga_launch_place(const ContigDesc* d_ct, const TileDesc* d_tiles, int n_tiles, const int32_t* d_tile_off, const int32_t* d_tile_soff,
                      const GroupArrays& ga, hipStream_t st);
// Overlapping starts over (chain, stop node) pairs instead of over every chain node (pga_launch_score with `stops`):
struct StopLaunch {
    const int32_t* sbase = nullptr;     // per contig of the group: first entry of ga.stop_list (n_contigs + 1)
    int64_t soff_begin = 0, n_pairs = 0; // ChainDesc::soff of the launch's first chain; (chain, stop) pairs of its chains
    int32_t n_stops = 0;                 // stop nodes of the group (entries of ga.stop_list)
    // the wave-batch scorer's 64-byte extras of every stop node, built in the same pass (nullptr: not wanted)
    const int32_t* topo_q2 = nullptr; void* ext = nullptr;
    // star_ptr of the nodes that are no stop nodes is -1: the launcher fills the chains' range first -- unless nobody will read it
    // (the wave-batch scorer takes the extras, the tail asks for stop nodes only, the caller does not want the node arrays)
    bool fill_star_ptr = true;
    // the start scorer runs over ga.start_list (n_starts entries) instead of over every node (its workgroups clear the fields of the
    // stop nodes between their start nodes at the end)
    bool starts_only = false; int32_t n_starts = 0;
    // per workgroup of k_ovl_stops (256 pairs from soff_begin on): the chain of its first pair, relative to the launch's first chain
    // (the caller's plan; nullptr: the workgroup searches)
    const int32_t* blk_chain = nullptr;
};
// per (group, contig): is any model of the group inside the contig's GC window?  (meta mode)
void pga_launch_group_enable(const ContigDesc* d_ct, int n_contigs, const int32_t* d_gc_count, const double* d_model_gc,
                             const int32_t* d_model_group, int n_models, int n_groups, uint8_t* d_enabled, hipStream_t st);
// runs of unknown bases of at least min_mask positions, unordered, at most `cap` of them; *d_count is reset first
void pga_launch_find_masks(const uint8_t* d_dig, const ContigDesc* d_ct, int n_contigs, const TileDesc* d_tiles, int n_tiles, int min_mask,
                           MaskRun* d_runs, int32_t* d_count, int cap, hipStream_t st);
int pga_extract_tile_size();
void pga_launch_orf_gc(const ContigDesc* d_ct, int n_contigs, const uint8_t* d_dig, const int32_t* d_p16, const GroupArrays& ga,
                       int n_nodes_total, const int32_t* d_node_contig_base, hipStream_t st);
// chains[0..n_chains): the chains of ONE translation-table group, contiguous in `off` from node_begin
void pga_launch_score(const ChainDesc* d_chains, int n_chains, int64_t node_begin, int64_t total, const uint8_t* d_dig,
                      const ContigDesc* d_ct, const GroupArrays& ga, const pga_training* d_models,
                      const ModelScoreConst* d_msc, const ModelConst* d_mc, const ChainArrays& ca, ScoreParams sp,
                      const ChainDesc* d_all_chains /* indexed by contig_chains[].x */, const int2* d_contig_chains /* per contig: first chain, count */,
                      const int32_t* d_node_contig_base, int n_contigs, int group_nodes, const unsigned* d_sd_lut, hipStream_t st,
                      int reuse_raw_cscore = 0 /* 1: the chains read the raw coding scores another chain left (ChainDesc::raw_off): no ORF walk */,
                      const double* d_gil = nullptr /* hexamer tables of the group's models, interleaved: [4096][il_stride] */, int il_stride = 0,
                      const int32_t* d_rank = nullptr /* model -> column of d_gil */,
                      const void* d_cs_tasks = nullptr, int n_cs_tasks = 0, const void* d_cs_entries = nullptr /* pga_cs_tasks: ORF walks from LDS tables */,
                      const StopLaunch* stops = nullptr);
// host: the tasks of the LDS form of the coding score for one group (pipeline.hip); false = models are not neighbours in the table
bool pga_cs_tasks(const int2* h_cc, int n_contigs, const ChainDesc* h_chains, const int32_t* h_cbase, const int32_t* model_rank, int task_nodes,
                  std::vector<int32_t>& tasks, std::vector<int32_t>& entries);
// the RBS search tabulated: 1920 words, filled once per context (see sd_hits in pipeline.hip)
void pga_launch_sd_lut(unsigned* d_lut, hipStream_t st);
