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
    double  _pad2[2];
};
// Per-node record only needed when the node is the target i. 32 bytes.
struct __attribute__((aligned(32))) DpTgt {
    int32_t n3ndx[3];
    int32_t n3stop[3];
    int32_t lo;      // first candidate index of the window (ref: lib.pyx:1221-1233)
    int32_t _pad;
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
};

// Node fields in device memory (struct of arrays; each pointer covers the whole batch).
struct NodeArrays {
    const int32_t* ndx; const int32_t* stop_val; const uint8_t* type; const int8_t* strand;   // indexed by topo_off + i
    const double* cscore; const double* sscore; const double* rscore; const double* uscore;   // indexed by off + i
    const int32_t* star_ptr;   // [n][3], indexed by off + i
};

struct DpBuffers {
    DpSrc* src; DpTgt* tgt;
    double* score; int32_t* traceb; int32_t* tbn; int8_t* ov_mark;
    int32_t* max_index; double* max_score;    // per chain: _find_max_index and its score
    int32_t* ipath;                           // per chain: max_index, or -1 when that node has no traceb (ref: lib.pyx:1311)
};

// kernel launchers (dp.hip)
// chains[0..n_chains) must be contiguous in `off`; node_begin = chains[0].off, total_nodes = their node count
void pga_launch_dp_prepare(const ChainDesc* d_chains, int n_chains, int64_t node_begin, int64_t total_nodes,
                           const NodeArrays& nodes, const ModelConst* d_models, DpBuffers buf, hipStream_t st);
void pga_launch_dp(const ChainDesc* d_chains, int n_chains, const ModelConst* d_models, DpBuffers buf,
                   int final, hipStream_t st);

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
};

// finder.hip
void pga_finder_release(pga_ctx*);
int  pga_finder_models_changed(pga_ctx*);
void pga_fill_model_const(ModelConst* mc, double st_wt);
int  pga_hip_try_(pga_ctx* c, hipError_t e, const char* what);
