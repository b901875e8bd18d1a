// Host model of the contig-per-wavefront connection scorer (pyrodigal_amd/csrc/dp_contig.hip): the step function of dpc_core.h --
// the same source the kernel compiles -- run node by node over one chain (one lane of the kernel's wavefront), with plain arrays
// where the kernel has topology registers, an LDS history and the chain's arrays in HBM.  TEST INFRASTRUCTURE: built and run by
// tests/test_dpc_model.py against the CPU oracle; nothing in the product links it.
#include "../pyrodigal_amd/csrc/dpc_core.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#ifndef DPC_TB
#define DPC_TB 64           // nodes per batch of topology (the tests also build the model with 4)
#endif

namespace {

struct HostX {
    const uint8_t* kf_; const int32_t* ndx_; const int32_t* stopv_; const int32_t* tb_; const double* score_; const double* cs_; const DpwExt* ext_;
    const double* igm_;
    int64_t* stats;
    int cur;                                    // the node being walked
    int curkind = 0; long long* dbg = nullptr;  // optional event counts by kind of the node being walked: [k] history reads, [4 + k] list entries, [8 + k] nodes
    DpcHist ring[DPC_HIST];
    int c_idx[3][DPC_CAND], c_ndx[3][DPC_CAND]; DpcHist c_val[3][DPC_CAND];
    bool in_hist(int j) const { return j > cur - DPC_HIST; }
    // the kernel has the topology of the batch it is in and of the one before (DPC_TB nodes each)
    int reach() const { const int r = (cur / DPC_TB) * DPC_TB - DPC_TB; return r > 0 ? r : 0; }
    DpcCarry carry_[3]; double l3v_[3]; int l3i_[3], l3s_[3], l3n_[3];
    int l3i(int f) const { return l3i_[f]; }
    int l3s(int f) const { return l3s_[f]; }
    int l3n(int f) const { return l3n_[f]; }
    void set_l3(int f, int i, int s, int n) { l3i_[f] = i; l3s_[f] = s; l3n_[f] = n; }
    template <class F> void for_near(int a, int b, int kind, F f) {
        if (a < reach() && a < b) stats[7]++;
        for (int j = a; j < b; j++) if (DPW_KIND(kf_[j]) == kind) f(j, ndx_[j]);
    }
    DpcCarry carry(int f) const { return carry_[f]; }
    void set_carry(int f, const DpcCarry& c) { carry_[f] = c; }
    double l3v(int f) const { return l3v_[f]; }
    void set_l3v(int f, double v) { l3v_[f] = v; }
    DpcHist hist(int j) {
        if (j < reach()) stats[7]++;            // a fast routine read beyond its reach (must stay 0)
        if (in_hist(j)) { stats[3]++; if (dbg) dbg[curkind]++; return ring[j % DPC_HIST]; }
        stats[2]++;
        const int tb = tb_[j];
        return DpcHist{tb == -1 ? -__builtin_huge_val() : score_[j], tb == -1 ? -1 : ndx_[tb]};
    }
    int ndx_of(int j) const { return ndx_[j]; }
    void hist_put(int i, const DpcHist& h) { ring[i % DPC_HIST] = h; }
    DpwS src(int j) {
        stats[2]++;
        DpwS s;
        s.j = j; s.kind = DPW_KIND(kf_[j]); s.frame = DPW_FRAME(kf_[j]); s.ndx = ndx_[j]; s.stop_val = stopv_[j];
        const int tb = tb_[j];
        s.score = score_[j]; s.tbn = tb == -1 ? -1 : ndx_[tb]; s.cs = cs_[j];
        s.vm = 0; s.x0 = s.x1 = s.x2 = 0.0;
        if (s.kind == 1) { s.vm = ext_[j].vm; s.x0 = ext_[j].x[0]; s.x1 = ext_[j].x[1]; s.x2 = ext_[j].x[2]; }
        return s;
    }
    void cand_put(int f, int k, int idx, int ndx, double sv, int tbn) { c_idx[f][k] = idx; c_ndx[f][k] = ndx; c_val[f][k] = DpcHist{sv, tbn}; }
    int cand_idx(int f, int k) const { return c_idx[f][k]; }
    int cand_ndx(int f, int k) const { return c_ndx[f][k]; }
    DpcHist cand_val(int f, int k) { stats[4]++; if (dbg) dbg[4 + curkind]++; return c_val[f][k]; }
    double igm(int d) const { return igm_[d]; }
    bool any(bool p) const { return p; }
};

}  // namespace

// stats: [0] nodes that went through the slow routine ([1] beyond the history's reach, [5] a running maximum cut by the window, [6] no list), [2] finished nodes read back from memory, [3] ... from the history,
// [4] candidate-list entries evaluated, [7] history reads beyond its reach (must be 0)
extern "C" int dpc_model_run(int n, const int32_t* ndx, const int32_t* stop_val, const uint8_t* type, const int8_t* strand,
                             const double* cscore, const double* sscore, const double* rscore, const double* uscore,
                             const int32_t* star_ptr, double st_wt, double* score, int32_t* traceb, int8_t* ov_mark,
                             int32_t* max_index, int64_t* stats /* [8] */) {
    double igm[64] = {0};
    for (int d = 0; d <= DPW_OPER_DIST; d++) igm[d] = (2.0 - ((double)d / DPW_OPER_DIST)) * 0.15 * st_wt;
    const DpwModel M{st_wt, -0.15 * st_wt, igm};
    for (int k = 0; k < 8; k++) stats[k] = 0;
    *max_index = -1;
    if (n <= 0) return 0;
    std::vector<uint8_t> kf(n); std::vector<int32_t> lo(n), q1(n), q2(n); std::vector<double> cs(n); std::vector<DpwExt> ext(n);
    for (int i = 0; i < n; i++) {
        const DpwTopo t = dpw_topo_node(ndx, stop_val, type, strand, n, i);
        kf[i] = t.kf; lo[i] = t.lo; q1[i] = t.q1; q2[i] = t.q2;
    }
    for (int i = 0; i < n; i++) {
        cs[i] = cscore[i] + sscore[i];
        const int kind = DPW_KIND(kf[i]);
        if (kind & 1) dpw_chain_ext(ndx, stop_val, strand, q2.data(), cscore, sscore, rscore, uscore, star_ptr, i, kind == 3, M, ext[i]);
    }
    HostX X;
    memset(X.ring, 0, sizeof X.ring);
    X.kf_ = kf.data(); X.ndx_ = ndx; X.stopv_ = stop_val; X.tb_ = traceb; X.score_ = score; X.cs_ = cs.data(); X.ext_ = ext.data(); X.stats = stats; X.igm_ = igm;
    static long long dbg_counts[12]; if (getenv("DPC_MODEL_DEBUG")) X.dbg = dbg_counts;
    DpcRegs R; DpcUni U;
    dpc_init(R, U, X);
    for (int i = 0; i < n; i++) {
        const DpcNode N{i, DPW_KIND(kf[i]), DPW_FRAME(kf[i]), kf[i], ndx[i], stop_val[i], lo[i], q1[i], q2[i]};
        DpcExt E; memset(&E, 0, sizeof E);
        E.cq[0] = E.cq[1] = E.cq[2] = DPW_NONE;
        if (N.kind & 1) {
            const DpwExt& e = ext[i];
            E.vm = e.vm;
            for (int q = 0; q < 3; q++) { E.x[q] = e.x[q]; E.n3n[q] = e.n3n[q]; E.n3s[q] = e.n3s[q]; E.cq[q] = e.cq[q]; }
        }
        DpcOut B{0.0, -1, -1, 0.0, -1};
        X.cur = i; X.curkind = N.kind; if (X.dbg) X.dbg[8 + N.kind]++;
        // the kernel's dispatch: candidates (fast routine of the node's kind, or the slow one), then what the node leaves
        if (N.kind == 0) {
            if (dpc_need_slow_begin(R, U, N, X)) { stats[0]++; stats[U.fp < X.reach() ? 1 : 5]++; dpc_cand_slow(R, U, N, cs[i], E, M, X, B); } else dpc_cand_f5(R, U, N, M, X, B);
            dpc_finish_f5(N, cs[i], X.carry(N.frame), X, B);
        } else if (N.kind == 1) {
            dpc_cand_f3(X.carry(N.frame), B);
            dpc_finish_f3(R, U, N, E, X, B);
        } else if (N.kind == 2) {
            if (dpc_need_slow_r5(U, N, X)) { stats[0]++; stats[6]++; dpc_cand_slow(R, U, N, cs[i], E, M, X, B); } else dpc_cand_r5(U, N, cs[i], M.negc, X, B);
            dpc_finish_r5(R, N, M.negc, B);
        } else {
            if (dpc_need_slow_begin(R, U, N, X) || dpc_need_slow_r3(U, N, E, X)) { stats[0]++; stats[U.fp < X.reach() ? 1 : (dpc_need_slow_begin(R, U, N, X) ? 5 : 6)]++; dpc_cand_slow(R, U, N, cs[i], E, M, X, B); }
            else dpc_cand_r3(R, U, N, E, M, X, B);
            dpc_finish_r3(U, N, X, B);
        }
        score[i] = B.val; traceb[i] = B.tb; ov_mark[i] = (int8_t)B.ov;
        X.hist_put(i, DpcHist{B.sv, B.tbn});
    }
    if (X.dbg) { fprintf(stderr, "[dpc model] events so far:"); for (int k = 0; k < 12; k++) fprintf(stderr, " %lld", dbg_counts[k]); fprintf(stderr, "\n"); }
    *max_index = R.end_idx;
    return 0;
}
