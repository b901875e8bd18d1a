// Host model of the contig-per-wavefront connection scorer (pyrodigal_amd/csrc/dp_contig.hip): the step function of dpc_core.h --
// the same source the kernel compiles -- run node by node over one chain (one lane of the kernel's wavefront), with plain arrays
// where the kernel has topology registers, an LDS history and the chain's arrays in HBM.  TEST INFRASTRUCTURE: built and run by
// tests/test_dpc_model.py against the CPU oracle; nothing in the product links it.
#include "../pyrodigal_amd/csrc/dpc_core.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

namespace {

struct HostX {
    const uint8_t* kf_; const int32_t* ndx_; const int32_t* stopv_; const int32_t* tb_; const double* score_; const double* cs_; const DpwExt* ext_;
    const double* igm_;
    int64_t* stats;
    int cur;                                    // the node being walked
    DpcHist ring[DPC_HIST];
    int c_idx[3][DPC_CAND], c_ndx[3][DPC_CAND]; DpcHist c_val[3][DPC_CAND];
    DpcCarry carry_[3]; double l3v_[3];
    DpcCarry carry(int f) const { return carry_[f]; }
    void set_carry(int f, const DpcCarry& c) { carry_[f] = c; }
    double l3v(int f) const { return l3v_[f]; }
    void set_l3v(int f, double v) { l3v_[f] = v; }
    DpcHist hist(int j) {
        if (j <= cur - DPC_HIST - 1 || j >= cur) stats[7]++;          // a fast routine read beyond the history (must stay 0)
        stats[3]++;
        return ring[j % DPC_HIST];
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
    DpcHist cand_val(int f, int k) { stats[4]++; return c_val[f][k]; }
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
    // the contig's topology, compiled (the kernel's k_dpc_compile): one record per node
    std::vector<DpcProg> prog(n);
    {
        int l3[3] = {-1, -1, -1};
        for (int i = 0; i < n; i++) {
            dpc_compile_node(ndx, stop_val, kf.data(), lo.data(), q1.data(), q2.data(), i, l3[0], l3[1], l3[2], prog[i]);
            if (DPW_KIND(kf[i]) == 3) l3[DPW_FRAME(kf[i])] = i;
        }
    }
    DpcRegs R;
    dpc_init(R, X);
    for (int i = 0; i < n; i++) {
        const DpcProg& P = prog[i];
        const int kind = dpc_prog_kind(P);
        const DpcNode N{i, kind, dpc_prog_frame(P), kf[i], ndx[i], stop_val[i], lo[i], q1[i], q2[i]};
        DpcExt E; memset(&E, 0, sizeof E);
        E.cq[0] = E.cq[1] = E.cq[2] = DPW_NONE;
        if (kind & 1) {
            const DpwExt& e = ext[i];
            E.vm = e.vm;
            for (int q = 0; q < 3; q++) { E.x[q] = e.x[q]; E.n3n[q] = e.n3n[q]; E.n3s[q] = e.n3s[q]; E.cq[q] = e.cq[q]; }
        }
        DpcOut B{0.0, -1, -1, 0.0, -1};
        X.cur = i;
        // the kernel's dispatch: candidates (fast routine of the node's kind, or the slow one), then what the node leaves
        if (kind == 0) {
            if (dpc_need_slow_begin(R, P, X)) { stats[0]++; stats[(P.w[0] & DPC_F_SLOW) ? 1 : 5]++; dpc_cand_slow(R, N, cs[i], E, M, X, B); } else dpc_cand_f5(R, P, i, M, X, B);
            dpc_finish_f5(P, i, cs[i], X.carry(N.frame), X, B);
        } else if (kind == 1) {
            dpc_cand_f3(X.carry(N.frame), B);
            dpc_finish_f3(R, P, i, E, X, B);
        } else if (kind == 2) {
            if (P.w[0] & DPC_F_SLOW) { stats[0]++; stats[6]++; dpc_cand_slow(R, N, cs[i], E, M, X, B); } else dpc_cand_r5(P, cs[i], M.negc, X, B);
            dpc_finish_r5(R, i, M.negc, B);
        } else {
            const bool s1 = dpc_need_slow_begin(R, P, X);
            if (s1 || dpc_need_slow_r3(P, i, E, X)) { stats[0]++; stats[(P.w[0] & DPC_F_SLOW) ? 1 : (s1 ? 5 : 6)]++; dpc_cand_slow(R, N, cs[i], E, M, X, B); }
            else dpc_cand_r3(R, P, i, E, M, X, B);
            dpc_finish_r3(P, i, X, B);
        }
        score[i] = B.val; traceb[i] = B.tb; ov_mark[i] = (int8_t)B.ov;
        X.hist_put(i, DpcHist{B.sv, B.tbn});
    }
    *max_index = R.end_idx;
    return 0;
}
