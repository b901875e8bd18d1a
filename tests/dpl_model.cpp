// Host model of the lane-per-chain connection scorer (pyrodigal_amd/csrc/dp_lane.hip): the step function of dpl_core.h -- the
// same source the kernel compiles -- run node by node over one chain, with plain arrays where the kernel has LDS rings and the
// chain's arrays in HBM.  TEST INFRASTRUCTURE: built and run by tests/test_dpl_model.py against the CPU oracle; nothing in the
// product links it.
#include "../pyrodigal_amd/csrc/dpl_core.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

namespace {

struct HostX {
    DplEnt r5[DPL_R5_RING], f3[DPL_F3_RING]; int f3t[DPL_F3_RING]; DplCand cand[3][DPL_CAND];
    const uint8_t* kf_; const int32_t* ndx_; const int32_t* q2_; const int32_t* tbn_; const int32_t* tb_; const double* score_;
    int64_t* stats; const double* igm_;
    double igm(int d) const { return igm_[d]; }
    DplEnt r5_get(int s) const { return r5[s]; }
    void r5_put(int s, const DplEnt& e) { r5[s] = e; }
    DplEnt f3_get(int s) const { return f3[s]; }
    void f3_put(int s, const DplEnt& e) { f3[s] = e; }
    int f3t_get(int s) const { return f3t[s]; }
    void f3t_put(int s, int v) { f3t[s] = v; }
    DplCand cand_get(int f, int k) const { stats[3]++; return cand[f][k]; }
    void cand_put(int f, int k, const DplCand& c) { cand[f][k] = c; }
    DplFin fin(int j) const { stats[2]++; return DplFin{ndx_[j], q2_[j], kf_[j], tbn_[j], tb_[j], score_[j]}; }
    void note(int k) { stats[k == 2 ? 6 : k]++; }
};

}  // namespace

// stats: [0] gene begins that went through the window scan, [1] gene begins that read their near gene ends back after a ring overflow, [2] finished nodes read back
// from memory, [3] candidate-list entries evaluated, [4] most reverse starts in the ring, [5] most forward stops in the ring, [6] reverse targets
// that walked their chain of overlap candidates in memory (list not applicable or overflowed)
extern "C" int dpl_model_run(int n, const int32_t* ndx, const int32_t* stop_val, const uint8_t* type, const int8_t* strand,
                             const double* cscore, const double* sscore, const double* rscore, const double* uscore,
                             const int32_t* star_ptr, double st_wt, double* score, int32_t* traceb, int8_t* ov_mark,
                             int32_t* max_index, int64_t* stats /* [8] */) {
    double igm[64] = {0};
    for (int d = 0; d <= DPW_OPER_DIST; d++) igm[d] = (2.0 - ((double)d / DPW_OPER_DIST)) * 0.15 * st_wt;
    const DpwModel M{st_wt, -0.15 * st_wt, igm};
    for (int k = 0; k < 8; k++) stats[k] = 0;
    *max_index = -1;
    if (n <= 0) return 0;
    std::vector<uint8_t> kf(n); std::vector<int32_t> lo(n), q1(n), q2(n), tbn(n, -1); std::vector<double> cs(n); std::vector<DpwExt> ext(n);
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
    memset(X.r5, 0, sizeof X.r5); memset(X.f3, 0, sizeof X.f3);
    X.kf_ = kf.data(); X.ndx_ = ndx; X.q2_ = q2.data(); X.tbn_ = tbn.data(); X.tb_ = traceb; X.score_ = score; X.stats = stats; X.igm_ = igm;
    DplState S;
    dpl_init(S);
    for (int i = 0; i < n; i++) {
        DpwT T; memset(&T, 0, sizeof T);
        T.i = i; T.kind = DPW_KIND(kf[i]); T.frame = DPW_FRAME(kf[i]); T.ndx = ndx[i]; T.stop_val = stop_val[i]; T.lo = lo[i]; T.q1 = q1[i]; T.q2 = q2[i];
        T.cs = cs[i]; T.csd = T.cs + M.negc;
        T.cq0 = T.cq1 = T.cq2 = DPW_NONE;
        if (T.kind & 1) {
            const DpwExt& e = ext[i];
            T.vm = e.vm; T.x0 = e.x[0]; T.x1 = e.x[1]; T.x2 = e.x[2];
            T.n3n0 = e.n3n[0]; T.n3n1 = e.n3n[1]; T.n3n2 = e.n3n[2]; T.n3s0 = e.n3s[0]; T.n3s1 = e.n3s[1]; T.n3s2 = e.n3s[2];
            T.cq0 = e.cq[0]; T.cq1 = e.cq[1]; T.cq2 = e.cq[2];
        }
        DpwBest B;
        dpl_step(S, T, kf[i], M, X, B);
        if (S.r5_cnt > stats[4]) stats[4] = S.r5_cnt;
        if (S.f3_cnt > stats[5]) stats[5] = S.f3_cnt;
        score[i] = B.val; traceb[i] = B.tb; ov_mark[i] = (int8_t)B.ov; tbn[i] = B.tbn;
    }
    *max_index = S.end_idx;
    return 0;
}
