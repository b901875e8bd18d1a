// Contig-per-wavefront connection scoring (dp_contig.hip): ONE wavefront walks ONE contig node by node and its lanes are the
// MODELS scored on that contig (the chains of one translation-table group).  Every model of a contig walks the same topology --
// kinds, frames, positions, windows, candidate links never depend on a model -- so everything that is
// topology is wave-uniform here: it lives in scalar registers, every branch on it is a scalar branch, a node only runs the
// code of its own kind, and loops over near nodes / candidates have the same trip count in every lane.  What differs per lane
// is values: scores, the extras of a stop node (through star_ptr), running maxima and their argmax.
//
// The case analysis is the lane kernel's (dpl_core.h, pinned against the plain restatement of the reference on the CPU): every class of candidates of a node comes
// from a running structure that costs O(1) per node.  What changes with a uniform topology:
//   * the "rings" of near gene ends are not copies but an index range: the nodes [fp, i) not folded into the far maxima yet,
//     fp = the first node within 3 * OPER_DIST bases of the last gene begin (DpwTopo::q1); their values come from a short
//     per-lane history of finished nodes (LDS), which is also where the results wait to be written out in whole lines;
//   * EVERY gene end is a ring / list entry, reached or not (an entry that was never reached carries -inf): which entries
//     exist must not depend on the lane;
//   * the last reverse stop of a frame and the list of forward stops that can overlap the 3' end of its genes are uniform
//     records (index, position), per-lane values;
//   * the position of a node's traceb node only matters when the node is a forward stop (towards reverse targets), where it
//     is the position of the best start of its ORF: only the forward carries keep it.
// Where the state lives: the far maxima and the best gene end so far in registers (DpcRegs: every node touches them); what is
// indexed by a frame -- the forward carries, the score of a frame's last reverse stop -- behind the accessor (the kernel keeps
// them in LDS, [frame][lane]: a node reads its frame's record at an address that is a scalar offset, no selects, no copy of the
// code per frame); the uniform records in DpcUni.
// A node is done in two halves: its candidates -> B (dpc_cand_*: the FAST routines, which assume that the nodes they read are
// within reach of the history and that the candidate lists apply -- dpc_need_slow_* say when they do not -- or dpc_cand_slow:
// the reference's own loop over the whole window, pair by pair, with every source read back from memory: exact for any node,
// rare, and the only place that knows about windows that cut running maxima, overflowed lists and deep near zones), then
// dpc_finish_*: what the node leaves for later ones.  Written once for the device and the host: tests/dpc_model.cpp runs the
// same routines model by model against the plain restatement of the reference's loop on the CPU.
//
// Same recurrence (ref: lib.pyx:1205-1237 `_score_connections`, _connection.h:94-408, impl/generic.h:29-36):
//     score[i] = max(0, max_j (score[j] + w(j, i))) over the window [lo_i, i), ties -> largest j.
#pragma once

#include "dpl_core.h"

// loops over the three frames must be unrolled on the device: an index that is not a constant sends the arrays they walk to scratch memory
#if defined(__HIPCC__)
#define DPC_UNROLL _Pragma("unroll")
#else
#define DPC_UNROLL
#endif

#ifndef DPC_CAND
#define DPC_CAND 6          // forward stops kept per reverse frame (at most 7: three bits of DpcUni::cn)
#endif
#ifndef DPC_HIST
#define DPC_HIST 32         // nodes of history a lane keeps (a power of two)
#endif

// a node's topology (wave-uniform)
struct DpcNode { int i, kind, frame, kfb, ndx, stop_val, lo, q1, q2; };
// what a lane keeps of a finished gene end: its score as a SOURCE (-inf: never reached) and, for a forward stop, the position of
// its traceb node
struct DpcHist { double sv; int tbn; };
// lexicographic (value, index) running maximum
struct DpcMax { double v; int i; };
// forward carry of a frame: best start / operon offer since the frame's last stop, its index and position
struct DpcCarry { double v; int i, n; };
// a node's result: score, traceb, ov_mark, and what the history keeps of it
struct DpcOut { double val; int tb, ov; double sv; int tbn; };
// the extras of a stop node (DpwExt), per lane
struct DpcExt { double x[3]; int n3n[3], n3s[3], cq[3], vm; };

// per lane, in registers
struct DpcRegs {
    DpcMax r5_all;              // a over every reached reverse start so far
    DpcMax r5_far, f3_far;      // a over the reached reverse starts / forward stops before fp (more than 180 bases behind)
    double end_best; int end_idx, end_tb;
};
// uniform (the last reverse stop of each frame -- index, stop_val, position -- is uniform as well and lives behind the accessor:
// indexed by a frame, it would drag this struct into scratch memory on the device)
struct DpcUni {
    int fp;                     // nodes before fp are folded into the far maxima
    int cn;                     // per reverse frame f: bits 4f .. 4f+2 = entries of its candidate list, bit 4f+3 = the list is incomplete
};

// What the routines need from their surroundings (the kernel: topology of the batch in registers; history, lists and the
// per-frame records in LDS; everything else in HBM; the host model: plain arrays):
//   int reach()                 the oldest node whose topology is at hand (uniform): the fast routines only look at nodes from there on
//   for_near(a, b, kind, f)     f(j, ndx_j) for the nodes j of [a, b), ascending, of the ONE kind named (DPC_K_F3 or DPC_K_R5)
//                               (a, b uniform, a >= reach())
//   DpcHist hist(j)             a finished gene end j >= reach(), j uniform (the last DPC_HIST nodes from the history, older ones
//                               read back from memory)
//   int ndx_of(j)               position of node j >= reach() (uniform)
//   DpcCarry carry(f) / set_carry(f, c);  double l3v(f) / set_l3v(f, v)         the per-frame records (f uniform), per lane
//   int l3i(f), l3s(f), l3n(f) / set_l3(f, i, s, n)     the last reverse stop of frame f: index (-1: none), stop_val, position (uniform)
//   cand_put(f, k, idx, ndx, sv, tbn) / cand_idx(f, k) / cand_ndx(f, k) (uniform) / cand_val(f, k) -> DpcHist (per lane)
//   double igm(d)               the intergenic term at distance 0 <= d <= OPER_DIST (d uniform)
//   DpwS src(j)                 ANY finished node j < i as a source, read back from memory (slow path; j uniform):
//                               kind, frame, ndx, stop_val, score, tbn (-1: never reached), cs, vm, x0..x2
//   bool any(p)                 p holds in some lane of the wave
#define DPC_K_F3 1
#define DPC_K_R5 2

template <class X>
DPW_HD void dpc_init(DpcRegs& R, DpcUni& U, X& x) {
    const double NI = -__builtin_huge_val();
    R.r5_all = DpcMax{NI, -1}; R.r5_far = DpcMax{NI, -1}; R.f3_far = DpcMax{NI, -1};
    R.end_best = -1.0; R.end_idx = -1; R.end_tb = -1;
    U.fp = 0; U.cn = 0;
    DPC_UNROLL for (int f = 0; f < 3; f++) { x.set_l3(f, -1, 0, 0); x.set_carry(f, DpcCarry{NI, -1, -1}); x.set_l3v(f, 0.0); }
}

// ascending take: candidates arrive in index order, so ">=" is the whole tie rule
DPW_HD void dpc_max_take_asc(DpcMax& m, const bool ok, const double v, const int i) {
    const bool t = ok & (v >= m.v);
    m.v = t ? v : m.v; m.i = t ? i : m.i;
}
// a candidate later in the chain than everything taken so far: ">=" is the reference's rule (ref: _connection.h:135-139)
DPW_HD void dpc_take_asc(DpcOut& B, const bool ok, const double v, const int j) {
    const bool t = ok & (v >= B.val);
    B.val = t ? v : B.val; B.tb = t ? j : B.tb; B.ov = t ? -1 : B.ov;
}
// a candidate from anywhere: the lexicographic (value, index) rule
DPW_HD void dpc_take_lex(DpcOut& B, const bool ok, const double v, const int j) {
    const bool t = ok & ((v > B.val) | ((v == B.val) & (j > B.tb)));
    B.val = t ? v : B.val; B.tb = t ? j : B.tb; B.ov = t ? -1 : B.ov;
}

// nodes [U.fp, upto) leave the near zone: the reached gene ends among them join the far maxima (each kind in chain order)
template <class X>
DPW_HD void dpc_fold(DpcRegs& R, DpcUni& U, const int upto, const double negc, X& x) {
    const double NI = -__builtin_huge_val();
    if (upto > U.fp) {
        x.for_near(U.fp, upto, DPC_K_R5, [&](const int j, const int) { const DpcHist h = x.hist(j); dpc_max_take_asc(R.r5_far, h.sv > NI, h.sv + negc, j); });
        x.for_near(U.fp, upto, DPC_K_F3, [&](const int j, const int) { const DpcHist h = x.hist(j); dpc_max_take_asc(R.f3_far, h.sv > NI, h.sv + negc, j); });
        U.fp = upto;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// When the fast routines do not apply to node N (uniform over the wave: `any` of the lanes is enough to send all of them
// through the slow routine, which is exact for every lane):
//   * a gene begin whose unfolded range reaches back beyond the nodes whose topology is at hand (the batch before this one);
//   * a gene begin whose window start has passed the argmax of a running maximum it reads (it only can from node 1000 on);
//   * a reverse node whose overlap candidates are not the list of its frame (its stop is not the frame's last reverse stop --
//     the window cut it off --, or the list overflowed) while the static chain of candidates is not empty.
template <class X>
DPW_HD bool dpc_need_slow_begin(const DpcRegs& R, const DpcUni& U, const DpcNode& N, X& x) {
    if (U.fp < x.reach()) return true;
    if (N.lo <= 0) return false;
    const DpcMax& rmax = N.kind == 3 ? R.r5_far : R.r5_all;
    return x.any(((rmax.i >= 0) & (rmax.i < N.lo)) | ((R.f3_far.i >= 0) & (R.f3_far.i < N.lo)));
}
template <class X>
DPW_HD bool dpc_need_slow_r5(const DpcUni& U, const DpcNode& N, X& x) {
    const int f = N.frame;
    const int c4 = (U.cn >> (4 * f)) & 15;
    const bool list = x.l3i(f) >= 0 && x.l3n(f) == N.stop_val && !(c4 & 8);
    return !list && N.q2 < N.i;
}
template <class X>
DPW_HD bool dpc_need_slow_r3(const DpcUni& U, const DpcNode& N, const DpcExt& e, X& x) {
    bool slow = false;
    DPC_UNROLL for (int q = 0; q < 3; q++) {
        const int c4 = (U.cn >> (4 * q)) & 15;
        const bool list = (x.l3i(q) >= 0) & (x.l3n(q) == e.n3s[q]) & !(c4 & 8);
        slow = slow | (((e.vm >> q) & 1) & !list & (e.cq[q] < N.i));
    }
    return x.any(slow);
}

// the target of a pair as dpw_pair wants it
DPW_HD DpwT dpc_target(const DpcNode& N, const double cs, const double negc, const DpcExt& e) {
    DpwT T;
    T.i = N.i; T.kind = N.kind; T.frame = N.frame; T.ndx = N.ndx; T.stop_val = N.stop_val; T.lo = N.lo; T.q1 = N.q1; T.q2 = N.q2;
    T.cs = cs; T.csd = cs + negc;
    T.vm = 0; T.x0 = T.x1 = T.x2 = 0.0;
    T.n3n0 = T.n3n1 = T.n3n2 = T.n3s0 = T.n3s1 = T.n3s2 = 0; T.cq0 = T.cq1 = T.cq2 = DPW_NONE;
    if (N.kind & 1) {
        T.vm = e.vm; T.x0 = e.x[0]; T.x1 = e.x[1]; T.x2 = e.x[2];
        T.n3n0 = e.n3n[0]; T.n3n1 = e.n3n[1]; T.n3n2 = e.n3n[2]; T.n3s0 = e.n3s[0]; T.n3s1 = e.n3s[1]; T.n3s2 = e.n3s[2];
        T.cq0 = e.cq[0]; T.cq1 = e.cq[1]; T.cq2 = e.cq[2];
    }
    return T;
}

// The slow routine: every node of the window against this node, pair by pair, in chain order -- the reference's loop
// (ref: lib.pyx:1221-1237 over _connection.h:94-408).  For a gene begin the running maxima are rebuilt over the window on the way
// and the unfolded range restarts at q1 (what lies before the window is dropped: windows of gene begins only move forward).
template <class X>
DPW_HD void dpc_cand_slow(DpcRegs& R, DpcUni& U, const DpcNode& N, const double cs, const DpcExt& e, const DpwModel& M, X& x, DpcOut& B) {
    const double NI = -__builtin_huge_val();
    const DpwT T = dpc_target(N, cs, M.negc, e);
    const bool begin = N.kind == 0 || N.kind == 3;
    const int fp = begin ? (N.q1 > U.fp ? N.q1 : U.fp) : U.fp;
    if (begin) { R.r5_all = DpcMax{NI, -1}; R.r5_far = DpcMax{NI, -1}; R.f3_far = DpcMax{NI, -1}; U.fp = fp; }
    DpwBest W{0.0, -1, -1, -1};
    for (int j = N.lo; j < N.i; j++) {
        const DpwS s = x.src(j);
        bool ok; double w; int mf;
        dpw_pair(s, T, M, ok, w, mf);
        dpw_take(W, ok, s.score + w, j, mf, s.ndx);
        if (begin && (s.kind == 1 || s.kind == 2)) {
            const bool reached = s.tbn != -1;
            const double a = s.score + M.negc;
            if (s.kind == 2) { dpc_max_take_asc(R.r5_all, reached, a, j); dpc_max_take_asc(R.r5_far, reached & (j < fp), a, j); }
            else dpc_max_take_asc(R.f3_far, reached & (j < fp), a, j);
        }
    }
    B.val = W.val; B.tb = W.tb; B.ov = W.ov;
}

// ------------------------------------------------------------------------------------------------------------------------
// Fast candidates.

// a forward start: every gene end of the window (ref: _connection.h:117-130)
template <class X>
DPW_HD void dpc_cand_f5(DpcRegs& R, DpcUni& U, const DpcNode& N, const DpwModel& M, X& x, DpcOut& B) {
    dpc_fold(R, U, N.q1, M.negc, x);
    // forward stops, in chain order: the far ones, then those within 3 * OPER_DIST bases pair by pair
    dpc_take_asc(B, true, R.f3_far.v, R.f3_far.i);
    x.for_near(U.fp > N.lo ? U.fp : N.lo, N.i, DPC_K_F3, [&](const int j, const int nj) {
        const int d = N.ndx - nj;
        if (d > 2) {
            const DpcHist h = x.hist(j);
            const double w = d > DPL_NEAR ? M.negc : (d <= DPW_OPER_DIST ? x.igm(d) : 0.0);
            dpc_take_asc(B, true, h.sv + w, j);
        }
    });
    // every reverse start so far: the weight towards a forward start never depends on the distance
    dpc_take_lex(B, true, R.r5_all.v, R.r5_all.i);
}

// a forward stop: the best start / operon partner of its ORF (ref: :166-188) -- the frame's carry
DPW_HD void dpc_cand_f3(const DpcCarry& c, DpcOut& B) {
    const bool reached = (c.i >= 0) & (c.v >= 0.0);
    B.val = reached ? c.v : 0.0; B.tb = reached ? c.i : -1; B.tbn = reached ? c.n : -1;
}

// A reached forward stop of a candidate list against a reverse start (ref: _connection.h:238-254; dpl_cand_r5 with the uniform part
// of the test hoisted): rel = c.ndx - stop_val
DPW_HD void dpc_list_r5(DpcOut& B, const DpcNode& N, const double csd, const double c_sv, const int c_tbn, const int c_idx, const int c_ndx) {
    const int rel = c_ndx - N.stop_val;
    const bool uni = (c_idx >= N.lo) & (rel > -4) & (rel + 5 < DPW_MAX_OPP_OVLP) & (rel < N.ndx - c_ndx + 3);
    if (uni) dpc_take_asc(B, c_tbn < N.stop_val - 3 - rel, c_sv + csd, c_idx);
}
// ... and against a reverse stop, through the best admissible overlapping start (ref: :288-336): the first q with the largest x
DPW_HD void dpc_list_r3(DpcOut& B, const DpcNode& N, const DpcExt& e, const bool have, const double negc, const double c_sv, const int c_tbn,
                        const int c_idx, const int c_ndx) {
    const int left = c_ndx + 2;
    if ((c_idx >= N.lo) & (left < N.ndx - 2)) {
        double maxval = 0.0; int mf = -1;
        DPC_UNROLL for (int q = 0; q < 3; q++) {
            const int ovlp = left - e.n3s[q] + 3;
            const bool tk = ((e.vm & (1 << q)) != 0) & (ovlp > 0) & (ovlp < DPW_MAX_OPP_OVLP) & (ovlp < e.n3n[q] - left) & (ovlp < e.n3s[q] - c_tbn - 2) & (e.x[q] > maxval);
            maxval = tk ? e.x[q] : maxval; mf = tk ? q : mf;
        }
        const double v = c_sv + (mf != -1 ? maxval : negc);
        const bool t = have & ((v > B.val) | ((v == B.val) & (c_idx > B.tb)));
        B.val = t ? v : B.val; B.tb = t ? c_idx : B.tb; B.ov = t ? mf : B.ov;
    }
}

// a reverse start of frame f: its own stop (ref: :228-235) and the forward stops overlapping its gene's 3' end (ref: :238-254)
template <class X>
DPW_HD void dpc_cand_r5(const DpcUni& U, const DpcNode& N, const double cs, const double negc, X& x, DpcOut& B) {
    const int f = N.frame;
    const int l3i = x.l3i(f), l3s = x.l3s(f), l3n = x.l3n(f);
    if ((l3i >= 0) & (l3i >= N.lo) & (l3s > N.ndx)) dpc_take_asc(B, true, x.l3v(f) + cs, l3i);
    const int c4 = (U.cn >> (4 * f)) & 15;
    if ((l3i >= 0) & (l3n == N.stop_val) & !(c4 & 8)) {
        // (the list follows the frame's last reverse stop: in chain order, after the stop itself)
        const double csd = cs + negc;
        for (int k = 0; k < (c4 & 7); k++) {
            const DpcHist v = x.cand_val(f, k);
            dpc_list_r5(B, N, csd, v.sv, v.tbn, x.cand_idx(f, k), x.cand_ndx(f, k));
        }
    }
}

// a reverse stop: every gene end of the window (ref: :288-342), the reverse stop whose ORF covers it (ref: :345-356)
template <class X>
DPW_HD void dpc_cand_r3(DpcRegs& R, DpcUni& U, const DpcNode& N, const DpcExt& e, const DpwModel& M, X& x, DpcOut& B) {
    dpc_fold(R, U, N.q1, M.negc, x);
    // the far gene ends (the weight is the constant): both kinds, either order
    dpc_take_asc(B, true, R.r5_far.v, R.r5_far.i);
    dpc_take_lex(B, true, R.f3_far.v, R.f3_far.i);
    // near gene ends, pair by pair: reverse starts with the distance term, forward stops the plain connection (a forward stop's
    // offer through an overlapping start of this stop is met on the candidate lists, and is larger)
    const int a0 = U.fp > N.lo ? U.fp : N.lo;
    x.for_near(a0, N.i, DPC_K_R5, [&](const int j, const int nj) {
        const int d = N.ndx - nj;
        if (d > 2) {
            const DpcHist h = x.hist(j);
            const double w = d > DPL_NEAR ? M.negc : (d <= DPW_OPER_DIST ? x.igm(d) : 0.0);
            dpc_take_lex(B, true, h.sv + w, j);
        }
    });
    x.for_near(a0, N.i, DPC_K_F3, [&](const int j, const int nj) {
        if (N.ndx - nj > 4) { const DpcHist h = x.hist(j); dpc_take_lex(B, true, h.sv + M.negc, j); }
    });
    // the reverse stop whose ORF covers this one, per frame of an overlapping start: an operon (ref: :345-356)
    DPC_UNROLL for (int q = 0; q < 3; q++)
        if ((x.l3i(q) >= 0) & (x.l3i(q) >= N.lo) & (x.l3s(q) > N.ndx)) dpc_take_lex(B, (e.vm & (1 << q)) != 0, x.l3v(q) + e.x[q], x.l3i(q));
    // forward stops that overlap the 3' end of the gene of an overlapping start (the start of frame q has its stop at n3s[q]):
    // the list of frame q where it is that stop's
    DPC_UNROLL for (int q = 0; q < 3; q++) {
        const int c4 = (U.cn >> (4 * q)) & 15;
        if ((x.l3i(q) >= 0) & !(c4 & 8)) {
            const bool have = ((e.vm & (1 << q)) != 0) & (x.l3n(q) == e.n3s[q]);
            for (int k = 0; k < (c4 & 7); k++) {
                const DpcHist v = x.cand_val(q, k);
                dpc_list_r3(B, N, e, have, M.negc, v.sv, v.tbn, x.cand_idx(q, k), x.cand_ndx(q, k));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// What a finished node leaves for later ones.
DPW_HD void dpc_note_end(DpcRegs& R, const DpcOut& B, const int i) {
    const bool new_end = B.val >= R.end_best;
    R.end_best = new_end ? B.val : R.end_best; R.end_idx = new_end ? i : R.end_idx; R.end_tb = new_end ? B.tb : R.end_tb;
}
template <class X>
DPW_HD void dpc_cand_push(DpcUni& U, X& x, const int f, const int idx, const int ndx, const double sv, const int tbn) {
    const int c4 = (U.cn >> (4 * f)) & 15;
    if (!(c4 & 8)) {
        if ((c4 & 7) == DPC_CAND) U.cn |= 8 << (4 * f);
        else { x.cand_put(f, c4 & 7, idx, ndx, sv, tbn); U.cn += 1 << (4 * f); }
    }
}
// a forward start offers score + cs to the stop of its ORF (a later node wins a tie); `c`: the carry of its frame
template <class X>
DPW_HD void dpc_finish_f5(const DpcNode& N, const double cs, const DpcCarry& c, X& x, DpcOut& B) {
    const double g = B.val + cs;
    const bool t = g >= c.v;
    x.set_carry(N.frame, DpcCarry{t ? g : c.v, t ? N.i : c.i, t ? N.ndx : c.n});
    B.sv = B.val; B.tbn = -1;
}
// a forward stop restarts the running maximum of its own frame and, when reached, offers score + x to the frames whose next stop's
// ORF holds it (operon partners); it may overlap the 3' end of the reverse genes that end at the last reverse stop of a frame
template <class X>
DPW_HD void dpc_finish_f3(DpcRegs& R, DpcUni& U, const DpcNode& N, const DpcExt& e, X& x, DpcOut& B) {
    const double NI = -__builtin_huge_val();
    const bool reached = B.tb != -1;
    B.sv = reached ? B.val : NI;
    dpc_note_end(R, B, N.i);
    x.set_carry(N.frame, DpcCarry{NI, -1, -1});
    DPC_UNROLL for (int q = 0; q < 3; q++) {
        if (DPW_INORF(N.kfb, q)) {
            const DpcCarry c = x.carry(q);
            const double o = B.val + e.x[q];
            const bool t = reached & ((e.vm & (1 << q)) != 0) & (o >= c.v);
            x.set_carry(q, DpcCarry{t ? o : c.v, t ? N.i : c.i, t ? N.ndx : c.n});
        }
    }
    DPC_UNROLL for (int q = 0; q < 3; q++)
        if (x.l3i(q) >= 0 && N.ndx >= x.l3n(q) - 4 && N.ndx < x.l3n(q) + DPW_MAX_OPP_OVLP - 5) dpc_cand_push(U, x, q, N.i, N.ndx, B.sv, B.tbn);
}
DPW_HD void dpc_finish_r5(DpcRegs& R, const DpcNode& N, const double negc, DpcOut& B) {
    const double NI = -__builtin_huge_val();
    const bool reached = B.tb != -1;
    B.sv = reached ? B.val : NI; B.tbn = -1;
    dpc_note_end(R, B, N.i);
    dpc_max_take_asc(R.r5_all, reached, B.val + negc, N.i);
}
// a reverse stop becomes the last one of its frame; the frame's candidate list starts over with the forward stops up to four bases
// before it (oldest first: the list is in position order like the chain); they are near, i.e. not folded yet
template <class X>
DPW_HD void dpc_finish_r3(DpcUni& U, const DpcNode& N, X& x, DpcOut& B) {
    B.sv = B.val; B.tbn = -1;
    const int f = N.frame;
    x.set_l3v(f, B.val);
    x.set_l3(f, N.i, N.stop_val, N.ndx);
    U.cn &= ~(15 << (4 * f));
    int a0 = U.fp > N.i - 16 ? U.fp : N.i - 16;             // at most five positions, two nodes each: ten nodes
    if (a0 < x.reach()) { a0 = x.reach(); if (x.ndx_of(a0) >= N.ndx - 4) U.cn |= 8 << (4 * f); }      // (a reach this short only exists in the tests)
    x.for_near(a0, N.i, DPC_K_F3, [&](const int j, const int nj) {
        if (nj >= N.ndx - 4) { const DpcHist h = x.hist(j); dpc_cand_push(U, x, f, j, nj, h.sv, h.tbn); }
    });
}
