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
// code per frame).  Nothing uniform is carried along: what a node has to do comes compiled in its record (DpcProg, below).
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
#define DPC_CAND 6          // forward stops kept per reverse frame (at most 6: the slot masks of DpcProg)
#endif
#ifndef DPC_HIST
#define DPC_HIST 32         // nodes of history a lane keeps (a power of two)
#endif

// a node's topology (wave-uniform; the slow routine's view of it)
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

// What the routines need from their surroundings (the kernel: history, lists and the per-frame records in LDS, everything else in
// HBM; the host model: plain arrays):
//   DpcHist hist(j)             a finished gene end j of the last DPC_HIST nodes (j uniform)
//   int ndx_of(j)               position of node j (uniform)
//   DpcCarry carry(f) / set_carry(f, c);  double l3v(f) / set_l3v(f, v)         the per-frame records (f uniform), per lane
//   cand_put(f, k, idx, ndx, sv, tbn) / cand_idx(f, k) / cand_ndx(f, k) (uniform) / cand_val(f, k) -> DpcHist (per lane)
//   double igm(d)               the intergenic term at distance 0 <= d <= OPER_DIST (d uniform)
//   DpwS src(j)                 ANY finished node j < i as a source, read back from memory (slow path; j uniform):
//                               kind, frame, ndx, stop_val, score, tbn (-1: never reached), cs, vm, x0..x2
//   bool any(p)                 p holds in some lane of the wave

template <class X>
DPW_HD void dpc_init(DpcRegs& R, X& x) {
    const double NI = -__builtin_huge_val();
    R.r5_all = DpcMax{NI, -1}; R.r5_far = DpcMax{NI, -1}; R.f3_far = DpcMax{NI, -1};
    R.end_best = -1.0; R.end_idx = -1; R.end_tb = -1;
    DPC_UNROLL for (int f = 0; f < 3; f++) { x.set_carry(f, DpcCarry{NI, -1, -1}); x.set_l3v(f, 0.0); }
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

// ------------------------------------------------------------------------------------------------------------------------
// The topology of a contig, compiled: one 64-byte record per node that says what the node's fast routine has to do -- which of
// the 32 nodes before it to fold or to pair with (bit k of a mask = node i - 1 - k), which slots of which candidate list apply,
// where the last reverse stop of each frame is -- so that the serial walk makes no topological decision of its own: it reads
// the record (scalar loads, one node ahead) and does the per-lane arithmetic.  dpc_compile_node is run once per node and
// translation-table group by a parallel pass (k_dpc_compile; the models of a contig share the records).
struct DpcProg { int32_t w[16]; };
//  w0   kf (kind | frame << 2 | in-ORF bits << 4) | flags << 8 | (reverse stop: mask of the forward stops to re-enter, << 16)
//  w1   ndx        w2   stop_val        w3   lo
//  forward start:   w4 fold R5   w5 fold F3   w6 near F3   w7 near F3 within OPER_DIST bases (the distance term is the table's)
//  reverse stop:    w4 fold R5   w5 fold F3   w6 near F3   w7 near R5   w8 near R5 within OPER_DIST bases
//                   w9-11 index of the last reverse stop of frame 0..2 (-1: none)   w12-14 its position
//                   w15 operon bits (3) | list incomplete bits (3) << 3 | list slot masks (3 x 6) << 6
//  reverse start:   w4 index of its own stop (the last reverse stop of its frame)   w5 list slot mask (6)
//  forward stop:    w4 enter-the-list bits (3) | slot in the list of frame q << (3 + 3 q)
#define DPC_F_SLOW    (1 << 8)      // the fast routine does not apply (see dpc_compile_node): dpc_cand_slow
#define DPC_F_OWN     (1 << 9)      // reverse start: its own stop is a candidate
#define DPC_F_WINDOW  (1 << 10)     // gene begin: lo > 0, the running maxima have to be checked against the window
#ifndef DPC_REACH
#define DPC_REACH 32                // a mask reaches this many nodes back: the history's depth
#endif

DPW_HD int dpc_prog_kind(const DpcProg& P) { return P.w[0] & 3; }
DPW_HD int dpc_prog_frame(const DpcProg& P) { return (P.w[0] >> 2) & 3; }

// The forward stops of the candidate list of the reverse stop `l3` (position l3n) as node i sees it: those at positions
// [l3n - 4, l3n + MAX_OPP_OVLP - 5) with an index below i, in chain order; f(slot, c) for the first DPC_CAND of them; returns how
// many there are (more than DPC_CAND: the list is incomplete).
template <class F>
DPW_HD int dpc_list_walk(const int32_t* ndx, const uint8_t* kf, const int l3, const int i, F f) {
    const int l3n = ndx[l3];
    int c = l3;
    while (c > 0 && ndx[c - 1] >= l3n - 4) c--;
    int s = 0;
    for (; c < i && ndx[c] < l3n + DPW_MAX_OPP_OVLP - 5; c++) {
        if (DPW_KIND(kf[c]) != 1) continue;
        if (s < DPC_CAND) f(s, c);
        s++;
    }
    return s;
}

// kf / lo / q1 / q2: DpwTopo of every node of the contig (dpw_topo_node); l3i[q]: the last reverse stop of frame q before node i
DPW_HD void dpc_compile_node(const int32_t* ndx, const int32_t* stopv, const uint8_t* kf, const int32_t* lo, const int32_t* q1, const int32_t* q2,
                             const int i, const int l3i0, const int l3i1, const int l3i2, DpcProg& P) {
    for (int k = 0; k < 16; k++) P.w[k] = 0;
    const int kind = DPW_KIND(kf[i]), f = DPW_FRAME(kf[i]);
    const int my = ndx[i];
    int w0 = kf[i];
    P.w[1] = my; P.w[2] = stopv[i]; P.w[3] = lo[i];
    if (kind == 0 || kind == 3) {
        // what is folded here: the gene ends between the near zone of the gene begin before this one and this one's
        int pg = i - 1;
        while (pg >= 0 && pg >= i - DPC_REACH - 1 && (DPW_KIND(kf[pg]) == 1 || DPW_KIND(kf[pg]) == 2)) pg--;
        const bool found = pg >= 0 && pg >= i - DPC_REACH - 1;
        const int fpb = found ? q1[pg] : (pg < 0 ? 0 : -1);
        if (fpb < 0 || fpb < i - DPC_REACH) w0 |= DPC_F_SLOW;
        else {
            for (int j = fpb; j < q1[i]; j++) {
                const int k = DPW_KIND(kf[j]);
                if (k == 2) P.w[4] |= 1 << (i - 1 - j); else if (k == 1) P.w[5] |= 1 << (i - 1 - j);
            }
            for (int j = q1[i] > fpb ? q1[i] : fpb; j < i; j++) {
                const int k = DPW_KIND(kf[j]), d = my - ndx[j];
                if (k == 1 && d > (kind == 0 ? 2 : 4)) { P.w[6] |= 1 << (i - 1 - j); if (kind == 0 && d <= DPW_OPER_DIST) P.w[7] |= 1 << (i - 1 - j); }
                if (kind == 3 && k == 2 && d > 2) { P.w[7] |= 1 << (i - 1 - j); if (d <= DPW_OPER_DIST) P.w[8] |= 1 << (i - 1 - j); }
            }
        }
        if (lo[i] > 0) w0 |= DPC_F_WINDOW;
    }
    if (kind == 3) {
        const int l3[3] = {l3i0, l3i1, l3i2};
        int w15 = 0;
        for (int q = 0; q < 3; q++) {
            P.w[9 + q] = l3[q]; P.w[12 + q] = l3[q] >= 0 ? ndx[l3[q]] : INT_MIN;
            if (l3[q] < 0) continue;
            if (l3[q] >= lo[i] && stopv[l3[q]] > my) w15 |= 1 << q;
            int m = 0;
            const int cnt = dpc_list_walk(ndx, kf, l3[q], i, [&](const int s, const int c) { if (c >= lo[i] && ndx[c] + 2 < my - 2) m |= 1 << s; });
            if (cnt > DPC_CAND) w15 |= 8 << q;
            w15 |= m << (6 + 6 * q);
        }
        P.w[15] = w15;
        // the forward stops up to four bases before this stop start its frame's list over (at most five positions, two nodes each)
        int rp = 0;
        for (int j = i - 1; j >= 0 && j >= i - 16 && ndx[j] >= my - 4; j--) if (DPW_KIND(kf[j]) == 1) rp |= 1 << (i - 1 - j);
        w0 |= rp << 16;
    } else if (kind == 2) {
        const int l3 = f == 0 ? l3i0 : (f == 1 ? l3i1 : l3i2);
        P.w[4] = l3;
        bool usable = false;
        if (l3 >= 0) {
            if (l3 >= lo[i] && stopv[l3] > my) w0 |= DPC_F_OWN;
            if (ndx[l3] == stopv[i]) {
                int m = 0;
                const int cnt = dpc_list_walk(ndx, kf, l3, i, [&](const int s, const int c) {
                    const int rel = ndx[c] - stopv[i];
                    if (c >= lo[i] && rel > -4 && rel + 5 < DPW_MAX_OPP_OVLP && rel < my - ndx[c] + 3) m |= 1 << s;
                });
                usable = cnt <= DPC_CAND;
                P.w[5] = m;
            }
        }
        if (!usable && q2[i] < i) w0 |= DPC_F_SLOW;
    } else if (kind == 1) {
        const int l3[3] = {l3i0, l3i1, l3i2};
        int w4 = 0;
        for (int q = 0; q < 3; q++) {
            if (l3[q] < 0) continue;
            const int l3n = ndx[l3[q]];
            if (!(my >= l3n - 4 && my < l3n + DPW_MAX_OPP_OVLP - 5)) continue;
            const int slot = dpc_list_walk(ndx, kf, l3[q], i, [](int, int) {});
            if (slot < DPC_CAND) w4 |= (1 << q) | (slot << (3 + 3 * q));
        }
        P.w[4] = w4;
    }
    P.w[0] = w0;
}

// ------------------------------------------------------------------------------------------------------------------------
// What the record cannot know: the window start has passed the argmax of a running maximum the node reads (it only can from node
// 1000 on), or a lane's overlapping start belongs to a gene whose stop is not the last reverse stop of its frame (or whose list is
// incomplete) while its static chain of candidates is not empty.  `any` lane is enough to send all of them through the slow routine.
template <class X>
DPW_HD bool dpc_need_slow_begin(const DpcRegs& R, const DpcProg& P, X& x) {
    if (P.w[0] & DPC_F_SLOW) return true;
    if (!(P.w[0] & DPC_F_WINDOW)) return false;
    const int lo = P.w[3];
    const DpcMax& rmax = dpc_prog_kind(P) == 3 ? R.r5_far : R.r5_all;
    return x.any(((rmax.i >= 0) & (rmax.i < lo)) | ((R.f3_far.i >= 0) & (R.f3_far.i < lo)));
}
template <class X>
DPW_HD bool dpc_need_slow_r3(const DpcProg& P, const int i, const DpcExt& e, X& x) {
    bool slow = false;
    DPC_UNROLL for (int q = 0; q < 3; q++) {
        const bool list = (P.w[9 + q] >= 0) & (P.w[12 + q] == e.n3s[q]) & !((P.w[15] >> (3 + q)) & 1);
        slow = slow | (((e.vm >> q) & 1) & !list & (e.cq[q] < i));
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
// (split at q1: what the next gene begin folds starts there; what lies before the window is dropped: windows of gene begins only
// move forward).
template <class X>
DPW_HD void dpc_cand_slow(DpcRegs& R, const DpcNode& N, const double cs, const DpcExt& e, const DpwModel& M, X& x, DpcOut& B) {
    const double NI = -__builtin_huge_val();
    const DpwT T = dpc_target(N, cs, M.negc, e);
    const bool begin = N.kind == 0 || N.kind == 3;
    if (begin) { R.r5_all = DpcMax{NI, -1}; R.r5_far = DpcMax{NI, -1}; R.f3_far = DpcMax{NI, -1}; }
    DpwBest W{0.0, -1, -1, -1};
    for (int j = N.lo; j < N.i; j++) {
        const DpwS s = x.src(j);
        bool ok; double w; int mf;
        dpw_pair(s, T, M, ok, w, mf);
        dpw_take(W, ok, s.score + w, j, mf, s.ndx);
        if (begin && (s.kind == 1 || s.kind == 2)) {
            const bool reached = s.tbn != -1;
            const double a = s.score + M.negc;
            if (s.kind == 2) { dpc_max_take_asc(R.r5_all, reached, a, j); dpc_max_take_asc(R.r5_far, reached & (j < N.q1), a, j); }
            else dpc_max_take_asc(R.f3_far, reached & (j < N.q1), a, j);
        }
    }
    B.val = W.val; B.tb = W.tb; B.ov = W.ov;
}

// ------------------------------------------------------------------------------------------------------------------------
// Fast candidates, driven by the node's record.  Bit k of a mask is node i - 1 - k: walking the set bits from the top is chain
// order.
template <class F>
DPW_HD void dpc_bits_desc(unsigned m, F f) {
    while (m) {
        const int k = 31 - __builtin_clz(m);
        m &= ~(1u << k);
        f(k);
    }
}
// the gene ends that leave the near zone join the far maxima (each kind in chain order)
template <class X>
DPW_HD void dpc_fold(DpcRegs& R, const DpcProg& P, const int i, const double negc, X& x) {
    const double NI = -__builtin_huge_val();
    dpc_bits_desc((unsigned)P.w[4], [&](const int k) { const int j = i - 1 - k; const DpcHist h = x.hist(j); dpc_max_take_asc(R.r5_far, h.sv > NI, h.sv + negc, j); });
    dpc_bits_desc((unsigned)P.w[5], [&](const int k) { const int j = i - 1 - k; const DpcHist h = x.hist(j); dpc_max_take_asc(R.f3_far, h.sv > NI, h.sv + negc, j); });
}

// a forward start: every gene end of the window (ref: _connection.h:117-130)
template <class X>
DPW_HD void dpc_cand_f5(DpcRegs& R, const DpcProg& P, const int i, const DpwModel& M, X& x, DpcOut& B) {
    dpc_fold(R, P, i, M.negc, x);
    // forward stops, in chain order: the far ones, then those within 3 * OPER_DIST bases pair by pair
    dpc_take_asc(B, true, R.f3_far.v, R.f3_far.i);
    const unsigned tab = (unsigned)P.w[7];
    dpc_bits_desc((unsigned)P.w[6], [&](const int k) {
        const int j = i - 1 - k;
        const DpcHist h = x.hist(j);
        const double w = ((tab >> k) & 1) ? x.igm(P.w[1] - x.ndx_of(j)) : 0.0;
        dpc_take_asc(B, true, h.sv + w, j);
    });
    // every reverse start so far: the weight towards a forward start never depends on the distance
    dpc_take_lex(B, true, R.r5_all.v, R.r5_all.i);
}

// a forward stop: the best start / operon partner of its ORF (ref: :166-188) -- the frame's carry
DPW_HD void dpc_cand_f3(const DpcCarry& c, DpcOut& B) {
    const bool reached = (c.i >= 0) & (c.v >= 0.0);
    B.val = reached ? c.v : 0.0; B.tb = reached ? c.i : -1; B.tbn = reached ? c.n : -1;
}

// a reverse start of frame f: its own stop (ref: :228-235) and the forward stops overlapping its gene's 3' end (ref: :238-254; the
// tests that only look at positions are in the record's slot mask): rel = c.ndx - stop_val, tbn < stop_val - 3 - rel
template <class X>
DPW_HD void dpc_cand_r5(const DpcProg& P, const double cs, const double negc, X& x, DpcOut& B) {
    const int f = dpc_prog_frame(P);
    if (P.w[0] & DPC_F_OWN) dpc_take_asc(B, true, x.l3v(f) + cs, P.w[4]);
    if (P.w[5]) {
        // (the list follows the frame's last reverse stop: in chain order, after the stop itself)
        const double csd = cs + negc;
        unsigned m = (unsigned)P.w[5];
        while (m) {
            const int k = __builtin_ctz(m);
            m &= m - 1;
            const DpcHist v = x.cand_val(f, k);
            dpc_take_asc(B, v.tbn < 2 * P.w[2] - 3 - x.cand_ndx(f, k), v.sv + csd, x.cand_idx(f, k));
        }
    }
}

// a reverse stop: every gene end of the window (ref: :288-342), the reverse stop whose ORF covers it (ref: :345-356)
template <class X>
DPW_HD void dpc_cand_r3(DpcRegs& R, const DpcProg& P, const int i, const DpcExt& e, const DpwModel& M, X& x, DpcOut& B) {
    dpc_fold(R, P, i, M.negc, x);
    // the far gene ends (the weight is the constant): both kinds, either order
    dpc_take_asc(B, true, R.r5_far.v, R.r5_far.i);
    dpc_take_lex(B, true, R.f3_far.v, R.f3_far.i);
    // near gene ends, pair by pair: reverse starts with the distance term, forward stops the plain connection (a forward stop's
    // offer through an overlapping start of this stop is met on the candidate lists, and is larger)
    const unsigned tab = (unsigned)P.w[8];
    dpc_bits_desc((unsigned)P.w[7], [&](const int k) {
        const int j = i - 1 - k;
        const DpcHist h = x.hist(j);
        const double w = ((tab >> k) & 1) ? x.igm(P.w[1] - x.ndx_of(j)) : 0.0;
        dpc_take_lex(B, true, h.sv + w, j);
    });
    dpc_bits_desc((unsigned)P.w[6], [&](const int k) { const int j = i - 1 - k; const DpcHist h = x.hist(j); dpc_take_lex(B, true, h.sv + M.negc, j); });
    // the reverse stop whose ORF covers this one, per frame of an overlapping start: an operon (ref: :345-356)
    DPC_UNROLL for (int q = 0; q < 3; q++)
        if ((P.w[15] >> q) & 1) dpc_take_lex(B, (e.vm & (1 << q)) != 0, x.l3v(q) + e.x[q], P.w[9 + q]);
    // forward stops that overlap the 3' end of the gene of an overlapping start (ref: :288-336).  The start of frame q has its stop
    // at n3s[q]; where that is the last reverse stop of frame q, the forward stops that can overlap it are that frame's list.  A
    // forward stop c goes through start q when  ovlp = c.ndx + 5 - n3s[q]  is in (0, MAX_OPP_OVLP) -- what the list holds --,
    // ovlp < n3n[q] - (c.ndx + 2),  ovlp < n3s[q] - c.tbn - 2  and x[q] > 0; of the starts it can go through the reference keeps the
    // first with the largest x, and of the forward stops the last with the largest value: the lexicographic maximum over the
    // (stop, start) pairs, walked start by start, a later start only on a strictly larger value.
    DPC_UNROLL for (int q = 0; q < 3; q++) {
        unsigned m = ((unsigned)P.w[15] >> (6 + 6 * q)) & 63u;
        if (m) {
            const int l3n = P.w[12 + q];
            const bool have = ((e.vm & (1 << q)) != 0) & (l3n == e.n3s[q]) & (e.x[q] > 0.0);
            if (x.any(have)) {
                while (m) {
                    const int k = __builtin_ctz(m);
                    m &= m - 1;
                    const DpcHist c = x.cand_val(q, k);
                    const int c_ndx = x.cand_ndx(q, k), c_idx = x.cand_idx(q, k);
                    const int ovlp = c_ndx + 5 - l3n;
                    const double v = c.sv + e.x[q];
                    const bool t = have & (e.n3n[q] > ovlp + c_ndx + 2) & (c.tbn < l3n - 2 - ovlp) & ((v > B.val) | ((v == B.val) & (c_idx > B.tb)));
                    B.val = t ? v : B.val; B.tb = t ? c_idx : B.tb; B.ov = t ? q : B.ov;
                }
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
// a forward start offers score + cs to the stop of its ORF (a later node wins a tie); `c`: the carry of its frame
template <class X>
DPW_HD void dpc_finish_f5(const DpcProg& P, const int i, const double cs, const DpcCarry& c, X& x, DpcOut& B) {
    const double g = B.val + cs;
    const bool t = g >= c.v;
    x.set_carry(dpc_prog_frame(P), DpcCarry{t ? g : c.v, t ? i : c.i, t ? P.w[1] : c.n});
    B.sv = B.val; B.tbn = -1;
}
// a forward stop restarts the running maximum of its own frame and, when reached, offers score + x to the frames whose next stop's
// ORF holds it (operon partners); it enters the candidate lists of the reverse stops whose genes' 3' ends it can overlap
template <class X>
DPW_HD void dpc_finish_f3(DpcRegs& R, const DpcProg& P, const int i, const DpcExt& e, X& x, DpcOut& B) {
    const double NI = -__builtin_huge_val();
    const bool reached = B.tb != -1;
    B.sv = reached ? B.val : NI;
    dpc_note_end(R, B, i);
    x.set_carry(dpc_prog_frame(P), DpcCarry{NI, -1, -1});
    DPC_UNROLL for (int q = 0; q < 3; q++) {
        if (DPW_INORF(P.w[0], q)) {
            const DpcCarry c = x.carry(q);
            const double o = B.val + e.x[q];
            const bool t = reached & ((e.vm & (1 << q)) != 0) & (o >= c.v);
            x.set_carry(q, DpcCarry{t ? o : c.v, t ? i : c.i, t ? P.w[1] : c.n});
        }
    }
    DPC_UNROLL for (int q = 0; q < 3; q++)
        if ((P.w[4] >> q) & 1) x.cand_put(q, (P.w[4] >> (3 + 3 * q)) & 7, i, P.w[1], B.sv, B.tbn);
}
DPW_HD void dpc_finish_r5(DpcRegs& R, const int i, const double negc, DpcOut& B) {
    const double NI = -__builtin_huge_val();
    const bool reached = B.tb != -1;
    B.sv = reached ? B.val : NI; B.tbn = -1;
    dpc_note_end(R, B, i);
    dpc_max_take_asc(R.r5_all, reached, B.val + negc, i);
}
// a reverse stop becomes the last one of its frame; the frame's candidate list starts over with the forward stops up to four bases
// before it (oldest first: the list is in position order like the chain)
template <class X>
DPW_HD void dpc_finish_r3(const DpcProg& P, const int i, X& x, DpcOut& B) {
    B.sv = B.val; B.tbn = -1;
    const int f = dpc_prog_frame(P);
    x.set_l3v(f, B.val);
    int slot = 0;
    dpc_bits_desc((unsigned)P.w[0] >> 16, [&](const int k) {
        const int j = i - 1 - k;
        if (slot < DPC_CAND) { const DpcHist h = x.hist(j); x.cand_put(f, slot, j, x.ndx_of(j), h.sv, h.tbn); }
        slot++;
    });
}
