// Lane-per-chain connection scoring (dp_lane.hip): ONE lane walks ONE (contig, model) chain node by node, 64 chains to a
// wavefront, and every class of candidates a node has comes from a running structure that costs O(1) per node -- no lane of the
// wave ever visits a source on behalf of another lane.  Written once for the device and for the host: tests/dpl_model.cpp runs
// the same step function chain by chain on the CPU against the plain restatement of the reference's loop
// (tests/test_dpl_model.py), so the case analysis is pinned before a kernel ever runs.  No intrinsics in this file.
//
// Same recurrence as dpw_core.h (ref: lib.pyx:1205-1237 `_score_connections`, _connection.h:94-408, impl/generic.h:29-36):
//     score[i] = max(0, max_j (score[j] + w(j, i))) over the window [lo_i, i), ties -> largest j.
// Kinds: 0 = F5 forward start, 1 = F3 forward stop, 2 = R5 reverse start, 3 = R3 reverse stop; F5 / R3 are gene BEGINS, F3 / R5
// gene ENDS.  Which (source kind, target kind) pairs connect at all is what impl/generic.h:29-36 leaves:
//     F5 <- R5 (w = -0.15 st_wt), F5 <- F3 (distance term), R3 <- R5 (distance term), R3 <- F3 (-0.15 st_wt, or through an
//     overlapping start of the target), R3 <- R3 (operon), F3 <- F5 (same ORF), F3 <- F3 (operon), R5 <- R3 (own stop),
//     R5 <- F3 (opposite 3' ends overlapping).
// Where a lane finds them:
//   * gene begins <- gene ends far away (more than 3 * OPER_DIST bases: the weight is the constant): lexicographic running
//     maxima of a = score + (-0.15 st_wt), one per (source kind, lag):  every reverse start so far (F5 targets: the weight
//     towards a forward start never depends on the distance), the reverse starts / forward stops that lie more than 180 bases
//     behind the walk ("folded");
//   * gene begins <- gene ends within 180 bases: the ends that are not folded yet sit in two small per-lane rings (forward
//     stops, reverse starts) and are evaluated pair by pair with dpw_pair -- one or two forward stops, a handful of reverse starts;
//   * forward stops <- the starts / operon partners of their ORF: a per-frame running maximum that restarts at every forward
//     stop of the frame (as in dpw_core.h);
//   * reverse starts <- their own stop, reverse stops <- the reverse stop whose ORF covers them: "last reverse stop per frame";
//   * reverse nodes <- forward stops overlapping the 3' end of the gene: the static chains of candidates (q2 links), read back
//     from memory (a gene length behind the walk).
// The window [lo, i) only matters for the running maxima (every other class lies inside the window by construction, and is
// checked anyway): a maximum whose argmax fell out of the window is rebuilt by a scan of the window (dpl_rescan) -- exact, slow,
// and rare: scores grow along a chain, so the maxima are young.  A ring that overflows (more gene ends within 180 bases than it
// holds) folds its oldest entry early and sends the gene begins of the next 180 bases through the same scan.
#pragma once

#include "dpw_core.h"

#ifndef DPL_R5_RING
#define DPL_R5_RING 16      // entries, a power of two (the tests also build the model with rings of 2 and 1 entries)
#endif
#ifndef DPL_F3_RING
#define DPL_F3_RING 8
#endif
#ifndef DPL_CAND
#define DPL_CAND 4          // forward stops kept per reverse frame (those within MAX_OPP_OVLP bases of the frame's last reverse stop)
#endif
#define DPL_NEAR    (3 * DPW_OPER_DIST)

struct DplEnt { double score; int32_t ndx, idx; };
// a finished node as memory holds it: position, next-candidate link, topology byte; position of its traceb node, traceb (-1: not
// reached), score
struct DplFin { int32_t ndx, q2, kf, tbn, tb; double score; };
// a reached forward stop that can overlap the 3' end of a reverse gene: score, position, chain index, position of its traceb node
struct DplCand { double score; int32_t ndx, idx, tbn; };

// lexicographic (value, index) maximum with the position of the argmax node carried along
struct DplMax { double v; int i, n; };
DPW_HD void dpl_max_take(DplMax& m, const double v, const int i, const int n) {
    if (v > m.v || (v == m.v && i > m.i)) { m.v = v; m.i = i; m.n = n; }
}

struct DplState {
    DplMax r5_all;              // a over every reached reverse start so far
    DplMax r5_far, f3_far;      // a over the reverse starts / forward stops that left their ring (more than 180 bases behind)
    int r5_head, r5_cnt, f3_head, f3_cnt;      // rings: the last `cnt` entries pushed, newest at head - 1 (slots modulo the ring size)
    int r5_old, f3_old;         // position of the oldest entry of each ring (INT_MAX: empty): the fold test without a read
    int r5_ovf, f3_ovf;         // an entry left its ring early: gene begins at positions <= this go through the scan
    int cn;                     // per reverse frame f: bits 4f .. 4f+2 = forward stops in its candidate list, bit 4f+3 = the list is incomplete
    // forward frames: best start / operon offer since the last forward stop of the frame
    double rv0, rv1, rv2; int ri0, ri1, ri2, rn0, rn1, rn2;
    // last reverse stop of each frame: its score, index, stop_val, position
    double l3v0, l3v1, l3v2; int l3i0, l3i1, l3i2, l3s0, l3s1, l3s2, l3n0, l3n1, l3n2;
    double end_best; int end_idx, end_tb;
};

DPW_HD void dpl_init(DplState& S) {
    const double NI = -__builtin_huge_val();
    S.r5_all = DplMax{NI, -1, -1}; S.r5_far = DplMax{NI, -1, -1}; S.f3_far = DplMax{NI, -1, -1};
    S.r5_head = S.r5_cnt = S.f3_head = S.f3_cnt = 0;
    S.r5_ovf = S.f3_ovf = INT_MIN;
    S.r5_old = S.f3_old = INT_MAX;
    S.cn = 0;
    S.rv0 = S.rv1 = S.rv2 = NI; S.ri0 = S.ri1 = S.ri2 = -1; S.rn0 = S.rn1 = S.rn2 = -1;
    S.l3v0 = S.l3v1 = S.l3v2 = 0.0; S.l3i0 = S.l3i1 = S.l3i2 = -1; S.l3s0 = S.l3s1 = S.l3s2 = 0; S.l3n0 = S.l3n1 = S.l3n2 = 0;
    S.end_best = -1.0; S.end_idx = -1; S.end_tb = -1;
}

// What a step needs from its surroundings (the kernel: LDS rings + the chain's arrays in HBM; the host model: plain arrays):
//   DplEnt r5_get(slot) / r5_put(slot, e) / f3_get / f3_put      the lane's rings
//   int    f3t_get(slot) / f3t_put(slot, tbn)                   position of the traceb node of the forward stops in the ring
//   DplCand cand_get(f, k) / cand_put(f, k, c)                  the candidate list of reverse frame f
//   double igm(d)       the intergenic term at distance 0 <= d <= OPER_DIST: (2 - d / 60) * 0.15 * st_wt (ModelConst::igm)
//   DplFin fin(j)       a FINISHED node j < i of this chain, read back from memory (topology + results in one go)
//   void   note(k)      diagnostics of the host model (0: a window scan, 1: near gene ends read back after a ring overflow,
//                       2: a chain of overlap candidates walked in memory)
//
// dpl_rescan: every gene end of the window against this gene begin, pair by pair (what the reference's loop does for these
// sources), and the running maxima rebuilt over the window on the way.  `r5_first` / `f3_first`: chain index of the oldest entry
// still in the ring (or T.i when the ring is empty): ends before it are the folded ones.
template <class X>
DPW_HD void dpl_rescan(DplState& S, const DpwT& T, const DpwModel& M, X& x, DpwBest& B, const int r5_first, const int f3_first) {
    const double NI = -__builtin_huge_val();
    DplMax all{NI, -1, -1}, r5f{NI, -1, -1}, f3f{NI, -1, -1};
    for (int j = T.lo; j < T.i; j++) {
        const DplFin r = x.fin(j);
        const int k = DPW_KIND(r.kf);
        if ((k != 1 && k != 2) || r.tb == -1) continue;
        DpwS s;
        s.j = j; s.kind = k; s.frame = DPW_FRAME(r.kf); s.ndx = r.ndx; s.stop_val = 0; s.vm = 0; s.tbn = r.tbn; s.score = r.score;
        s.cs = 0.0; s.x0 = s.x1 = s.x2 = 0.0;
        bool ok; double w; int mf;
        dpw_pair(s, T, M, ok, w, mf);
        dpw_take(B, ok, s.score + w, j, mf, s.ndx);
        const double a = s.score + M.negc;
        if (k == 2) { dpl_max_take(all, a, j, s.ndx); if (j < r5_first) dpl_max_take(r5f, a, j, s.ndx); }
        else if (j < f3_first) dpl_max_take(f3f, a, j, s.ndx);
    }
    S.r5_all = all; S.r5_far = r5f; S.f3_far = f3f;
}

DPW_HD double dpl_sel3(const int k, const double a, const double b, const double c) { return k == 0 ? a : (k == 1 ? b : c); }
DPW_HD int dpl_sel3i(const int k, const int a, const int b, const int c) { return k == 0 ? a : (k == 1 ? b : c); }

// Lexicographic take without branches (the compiler turns it into compares and selects).
DPW_HD void dpl_take(DpwBest& b, const bool ok, const double v, const int j, const int ov, const int n) {
    const bool better = ok & ((v > b.val) | ((v == b.val) & (j > b.tb)));
    b.val = better ? v : b.val; b.tb = better ? j : b.tb; b.ov = better ? ov : b.ov; b.tbn = better ? n : b.tbn;
}
// the intergenic term of two same-strand nodes d bases apart, 0 <= d <= 3 * OPER_DIST (ref: _connection.h:52-78 with the nodes
// neither overlapping nor touching): the table value up to OPER_DIST, nothing beyond
template <class X>
DPW_HD double dpl_igm_near(X& x, const int d) { return d <= DPW_OPER_DIST ? x.igm(d) : 0.0; }

// A reached gene end within 180 bases of a gene begin, as one pair (what dpw_pair gives for these kinds, spelled out):
//   forward stop -> forward start (ref: :117-124)   ok: s + 2 < t       w: the distance term
//   reverse start -> reverse stop (ref: :337-342)   ok: s < t - 2       w: the distance term
//   forward stop -> reverse stop  (ref: :288-336)   ok: s + 2 < t - 2   w: -0.15 st_wt -- the plain connection only: the candidates
//       that go through an overlapping start of the target need the position of the source's own traceb node and are met on
//       the lists of candidates, with a larger value
// `same`: source and target on the same strand (the first two); e.ndx >= T.ndx - 180 for every entry of a ring.
template <class X>
DPW_HD void dpl_near(DpwBest& B, const DpwT& T, const DpwModel& M, X& x, const bool same, const DplEnt e) {
    const int d = T.ndx - e.ndx;
    const bool ok = (e.idx >= T.lo) & (d > (same ? 2 : 4));
    const double w = same ? (d > DPL_NEAR ? M.negc : dpl_igm_near(x, d < 0 ? 0 : d)) : M.negc;
    dpl_take(B, ok, e.score + w, e.idx, -1, e.ndx);
}

// a forward stop met on a chain of candidates (ref: _connection.h:238-254, 296-325), as one pair
DPW_HD void dpl_f3_candidate(DpwBest& B, const DpwT& T, const DpwModel& M, const int j, const DplFin& r) {
    DpwS s;
    s.j = j; s.kind = 1; s.frame = 0; s.ndx = r.ndx; s.stop_val = 0; s.vm = 0; s.tbn = r.tb == -1 ? -1 : r.tbn; s.score = r.score;
    s.cs = 0.0; s.x0 = s.x1 = s.x2 = 0.0;
    bool ok; double w; int mf;
    dpw_pair(s, T, M, ok, w, mf);
    dpw_take(B, ok, s.score + w, j, mf, r.ndx);
}

// A reached forward stop of a candidate list against a reverse start (ref: _connection.h:238-254; what dpw_pair gives, spelled out)
DPW_HD void dpl_cand_r5(DpwBest& B, const DpwT& T, const DplCand& c) {
    const int rel = c.ndx - T.stop_val;
    const bool ok = (c.idx >= T.lo) & (rel > -4) & (rel + 5 < DPW_MAX_OPP_OVLP) & (rel < T.ndx - c.ndx + 3) & (rel < T.stop_val - 3 - c.tbn);
    dpl_take(B, ok, c.score + T.csd, c.idx, -1, c.ndx);
}
// ... and against a reverse stop, through the best admissible overlapping start (ref: :288-336): the first q with the largest x
DPW_HD void dpl_cand_r3(DpwBest& B, const DpwT& T, const DpwModel& M, const DplCand& c) {
    const int left = c.ndx + 2;
    const bool ok = (c.idx >= T.lo) & (left < T.ndx - 2);
    double maxval = 0.0; int mf = -1;
    {
        const int ovlp = left - T.n3s0 + 3;
        const bool tk = ((T.vm & 1) != 0) & (ovlp > 0) & (ovlp < DPW_MAX_OPP_OVLP) & (ovlp < T.n3n0 - left) & (ovlp < T.n3s0 - c.tbn - 2) & (T.x0 > maxval);
        maxval = tk ? T.x0 : maxval; mf = tk ? 0 : mf;
    }
    {
        const int ovlp = left - T.n3s1 + 3;
        const bool tk = ((T.vm & 2) != 0) & (ovlp > 0) & (ovlp < DPW_MAX_OPP_OVLP) & (ovlp < T.n3n1 - left) & (ovlp < T.n3s1 - c.tbn - 2) & (T.x1 > maxval);
        maxval = tk ? T.x1 : maxval; mf = tk ? 1 : mf;
    }
    {
        const int ovlp = left - T.n3s2 + 3;
        const bool tk = ((T.vm & 4) != 0) & (ovlp > 0) & (ovlp < DPW_MAX_OPP_OVLP) & (ovlp < T.n3n2 - left) & (ovlp < T.n3s2 - c.tbn - 2) & (T.x2 > maxval);
        maxval = tk ? T.x2 : maxval; mf = tk ? 2 : mf;
    }
    dpl_take(B, ok, c.score + (mf != -1 ? maxval : M.negc), c.idx, mf, c.ndx);
}

// The forward stops that can overlap the 3' end of a reverse gene whose stop is at `stop_pos` (frame f) against target T: they lie
// within [stop_pos - 4, stop_pos + MAX_OPP_OVLP - 5) (ref: _connection.h:238-254, 296-325).  The lane keeps those that follow the
// LAST reverse stop of each frame in a small list, filled as the forward stops are finished; a target whose stop is not that
// one, or whose list overflowed, walks the static chain of candidates in memory instead (first candidate `first`, q2 links).
template <class X>
DPW_HD void dpl_overlap_candidates(const DplState& S, const DpwT& T, const DpwModel& M, X& x, DpwBest& B, const int f, const int stop_pos,
                                   const int first) {
    const int l3i = dpl_sel3i(f, S.l3i0, S.l3i1, S.l3i2), l3n = dpl_sel3i(f, S.l3n0, S.l3n1, S.l3n2);
    const int c4 = (S.cn >> (4 * f)) & 15;
    if (l3i >= 0 && l3n == stop_pos && !(c4 & 8)) {
        for (int k = 0; k < (c4 & 7); k++) {
            const DplCand c = x.cand_get(f, k);
            if (T.kind == 2) dpl_cand_r5(B, T, c); else dpl_cand_r3(B, T, M, c);
        }
        return;
    }
    x.note(2);
    const int bound = stop_pos + DPW_MAX_OPP_OVLP - 5;
    for (int j = first; j < T.i;) {
        const DplFin r = x.fin(j);
        if (r.ndx >= bound) break;
        dpl_f3_candidate(B, T, M, j, r);
        j = r.q2;
    }
}
// a reached forward stop joins the list of every reverse frame whose last stop it can overlap
template <class X>
DPW_HD void dpl_cand_push(DplState& S, X& x, const int f, const DplCand& c) {
    const int c4 = (S.cn >> (4 * f)) & 15;
    if (c4 & 8) return;
    if ((c4 & 7) == DPL_CAND) { S.cn |= 8 << (4 * f); return; }
    x.cand_put(f, c4 & 7, c);
    S.cn += 1 << (4 * f);
}

// One node: candidates -> B, then the node's own contribution to the running structures.  `kfb`: the node's topology byte.
// Returns through B the node's score / traceb / ov_mark / position of the traceb node.
// Written for lanes in lock step on different kinds of node: short conditionals are selects, the per-frame state is picked and
// written back with selects, and only the loops over ring / list entries and the rare paths (window scan, ring overflow, a
// candidate chain walked in memory) are branches.
template <class X>
DPW_HD void dpl_step(DplState& S, const DpwT& T, const int kfb, const DpwModel& M, X& x, DpwBest& B) {
    const double NI = -__builtin_huge_val();
    B.val = 0.0; B.tb = -1; B.ov = -1; B.tbn = -1;
    const int i = T.i;
    const int f = T.frame;
    const bool k0 = T.kind == 0, k1 = T.kind == 1, k2 = T.kind == 2, k3 = T.kind == 3;
    // ---- fold the ring entries that are more than 180 bases behind this node (they are for every later node too)
    const int far_pos = T.ndx - DPL_NEAR;            // an end at a position below this is far
    while (S.r5_cnt > 0 && S.r5_old < far_pos) {
        const DplEnt e = x.r5_get((S.r5_head - S.r5_cnt) & (DPL_R5_RING - 1));
        dpl_max_take(S.r5_far, e.score + M.negc, e.idx, e.ndx);
        S.r5_cnt--;
        S.r5_old = S.r5_cnt > 0 ? x.r5_get((S.r5_head - S.r5_cnt) & (DPL_R5_RING - 1)).ndx : INT_MAX;
    }
    while (S.f3_cnt > 0 && S.f3_old < far_pos) {
        const DplEnt e = x.f3_get((S.f3_head - S.f3_cnt) & (DPL_F3_RING - 1));
        dpl_max_take(S.f3_far, e.score + M.negc, e.idx, e.ndx);
        S.f3_cnt--;
        S.f3_old = S.f3_cnt > 0 ? x.f3_get((S.f3_head - S.f3_cnt) & (DPL_F3_RING - 1)).ndx : INT_MAX;
    }
    // the frame's records, picked once: forward running maximum (forward stops read it), last reverse stop (reverse nodes)
    const int l3i = dpl_sel3i(f, S.l3i0, S.l3i1, S.l3i2), l3s = dpl_sel3i(f, S.l3s0, S.l3s1, S.l3s2), l3n = dpl_sel3i(f, S.l3n0, S.l3n1, S.l3n2);
    const double l3v = dpl_sel3(f, S.l3v0, S.l3v1, S.l3v2);
    if (k0 | k3) {
        // ---- a gene begin: every gene end of the window
        // the running maxima this kind of target reads: a forward start every reverse start so far, a reverse stop the far ones
        const DplMax rmax = k3 ? S.r5_far : S.r5_all;
        // is their argmax still inside the window?
        const bool stale = ((rmax.i >= 0) & (rmax.i < T.lo)) | ((S.f3_far.i >= 0) & (S.f3_far.i < T.lo));
        // an entry that left its ring early may still be near: then the near gene ends are read back from memory instead
        const bool ovf = (T.ndx <= S.f3_ovf) | (k3 & (T.ndx <= S.r5_ovf));
        if (stale) {
            x.note(0);
            const int r5_first = S.r5_cnt > 0 ? x.r5_get((S.r5_head - S.r5_cnt) & (DPL_R5_RING - 1)).idx : i;
            const int f3_first = S.f3_cnt > 0 ? x.f3_get((S.f3_head - S.f3_cnt) & (DPL_F3_RING - 1)).idx : i;
            dpl_rescan(S, T, M, x, B, r5_first, f3_first);
        } else {
            // far gene ends: the running maxima (the weight is the constant -0.15 st_wt)
            dpl_take(B, rmax.i >= 0, rmax.v, rmax.i, -1, rmax.n);
            dpl_take(B, S.f3_far.i >= 0, S.f3_far.v, S.f3_far.i, -1, S.f3_far.n);
            // near gene ends, pair by pair: forward stops for both kinds of gene begin, reverse starts for a reverse stop
            if (ovf) {
                x.note(1);
                for (int j = i - 1; j >= T.lo; j--) {
                    const DplFin r = x.fin(j);
                    if (r.ndx < far_pos) break;
                    const int k = DPW_KIND(r.kf);
                    if (!(k == 1 || (k == 2 && k3)) || r.tb == -1) continue;
                    dpl_near(B, T, M, x, (k == 1) != k3, DplEnt{r.score, r.ndx, j});
                }
            } else {
                for (int k = 0; k < S.f3_cnt; k++) dpl_near(B, T, M, x, k0, x.f3_get((S.f3_head - 1 - k) & (DPL_F3_RING - 1)));
                const int nr5 = k3 ? S.r5_cnt : 0;
                for (int k = 0; k < nr5; k++) dpl_near(B, T, M, x, true, x.r5_get((S.r5_head - 1 - k) & (DPL_R5_RING - 1)));
            }
        }
        if (k3) {
            // the reverse stop whose ORF covers this one, per frame of an overlapping start: an operon (ref: :345-356)
            dpl_take(B, ((T.vm & 1) != 0) & (S.l3i0 >= 0) & (S.l3i0 >= T.lo) & (S.l3s0 > T.ndx), S.l3v0 + T.x0, S.l3i0, -1, S.l3n0);
            dpl_take(B, ((T.vm & 2) != 0) & (S.l3i1 >= 0) & (S.l3i1 >= T.lo) & (S.l3s1 > T.ndx), S.l3v1 + T.x1, S.l3i1, -1, S.l3n1);
            dpl_take(B, ((T.vm & 4) != 0) & (S.l3i2 >= 0) & (S.l3i2 >= T.lo) & (S.l3s2 > T.ndx), S.l3v2 + T.x2, S.l3i2, -1, S.l3n2);
            // forward stops that overlap the 3' end of the gene of an overlapping start (the start of frame q has its stop at n3s)
            if (T.vm & 1) dpl_overlap_candidates(S, T, M, x, B, 0, T.n3s0, T.cq0);
            if (T.vm & 2) dpl_overlap_candidates(S, T, M, x, B, 1, T.n3s1, T.cq1);
            if (T.vm & 4) dpl_overlap_candidates(S, T, M, x, B, 2, T.n3s2, T.cq2);
        }
    } else if (k1) {
        // ---- a forward stop: the best start / operon partner of its ORF (ref: :166-188)
        const int ci = dpl_sel3i(f, S.ri0, S.ri1, S.ri2);
        dpl_take(B, ci >= 0, dpl_sel3(f, S.rv0, S.rv1, S.rv2), ci, -1, dpl_sel3i(f, S.rn0, S.rn1, S.rn2));
    } else {
        // ---- a reverse start: its own stop (ref: :228-235) ...
        dpl_take(B, (l3i >= 0) & (l3i >= T.lo) & (l3s > T.ndx), l3v + T.cs, l3i, -1, l3n);
        // ... and the forward stops overlapping its gene's 3' end (ref: :238-254)
        dpl_overlap_candidates(S, T, M, x, B, f, T.stop_val, T.q2);
    }

    // ---- the node is final: what it leaves for later nodes
    const bool reached = B.tb != -1;
    const bool new_end = (k1 | k2) & (B.val >= S.end_best);
    S.end_best = new_end ? B.val : S.end_best; S.end_idx = new_end ? i : S.end_idx; S.end_tb = new_end ? B.tb : S.end_tb;
    // forward frames: a forward start offers score + cs to the stop of its ORF; a forward stop restarts the running maximum of its
    // own frame and, when reached, offers score + x to the frames whose next stop's ORF holds it (operon partners)
    if (k0 | k1) {
        const double g = B.val + T.cs;
        double o0 = NI, o1 = NI, o2 = NI;                   // this node's offer to each frame
        bool z0 = false, z1 = false, z2 = false;            // restart the frame
        if (k0) { o0 = f == 0 ? g : NI; o1 = f == 1 ? g : NI; o2 = f == 2 ? g : NI; }
        else {
            z0 = f == 0; z1 = f == 1; z2 = f == 2;
            o0 = (reached & ((T.vm & 1) != 0) & (DPW_INORF(kfb, 0) != 0)) ? B.val + T.x0 : NI;
            o1 = (reached & ((T.vm & 2) != 0) & (DPW_INORF(kfb, 1) != 0)) ? B.val + T.x1 : NI;
            o2 = (reached & ((T.vm & 4) != 0) & (DPW_INORF(kfb, 2) != 0)) ? B.val + T.x2 : NI;
        }
        {   const double cur = z0 ? NI : S.rv0; const bool t = o0 > NI && o0 >= cur;      // a later node wins a tie
            S.rv0 = t ? o0 : cur; S.ri0 = t ? i : (z0 ? -1 : S.ri0); S.rn0 = t ? T.ndx : (z0 ? -1 : S.rn0); }
        {   const double cur = z1 ? NI : S.rv1; const bool t = o1 > NI && o1 >= cur;
            S.rv1 = t ? o1 : cur; S.ri1 = t ? i : (z1 ? -1 : S.ri1); S.rn1 = t ? T.ndx : (z1 ? -1 : S.rn1); }
        {   const double cur = z2 ? NI : S.rv2; const bool t = o2 > NI && o2 >= cur;
            S.rv2 = t ? o2 : cur; S.ri2 = t ? i : (z2 ? -1 : S.ri2); S.rn2 = t ? T.ndx : (z2 ? -1 : S.rn2); }
    }
    if (k1 & reached) {
        if (S.f3_cnt == DPL_F3_RING) {
            // the ring is full of forward stops that are still near: its oldest entry is folded early, and the gene begins it is
            // still near to read their near gene ends back from memory
            const DplEnt e = x.f3_get((S.f3_head - S.f3_cnt) & (DPL_F3_RING - 1));
            dpl_max_take(S.f3_far, e.score + M.negc, e.idx, e.ndx);
            if (e.ndx + DPL_NEAR > S.f3_ovf) S.f3_ovf = e.ndx + DPL_NEAR;
            S.f3_cnt--;
            S.f3_old = S.f3_cnt > 0 ? x.f3_get((S.f3_head - S.f3_cnt) & (DPL_F3_RING - 1)).ndx : INT_MAX;
        }
        x.f3_put(S.f3_head & (DPL_F3_RING - 1), DplEnt{B.val, T.ndx, i});
        x.f3t_put(S.f3_head & (DPL_F3_RING - 1), B.tbn);
        S.f3_head = (S.f3_head + 1) & (DPL_F3_RING - 1);
        if (S.f3_cnt == 0) S.f3_old = T.ndx;
        S.f3_cnt++;
        // it may overlap the 3' end of the reverse genes that end at the last reverse stop of a frame
        const DplCand me{B.val, T.ndx, i, B.tbn};
        if (S.l3i0 >= 0 && T.ndx >= S.l3n0 - 4 && T.ndx < S.l3n0 + DPW_MAX_OPP_OVLP - 5) dpl_cand_push(S, x, 0, me);
        if (S.l3i1 >= 0 && T.ndx >= S.l3n1 - 4 && T.ndx < S.l3n1 + DPW_MAX_OPP_OVLP - 5) dpl_cand_push(S, x, 1, me);
        if (S.l3i2 >= 0 && T.ndx >= S.l3n2 - 4 && T.ndx < S.l3n2 + DPW_MAX_OPP_OVLP - 5) dpl_cand_push(S, x, 2, me);
    }
    if (k2 & reached) {
        dpl_max_take(S.r5_all, B.val + M.negc, i, T.ndx);
        if (S.r5_cnt == DPL_R5_RING) {
            const DplEnt e = x.r5_get((S.r5_head - S.r5_cnt) & (DPL_R5_RING - 1));
            dpl_max_take(S.r5_far, e.score + M.negc, e.idx, e.ndx);
            if (e.ndx + DPL_NEAR > S.r5_ovf) S.r5_ovf = e.ndx + DPL_NEAR;
            S.r5_cnt--;
            S.r5_old = S.r5_cnt > 0 ? x.r5_get((S.r5_head - S.r5_cnt) & (DPL_R5_RING - 1)).ndx : INT_MAX;
        }
        x.r5_put(S.r5_head & (DPL_R5_RING - 1), DplEnt{B.val, T.ndx, i});
        S.r5_head = (S.r5_head + 1) & (DPL_R5_RING - 1);
        if (S.r5_cnt == 0) S.r5_old = T.ndx;
        S.r5_cnt++;
    }
    if (k3) {
        const bool w0 = f == 0, w1 = f == 1, w2 = f == 2;
        S.l3v0 = w0 ? B.val : S.l3v0; S.l3i0 = w0 ? i : S.l3i0; S.l3s0 = w0 ? T.stop_val : S.l3s0; S.l3n0 = w0 ? T.ndx : S.l3n0;
        S.l3v1 = w1 ? B.val : S.l3v1; S.l3i1 = w1 ? i : S.l3i1; S.l3s1 = w1 ? T.stop_val : S.l3s1; S.l3n1 = w1 ? T.ndx : S.l3n1;
        S.l3v2 = w2 ? B.val : S.l3v2; S.l3i2 = w2 ? i : S.l3i2; S.l3s2 = w2 ? T.stop_val : S.l3s2; S.l3n2 = w2 ? T.ndx : S.l3n2;
        // the candidate list of the frame starts over: the reached forward stops up to four bases before this stop are in the
        // ring of forward stops (unless one of them left it early: then the list is incomplete and the chain in memory is walked)
        S.cn &= ~(15 << (4 * f));
        if (S.f3_ovf >= T.ndx - 4 + DPL_NEAR) S.cn |= 8 << (4 * f);
        for (int k = S.f3_cnt - 1; k >= 0; k--) {            // oldest first: the list is in position order like the chain
            const int slot = (S.f3_head - 1 - k) & (DPL_F3_RING - 1);
            const DplEnt e = x.f3_get(slot);
            if (e.ndx >= T.ndx - 4) dpl_cand_push(S, x, f, DplCand{e.score, e.ndx, e.idx, x.f3t_get(slot)});
        }
    }
}
