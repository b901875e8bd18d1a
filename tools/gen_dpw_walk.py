"""Generates pyrodigal_amd/csrc/dpw_walk_gfx950.inc: the pair steps of k_dp_wave (dp_wave.hip) as hand-written gfx950 assembly.

Two asm statements, DPW_ASM_NEAR and DPW_ASM_WALK (macros; dp_wave.hip expands them where the compiler's form of the same
loops stood).  Both run the entries of a batch's step schedule (dpw_core.h "Step schedule"): an entry arrives by one
s_load_dwordx16, its lane masks are SGPR pairs, a step is  readlane source value -> v_add_f64 -> v_cmp_ge_f64 -> s_and mask ->
(rarely) three moves under EXEC.  What the compiler made of the C++ form of this loop was 45-55 instructions for a step that
changes nothing; here it is about 20.

The statement pins the registers it needs by halves (running values, the x[] of the stops, source tiles) with "{vN}" constraints
and takes everything else as named operands; scratch registers are clobbers.  `python tools/gen_dpw_walk.py --check` verifies the
gfx950 wait-state rules the assembler does not check (VALU-written SGPR read by a VALU: 2 states; as a lane select: 4; VALU-written
VGPR read by v_readlane: 1) over the control-flow graph of each block; the Makefile regenerates the file when this script changes.

usage: python tools/gen_dpw_walk.py [--check] > pyrodigal_amd/csrc/dpw_walk_gfx950.inc
"""
import os
import re
import sys

EXP = os.environ.get("DPW_EXP", "")          # timing experiments (results wrong): noop | nof3 | nor5 | nor3 | nopull

# ---- fixed registers -------------------------------------------------------------------------------------------------------
VB = int(os.environ.get("DPW_VBASE", "40"))          # the first of the thirty vector registers the statements name (v40 .. v69: with them at v64 .. v93 the
                                                     # kernel spilled 36 bytes per lane at five waves per SIMD, here 12 -- 1.22 -> 1.19 ms per launch -- and could not be built for six)
def _v(k, n=1): return "v%d" % (VB + k) if n == 1 else "v[%d:%d]" % (VB + k, VB + k + n - 1)
PIN = {  # C variable -> pinned VGPRs (both blocks)
    "LV": _v(0, 2), "LT": _v(2), "X0": _v(4, 2), "X1": _v(6, 2), "X2": _v(8, 2),
}
PIN_WALK = {"SVL": _v(28, 2)}
PIN_NEAR = {"NS": _v(18, 2), "NB": _v(20), "NVM": _v(21), "NX0": _v(22, 2), "NX1": _v(24, 2), "NX2": _v(26, 2)}
VT = {"W": _v(10, 2), "W_lo": _v(10), "W_hi": _v(11), "TG": _v(12), "A": _v(13), "MV": _v(14, 2), "MV_lo": _v(14), "MV_hi": _v(15), "MI": _v(16),
      "LV_lo": _v(0), "LV_hi": _v(1), "X0_lo": _v(4), "X0_hi": _v(5), "X1_lo": _v(6), "X1_hi": _v(7), "X2_lo": _v(8), "X2_hi": _v(9),
      "NS_lo": _v(18), "NS_hi": _v(19), "NX0_lo": _v(22), "NX0_hi": _v(23), "NX1_lo": _v(24), "NX1_hi": _v(25), "NX2_lo": _v(26), "NX2_hi": _v(27)}
VT.update({"SVL_lo": _v(28), "SVL_hi": _v(29)})
VT.update(PIN); VT.update(PIN_NEAR); VT.update(PIN_WALK)
V_CLOBBER = [_v(k) for k in range(10, 17)]
def line_regs(base, slot):
    """names of a slot's words inside the sixteen scalar registers of a line from s<base> on: lane, s_ndx, code, j, m[0], m[1]; for a
    forward stop (slot 0 only, the whole line) four more masks behind them"""
    o = base + 8 * slot
    d = {"E%d" % i: "s%d" % (o + i) for i in range(4)}
    d["MA"] = "s[%d:%d]" % (o + 4, o + 5); d["MB"] = "s[%d:%d]" % (o + 6, o + 7)
    if slot == 0:
        for q, nm in enumerate(("MC", "MD", "ME", "MF")): d[nm] = "s[%d:%d]" % (base + 8 + 2 * q, base + 9 + 2 * q)
    d["LINE"] = "s[%d:%d]" % (base, base + 15)
    return d


EP_REGS = "s[16:17]"
def lbase(near):
    """sixteen scalar registers per line in flight.  Near lists are a line or two long: the line at hand and the next one.  The walk's
    list is some twenty lines, and a scalar load that misses the scalar cache takes longer than two slots' steps: TWO lines at hand and
    two on their way (scalar loads return out of order, so every wait is for all of them: what is in flight is half the sets)."""
    return (36, 52) if near else (36, 52, 68, 84)
ST = {"EP": EP_REGS, "EP_lo": "s16", "EP_hi": "s17",
      "SC": "s[18:19]", "SC_lo": "s18", "SC_hi": "s19", "TAGK": "s20", "TMP": "s21", "TM": "s[22:23]", "OK": "s[24:25]",
      "C0": "s[26:27]", "C1": "s[28:29]", "C2": "s[30:31]",
      "SX0": "s[26:27]", "SX0_lo": "s26", "SX0_hi": "s27", "SX1_lo": "s28", "SX1_hi": "s29", "SX2_lo": "s30", "SX2_hi": "s31",
      "SV": "s[26:27]", "SV_lo": "s26", "SV_hi": "s27", "BVS_lo": "s28", "BVS_hi": "s29", "CM": "s[30:31]", "C2b": "s[22:23]", "BI": "s21", "CI": "s35",
      "TBN": "s35", "LHS": "s14", "SVM": "s15"}         # (s32 - s34 and s100 / s101 are the compiler's: stack and frame pointers, scratch)
def s_clobber(near):
    return ["s%d" % i for i in range(18, 32)] + ["s35", "s14", "s15"] + ["s%d" % i for i in range(36, 36 + 16 * len(lbase(near)))] + ["vcc"]


def body(near, a, nxt, other, slot):
    """one slot (in the current register set) through its kind's step; every path ends with a branch to `nxt` (the line's second slot, or
    the other half's loop head); a forward stop takes the whole line and goes on to `other`.  Out-of-line pieces (rare paths) are
    returned as a second list."""
    cold = []
    c = cold.append
    sc_lo, sc_hi = ("{NS_lo}", "{NS_hi}") if near else ("{SVL_lo}", "{SVL_hi}")          # what a gene end offers (-inf while it was never reached)
    r3_lo, r3_hi = ("{NS_lo}", "{NS_hi}") if near else ("{LV_lo}", "{LV_hi}")            # a reverse stop (a gene begin) offers its plain value

    def commit(tag, to=None):
        a("s_mov_b64 exec, vcc")
        a("v_mov_b64_e32 {LV}, {W}")
        if not near: a("v_mov_b64_e32 {SVL}, {W}")
        a("v_mov_b32_e32 {LT}, %s" % tag)
        a("s_mov_b64 exec, -1")
        a("s_branch " + (to or nxt))

    def igm_lookup(dst, mask, label):
        # lanes of `mask` (all within 3 * OPER_DIST bases of the source): dst = igm[d] for d = ndx - s_ndx <= OPER_DIST, 0 beyond.  The
        # table sits in LDS, and an LDS read shares its counter with the scalar load of the next line, which is in flight here: waiting
        # for the one waits for both.  So the schedule says (code bit 6) whether ANY lane of the mask lies within OPER_DIST bases;
        # mostly none does, and the term is 0 without a look-up.
        return ["s_mov_b64 exec, " + mask, "v_mov_b64_e32 " + dst + ", 0", "s_bitcmp0_b32 {E2}, 6", "s_cbranch_scc1 " + label,
                "v_subrev_u32_e32 {A}, {E1}, %[ndx]", "v_cmp_gt_u32_e32 vcc, 61, {A}",
                "s_and_b64 exec, exec, vcc", "v_lshl_add_u32 {A}, {A}, 3, %[igmb]", "ds_read_b64 " + dst + ", {A}", "s_mov_b64 exec, " + mask,
                "s_waitcnt lgkmcnt(0)", label + ":"]

    # ---- R5 (falls in from the dispatch): a reverse start offers score + the intergenic term to the gene begins behind it
    a("v_readlane_b32 {SC_lo}, %s, {E0}" % sc_lo)
    a("v_readlane_b32 {SC_hi}, %s, {E0}" % sc_hi)
    a("s_cmp_eq_u64 {MB}, 0")
    a("s_cbranch_scc0 Lr5tab_%=")
    a("v_add_f64 {W}, {SC}, %[negc]")
    a("Lr5c_%=:")
    a("v_cmp_ge_f64_e32 vcc, {W}, {LV}")
    a("s_and_b64 vcc, vcc, {MA}")
    a("s_cbranch_vccnz Lr5take_%=")              # nobody takes it (two steps in three): straight on into the next slot, no branch taken
    # ---- everything below is out of line: the slots of a line, and the lines of a round, follow each other in the order they are met,
    #      and a reverse start that changes nothing -- the commonest step -- falls from one slot into the next
    a = c
    a("Lr5take_%=:")
    commit("{E3}")
    rest = []
    a = rest.append                              # (the reverse stop's piece comes first: the dispatch of a stop node falls into it)
    c("Lr5tab_%=:")                              # reverse stops within 3 * OPER_DIST bases: the distance term instead of the constant
    c("v_add_f64 {W}, {SC}, %[negc]")
    for l in igm_lookup("{MV}", "{MB}", "Lr5t0_%="): c(l)
    c("v_add_f64 {W}, {SC}, {MV}")
    c("s_mov_b64 exec, -1")
    c("s_branch Lr5c_%=")
    # ---- R3: a reverse stop offers score + cs to the reverse starts of its ORF (MA), score + x[frame] to the reverse stops inside it
    #      that have an overlapping start in its frame (MB & r3v[frame])
    a("Lr3_%=:")
    a("v_readlane_b32 {SC_lo}, %s, {E0}" % r3_lo)
    a("v_readlane_b32 {SC_hi}, %s, {E0}" % r3_hi)
    a("s_bitcmp1_b32 {E2}, 3")
    a("s_cbranch_scc1 Lr3f2_%=")
    a("s_bitcmp1_b32 {E2}, 2")
    a("s_cbranch_scc1 Lr3f1_%=")
    for f, lab in ((0, None), (1, "Lr3f1_%="), (2, "Lr3f2_%=")):
        if lab: a(lab + ":")
        a("s_and_b64 {TM}, {MB}, %%[r3v%d]" % f)
        a("s_or_b64 {TM}, {TM}, {MA}")
        a("s_cbranch_scc0 " + nxt)
        a("v_add_f64 {W}, {SC}, {X%d}" % f)
        if f < 2: a("s_branch Lr3j_%=")
    a("Lr3j_%=:")
    a("s_cmp_eq_u64 {MA}, 0")
    a("s_cbranch_scc1 Lr3c_%=")
    a("s_mov_b64 exec, {MA}")
    a("v_add_f64 {W}, {SC}, %[cs]")
    a("s_mov_b64 exec, -1")
    a("Lr3c_%=:")
    a("v_cmp_ge_f64_e32 vcc, {W}, {LV}")
    a("s_and_b64 vcc, vcc, {TM}")
    a("s_cbranch_vccz " + nxt)
    commit("{E3}")
    if slot == 0:
        # ---- F3: a forward stop; all four kinds of targets
        a("Lf3_%=:")
        if near:
            a("v_readlane_b32 {TBN}, {NB}, {E0}")
        else:
            a("v_readlane_b32 {TAGK}, {LT}, {E0}")
            a("s_cmp_eq_u64 {MF}, 0")
            a("s_cbranch_scc1 Lnopull_%=")
            # pull: the forward starts of its ORF before it in the batch (final by now): (value, index) maximum, ties to the larger index
            a("v_add_f64 {MV}, {LV}, %[cs]")         # what each lane offers as a forward start
            a("v_readlane_b32 {BVS_lo}, {LV_lo}, {E0}")
            a("v_readlane_b32 {BVS_hi}, {LV_hi}, {E0}")
            a("s_and_b32 {BI}, {TAGK}, 0xfffffff")
            a("s_cmp_lt_i32 {TAGK}, 0")
            a("s_cselect_b32 {BI}, -1, {BI}")
            a("s_mov_b64 {CM}, {MF}")
            a("v_mov_b32_e32 {W_lo}, {BVS_lo}")      # the running best, uniform in a VGPR pair (a VALU compare takes one scalar operand)
            a("v_mov_b32_e32 {W_hi}, {BVS_hi}")
            # A candidate wins a tie when its index lies behind the current traceb's.  Candidates come in ascending order, so that is
            # every candidate after the first one taken, and before that the lanes behind lane BI - i0: loop A (strict >) over the lanes
            # at or below it until something is taken, loop B (>=) over everything else.
            a("s_sub_i32 {CI}, {BI}, %[i0]")
            a("s_add_i32 {CI}, {CI}, 1")
            a("s_max_i32 {CI}, {CI}, 0")
            a("s_bfm_b64 {C2b}, {CI}, 0")            # lanes below lane BI - i0 + 1 (BI < i0 + 63 here; before the batch: none)
            a("s_and_b64 {C2b}, {C2b}, {CM}")
            a("s_andn2_b64 {CM}, {CM}, {C2b}")
            a("s_cmp_eq_u64 {C2b}, 0")
            a("s_cbranch_scc1 LpullB_%=")
            a("LpullA_%=:")
            a("s_ff1_i32_b64 {CI}, {C2b}")
            a("s_bitset0_b64 {C2b}, {CI}")
            a("v_readlane_b32 {SV_lo}, {MV_lo}, {CI}")
            a("v_readlane_b32 {SV_hi}, {MV_hi}, {CI}")
            a("s_cmp_lg_u64 {C2b}, 0")
            a("s_nop 0")
            a("v_cmp_gt_f64_e32 vcc, {SV}, {W}")
            a("s_cbranch_vccnz LptakeA_%=")
            a("s_cbranch_scc1 LpullA_%=")
            a("s_branch LpullB_%=")
            a("LptakeA_%=:")
            a("v_mov_b32_e32 {W_lo}, {SV_lo}")
            a("v_mov_b32_e32 {W_hi}, {SV_hi}")
            a("s_add_i32 {TAGK}, {CI}, %[i0]")
            a("s_or_b64 {CM}, {CM}, {C2b}")           # what is left of loop A's lanes goes on in loop B
            a("LpullB_%=:")
            a("s_cmp_eq_u64 {CM}, 0")
            a("s_cbranch_scc1 LpullE_%=")
            a("LpullB1_%=:")
            a("s_ff1_i32_b64 {CI}, {CM}")
            a("s_bitset0_b64 {CM}, {CI}")
            a("v_readlane_b32 {SV_lo}, {MV_lo}, {CI}")
            a("v_readlane_b32 {SV_hi}, {MV_hi}, {CI}")
            a("s_nop 1")
            a("v_cmp_ge_f64_e32 vcc, {SV}, {W}")
            a("s_cbranch_vccz LpullB2_%=")
            a("v_mov_b32_e32 {W_lo}, {SV_lo}")
            a("v_mov_b32_e32 {W_hi}, {SV_hi}")
            a("s_add_i32 {TAGK}, {CI}, %[i0]")
            a("LpullB2_%=:")
            a("s_cmp_lg_u64 {CM}, 0")
            a("s_cbranch_scc1 LpullB1_%=")
            a("LpullE_%=:")
            a("s_cmp_lt_i32 {TAGK}, 0")              # a gene end that was never reached connects to nothing
            a("s_cbranch_scc1 " + other)
            a("s_lshl_b64 {TM}, 1, {E0}")            # (v_writelane with a scalar value AND a scalar lane select is over the constant-bus limit)
            a("s_mov_b64 exec, {TM}")
            a("v_mov_b64_e32 {LV}, {W}")
            a("v_mov_b64_e32 {SVL}, {W}")
            a("v_mov_b32_e32 {LT}, {TAGK}")
            a("s_mov_b64 exec, -1")
            a("Lnopull_%=:")
        a("s_or_b64 {TM}, {MA}, {MC}")
        a("s_or_b64 {TM}, {TM}, {MD}")
        a("s_or_b64 {TM}, {TM}, {ME}")
        a("s_cbranch_scc0 " + other)
        a("v_readlane_b32 {SC_lo}, %s, {E0}" % sc_lo)
        a("v_readlane_b32 {SC_hi}, %s, {E0}" % sc_hi)
        a("s_mov_b64 {OK}, {MA}")                    # forward starts behind it: always admissible
        a("v_mov_b64_e32 {W}, %[negc]")
        a("v_mov_b32_e32 {TG}, {E3}")
        a("s_cmp_eq_u64 {MB}, 0")                    # ... those within 3 * OPER_DIST bases: igm[d] up to OPER_DIST, 0 beyond
        a("s_cbranch_scc1 Lf3a_%=")
        for l in igm_lookup("{W}", "{MB}", "Lf3t0_%="): a(l)
        a("s_mov_b64 exec, -1")
        a("Lf3a_%=:")
        a("s_cmp_eq_u64 {MC}, 0")                    # forward stops whose ORF holds it: through the SOURCE's overlapping start of the lane's frame
        a("s_cbranch_scc1 Lf3b_%=")
        a("v_readlane_b32 {SVM}, %s, {E0}" % ("{NVM}" if near else "%[vm]"))
        a("s_nop 1")
        a("v_and_b32_e32 {A}, {SVM}, %[fbit]")
        a("v_cmp_ne_u32_e32 vcc, 0, {A}")
        a("s_and_b64 {TM}, vcc, {MC}")
        a("s_cbranch_scc0 Lf3b_%=")
        a("s_or_b64 {OK}, {OK}, {TM}")
        xs = ("{NX0_lo}", "{NX0_hi}", "{NX1_lo}", "{NX1_hi}", "{NX2_lo}", "{NX2_hi}") if near else ("{X0_lo}", "{X0_hi}", "{X1_lo}", "{X1_hi}", "{X2_lo}", "{X2_hi}")
        for q, nm in enumerate(("SX0_lo", "SX0_hi", "SX1_lo", "SX1_hi", "SX2_lo", "SX2_hi")):
            a("v_readlane_b32 {%s}, %s, {E0}" % (nm, xs[q]))
        for f in range(3):
            if f: a("s_mov_b64 exec, -1")             # (the compare must see every lane)
            a("v_cmp_eq_u32_e32 vcc, %d, %%[fbit]" % (1 << f))
            a("s_and_b64 exec, vcc, {TM}")
            a("v_mov_b32_e32 {W_lo}, {SX%d_lo}" % f)
            a("v_mov_b32_e32 {W_hi}, {SX%d_hi}" % f)
        a("s_mov_b64 exec, -1")
        a("Lf3b_%=:")
        a("s_or_b64 {TM}, {MD}, {ME}")               # reverse targets: they need the position of the source's own traceb node
        a("s_cbranch_scc0 Lf3d_%=")
        if not near:
            a("s_and_b32 {TMP}, {TAGK}, 0xfffffff")
            a("s_cmp_lt_u32 {TMP}, %[i0]")
            a("s_cbranch_scc1 Lf3pre_%=")
            a("v_readlane_b32 {TBN}, %[ndx], {TMP}")         # inside the batch: lane = index & 63
            a("s_branch Lf3q_%=")
            a("Lf3pre_%=:")
            a("v_readlane_b32 {TBN}, %[tbnpre], {E0}")
            a("Lf3q_%=:")
        a("s_add_i32 {LHS}, {TBN}, {E1}")
        a("s_add_i32 {LHS}, {LHS}, 7")
        a("s_cmp_eq_u64 {MD}, 0")                    # reverse starts whose static interval holds s_ndx: tbn + s_ndx + 7 < drhs0
        a("s_cbranch_scc1 Lf3r3_%=")
        a("v_cmp_lt_i32_e32 vcc, {LHS}, %[drhs0]")
        a("s_and_b64 {TM}, vcc, {MD}")
        a("s_or_b64 {OK}, {OK}, {TM}")
        a("s_mov_b64 exec, {MD}")
        a("v_mov_b64_e32 {W}, %[csd]")
        a("s_mov_b64 exec, -1")
        a("Lf3r3_%=:")
        a("s_cmp_eq_u64 {ME}, 0")                    # reverse stops: through the best admissible overlapping start of the LANE, or directly
        a("s_cbranch_scc1 Lf3d_%=")
        a("s_or_b64 {OK}, {OK}, {ME}")
        # (a candidate is admissible only for dlo < s_ndx < dhi: where no lane's widest interval holds s_ndx, every reverse stop takes
        #  the source directly -- the constant term W already holds -- and the three-candidate evaluation below is skipped)
        a("v_min3_i32 {A}, %[dlo0], %[dlo1], %[dlo2]")
        a("v_cmp_gt_i32_e64 {C0}, {E1}, {A}")
        a("v_max3_i32 {A}, %[dhi0], %[dhi1], %[dhi2]")
        a("v_cmp_lt_i32_e32 vcc, {E1}, {A}")
        a("s_and_b64 {C0}, {C0}, vcc")
        a("s_and_b64 {C0}, {C0}, {ME}")
        a("s_cbranch_scc0 Lf3d_%=")
        a("s_mov_b64 exec, {ME}")
        for q in range(3):
            a("v_cmp_gt_i32_e64 {C%d}, {E1}, %%[dlo%d]" % (q, q))
            a("v_cmp_lt_i32_e32 vcc, {E1}, %%[dhi%d]" % q)
            a("s_and_b64 {C%d}, {C%d}, vcc" % (q, q))
            a("v_cmp_lt_i32_e32 vcc, {LHS}, %%[drhs%d]" % q)
            a("s_and_b64 {C%d}, {C%d}, vcc" % (q, q))
        a("v_mov_b64_e32 {MV}, 0")
        a("v_mov_b32_e32 {MI}, 0")
        for q in range(3):
            a("v_cmp_gt_f64_e32 vcc, {X%d}, {MV}" % q)
            a("s_and_b64 vcc, vcc, {C%d}" % q)
            a("v_cndmask_b32_e32 {MV_lo}, {MV_lo}, {X%d_lo}, vcc" % q)
            a("v_cndmask_b32_e32 {MV_hi}, {MV_hi}, {X%d_hi}, vcc" % q)
            a("v_cndmask_b32_e64 {MI}, {MI}, %d, vcc" % (q + 1))
        a("v_cmp_eq_u32_e32 vcc, 0, {MI}")
        a("v_lshl_or_b32 {TG}, {MI}, 28, {TG}")
        a("v_mov_b64_e32 {W}, %[negc]")
        a("s_andn2_b64 exec, exec, vcc")
        a("v_mov_b64_e32 {W}, {MV}")
        a("s_mov_b64 exec, -1")
        a("Lf3d_%=:")
        a("v_add_f64 {W}, {SC}, {W}")
        a("v_cmp_ge_f64_e32 vcc, {W}, {LV}")
        a("s_and_b64 vcc, vcc, {OK}")
        a("s_cbranch_vccz " + other)
        commit("{TG}", other)
    return rest + cold


def block(near):
    """the instruction list of one block; names in {} are substituted (registers above, %[operand] for the statement's operands).
    Slots are 32 bytes, two to a line; a list starts on a line and ends with an END slot, so the loop has no counter and a line's
    address does not depend on the one before.  The line sets form two halves: while the lines of one half are worked on, the loads
    of the other half's lines are in flight (a step is some twenty instructions; the round trip of a scalar load several hundred
    cycles); every half begins with one s_waitcnt for all of them."""
    LB = lbase(near)
    n = len(LB); h = n // 2                     # lines per half
    out = []
    cold_all = []
    for q in range(h):
        out.append("s_load_dwordx16 {LINE@%d.0}, {EP}, 0x%x" % (q, 64 * q))
    for q in range(n):
        nxt_line = ("LtopH%d_%%=" if (q + 1) % h == 0 else "LlineH%d_%%=") % ((q + 1) % n)      # where the line after this one begins
        for slot in (0, 1):
            L = []
            a = L.append
            nxt = "Lslot1_%=" if slot == 0 else nxt_line       # where a slot's step goes when it is done
            if slot == 0:
                if q % h == 0:
                    a("Ltop_%=:")
                    a("s_waitcnt lgkmcnt(0)")
                    # the other half's lines: the first half's sit at EP + 64 q; the second half's were loaded relative to the old EP
                    for r in range(h):
                        t = (q + h + r) % n                                   # the set that is free now
                        off = 64 * (h + r) if q == 0 else 64 * (n + r)       # q == 0: lines h .. n-1 of this round; q == h: lines 0 .. h-1 of the next
                        a("s_load_dwordx16 {LINE@%d.0}, {EP}, 0x%x" % (t, off))
                    if q == h:
                        a("s_add_u32 {EP_lo}, {EP_lo}, %d" % (64 * n))
                        a("s_addc_u32 {EP_hi}, {EP_hi}, 0")
                else:
                    a("Lline_%=:")
            else:
                a("Lslot1_%=:")
            a("s_bitcmp1_b32 {E2}, 4")           # word 2 = kind | frame << 2 | 16 if not a reverse start; 16 = END, 48 = NOP; kinds 1 = F3, 2 = R5, 3 = R3
            a("s_cbranch_scc1 Lother_%=")
            if EXP in ("noop", "nor5"): a("s_branch " + nxt)
            C = []                               # the slot's out-of-line code
            b = C.append
            b("Lother_%=:")
            b("s_bitcmp1_b32 {E2}, 0")           # bit 0 set: a stop node
            b("s_cbranch_scc1 Lstop_%=")
            b("s_bitcmp1_b32 {E2}, 5")           # NOP: on to the next slot; else END
            b("s_cbranch_scc1 " + nxt)
            b("s_branch Lend%d_%%=" % q)
            b("Lstop_%=:")
            if slot == 0:
                b("s_bitcmp0_b32 {E2}, 1")       # bit 1 clear: F3 (a whole line)
                b("s_cbranch_scc1 " + (nxt_line if EXP in ("noop", "nof3") else "Lf3_%="))
            if EXP in ("noop", "nor3"): b("s_branch " + nxt)
            C += body(near, a, nxt, nxt_line, slot)          # (its first piece is the reverse stop's: Lstop falls into it)
            if q == n - 1 and slot == 1: a("s_branch LtopH0_%=")          # the round's last slot: back to the first line's set
            def rename(l):
                l = re.sub(r"(L\w+?)_%=", lambda m: ("%sH%dS%d_%%=" % (m.group(1), q, slot) if m.group(1) not in ("Ltop", "Lline", "Lslot1") else "%sH%d_%%=" % (m.group(1), q))
                           if not re.fullmatch(r"Lend\d?|LtopH\d|LlineH\d", m.group(1)) else m.group(0), l)
                return re.sub(r"\{(E[0-3]|M[A-F])\}", lambda m: "{%s@%d.%d}" % (m.group(1), q, slot), l)
            out += [rename(l) for l in L]
            cold_all += [rename(l) for l in C]
    out += cold_all
    # the END slot is consumed: the pointer moves to the line behind it.  Line q of the first half sits at EP + 64 q; when a line of the
    # second half is at hand EP has moved on by a whole round already.
    for q in range(n):
        out.append("Lend%d_%%=:" % q)
        d = 64 * (q + 1) if q < h else 64 * (q + 1) - 64 * n
        if d > 0:
            out.append("s_add_u32 {EP_lo}, {EP_lo}, %d" % d); out.append("s_addc_u32 {EP_hi}, {EP_hi}, 0")
        elif d < 0:
            out.append("s_sub_u32 {EP_lo}, {EP_lo}, %d" % -d); out.append("s_subb_u32 {EP_hi}, {EP_hi}, 0")
        if q < n - 1: out.append("s_branch Ldone_%=")
    out.append("Ldone_%=:")
    out.append("s_waitcnt lgkmcnt(0)")            # the loads that are still in flight write registers this statement hands back
    return out


def subst(line, near):
    """register names for the {NAME} / {NAME@set.slot} placeholders of a line of block `near` (the two blocks have different line sets)"""
    def rep(m):
        k = m.group(1)
        if "@" in k:
            nm, _, st = k.partition("@")
            h, _, sl = st.partition(".")
            return line_regs(lbase(near)[int(h)], int(sl))[nm]
        if k in VT: return VT[k]
        if k in ST: return ST[k]
        raise KeyError(k)
    return re.sub(r"\{([\w@.]+)\}", rep, line)


# ---- wait-state check over the control-flow graph -------------------------------------------------------------------------------
def regs_of(tok):
    """registers named by an operand token: 's40', 's[40:41]', 'vcc', 'v64', 'v[64:65]', '%[x]' (named operands count as opaque)"""
    tok = tok.strip()
    m = re.fullmatch(r"([sv])\[(\d+):(\d+)\]", tok)
    if m: return ["%s%d" % (m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    if re.fullmatch(r"[sv]\d+", tok): return [tok]
    if tok == "vcc": return ["vcc"]
    if tok == "exec": return ["exec"]
    m = re.fullmatch(r"%\[(\w+)\](_hi)?", tok)
    if m: return ["%" + m.group(1)]
    return []


def check(lines, name):
    ins = []
    labels = {}
    for l in lines:
        l = subst(l, name == "near")
        if l.endswith(":"):
            labels[l[:-1]] = len(ins)
            continue
        op, _, rest = l.partition(" ")
        ops = [o for o in re.split(r",\s*", rest) if o] if rest else []
        ins.append((op, ops, l))
    n = len(ins)
    succ = [[] for _ in range(n + 1)]
    for i, (op, ops, l) in enumerate(ins):
        if op == "s_branch": succ[i].append(labels[ops[0]])
        elif op.startswith("s_cbranch"): succ[i] += [labels[ops[0]], i + 1]
        else: succ[i].append(i + 1)

    def is_valu(op): return op.startswith("v_")

    def defs_uses(op, ops):
        """(sgprs written by a VALU, sgprs read by a VALU as operands, lane-select sgprs, vgprs written, vgprs read by a lane op)"""
        if not is_valu(op): return [], [], [], [], []
        wr_s, rd_s, lane, wr_v, rd_lane_v = [], [], [], [], []
        if op.startswith("v_readlane"):
            wr_s = regs_of(ops[0]); rd_lane_v = regs_of(ops[1]); lane = [r for r in regs_of(ops[2]) if r[0] == "s"]
        elif op.startswith("v_writelane"):
            wr_v = regs_of(ops[0]); rd_s = [r for r in regs_of(ops[1]) if r[0] in "s"]; lane = [r for r in regs_of(ops[2]) if r[0] == "s"]
        else:
            dst = regs_of(ops[0])
            if op.startswith("v_cmp"): wr_s = dst
            else: wr_v = dst
            for o in ops[1:]:
                rd_s += [r for r in regs_of(o) if r[0] == "s" or r == "vcc"]
            if op.startswith("v_cndmask") and op.endswith("_e32"): rd_s.append("vcc")
        return wr_s, rd_s, lane, wr_v, rd_lane_v

    INF = 9
    # state: reg -> issue slots since a VALU wrote it (SGPRs and VGPRs alike), minimum over paths
    state = [None] * (n + 1)
    state[0] = {}
    work = [0]
    errors = []
    while work:
        i = work.pop()
        if i >= n: continue
        st = dict(state[i])
        op, ops, l = ins[i]
        wr_s, rd_s, lane, wr_v, rd_lane_v = defs_uses(op, ops)
        for r in rd_s:
            if st.get(r, INF) < 2: errors.append("%s: '%s' reads %s %d slot(s) after a VALU wrote it (needs 2)" % (name, l, r, st[r]))
        for r in lane:
            if st.get(r, INF) < 4: errors.append("%s: '%s' takes lane select %s %d slot(s) after a VALU wrote it (needs 4)" % (name, l, r, st[r]))
        for r in rd_lane_v:
            if st.get(r, INF) < 1: errors.append("%s: '%s' reads %s right after a VALU wrote it (needs 1)" % (name, l, r))
        step = 1
        if op == "s_nop": step = int(ops[0]) + 1
        st = {r: d + step for r, d in st.items() if d + step < INF}
        for r in wr_s + wr_v: st[r] = 0
        if op.startswith("s_") and ops and not op.startswith(("s_cmp", "s_cbranch", "s_branch", "s_bitcmp", "s_waitcnt", "s_nop")):
            for r in regs_of(ops[0]): st.pop(r, None)          # rewritten by the scalar unit: interlocked, no wait states
        # the distance counts slots BETWEEN producer and consumer: a consumer right behind it sees 0
        for j in succ[i]:
            old = state[j]
            if old is None: state[j] = dict(st); work.append(j)
            else:
                changed = False
                for r, d in st.items():
                    if old.get(r, INF) > d: old[r] = d; changed = True
                if changed: work.append(j)
    return sorted(set(errors))


def statement(near):
    return [subst(l, near) for l in block(near)]


def c_statement(near):
    body = statement(near)
    nm = "DPW_ASM_NEAR" if near else "DPW_ASM_WALK"
    s = ["#define %s() \\" % nm, "    asm volatile( \\"]
    for l in body:
        s.append('        "%s\\n\\t" \\' % l)
    outs = ['"+{%s}"(a_lv)' % PIN["LV"], '"+{%s}"(a_lt)' % PIN["LT"], '"+{%s}"(a_ep)' % EP_REGS]
    if not near: outs.append('"+{%s}"(a_sv)' % PIN_WALK["SVL"])
    ins = ['"{%s}"(a_x0)' % PIN["X0"], '"{%s}"(a_x1)' % PIN["X1"], '"{%s}"(a_x2)' % PIN["X2"]]
    if near:
        ins += ['"{%s}"(a_ns)' % PIN_NEAR["NS"], '"{%s}"(a_nb)' % PIN_NEAR["NB"], '"{%s}"(a_nvm)' % PIN_NEAR["NVM"], '"{%s}"(a_nx0)' % PIN_NEAR["NX0"],
                '"{%s}"(a_nx1)' % PIN_NEAR["NX1"], '"{%s}"(a_nx2)' % PIN_NEAR["NX2"]]
    else:
        ins += ['[vm] "v"(a_vm)', '[tbnpre] "v"(a_tbnpre)', '[i0] "s"(a_i0)']
    ins += ['[ndx] "v"(a_ndx)', '[fbit] "v"(a_fbit)', '[cs] "v"(a_cs)', '[csd] "v"(a_csd)', '[negc] "v"(a_negc)', '[igmb] "s"(a_igmb)']
    for q in range(3):
        ins += ['[drhs%d] "v"(a_drhs%d)' % (q, q), '[dlo%d] "v"(a_dlo%d)' % (q, q), '[dhi%d] "v"(a_dhi%d)' % (q, q), '[r3v%d] "s"(a_r3v%d)' % (q, q)]
    s.append("        : " + ", ".join(outs) + " \\")
    s.append("        : " + ", ".join(ins) + " \\")
    s.append("        : " + ", ".join('"%s"' % r for r in V_CLOBBER + s_clobber(near)) + ', "memory")')
    return "\n".join(s)


if __name__ == "__main__":
    errs = check(block(True), "near") + check(block(False), "walk")
    if errs:
        sys.stderr.write("\n".join(errs) + "\n")
        sys.exit(1)
    if "--check" in sys.argv:
        print("wait states ok: near %d lines, walk %d lines" % (len(block(True)), len(block(False))))
        sys.exit(0)
    print("// GENERATED by tools/gen_dpw_walk.py -- do not edit; see that script for what the blocks do, their registers and the wait-state check.")
    print(c_statement(True))
    print()
    print(c_statement(False))
