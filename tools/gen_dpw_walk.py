"""Generates pyrodigal_amd/csrc/dpw_walk_gfx950.inc: the pair steps of k_dp_wave (dp_wave.hip) as hand-written gfx950 assembly.

Two asm statements, DPW_ASM_NEAR and DPW_ASM_WALK (macros; dp_wave.hip expands them where the compiler's form of the same
loops stood).  Round 6: both run from the per-NODE words of the step schedule (dpw_core.h "Step schedule") that every lane holds
in vector registers -- lane k's W0 = the lanes source k reaches, W1 = those whose distance term comes from the table -- instead of
lists of 32-byte slots fetched line by line through the scalar cache (that fetch was 12 200 of the walk's 20 000 cycles per batch:
ten dependent scalar-load round trips).  A step is now
    s_ff1 (next source lane) -> v_readlane x 2 (its reach mask) -> dispatch on the source's kind (lane masks of the batch's kinds,
    s_bitcmp1_b64) -> v_readlane x 2 (its value) -> v_add_f64 -> v_cmp_ge_f64 -> s_and mask -> (rarely) three moves under EXEC,
and nothing in it waits for memory.  The masks of a step are taken apart by target kind with scalar ANDs (a reverse stop reaches
reverse starts and reverse stops, a forward stop all four kinds: the relations are disjoint by the target's kind, so one word holds
their union; a forward stop's word also holds the forward starts it pulls, which sit BEFORE it).

The statement pins the registers it needs by halves (running values, the x[] of the stops, source values, the words) with "{vN}"
constraints and takes everything else as named operands; scratch registers are clobbers.  `python tools/gen_dpw_walk.py --check`
verifies the gfx950 wait-state rules the assembler does not check (VALU-written SGPR read by a VALU: 2 states; as a lane select: 4;
VALU-written VGPR read by v_readlane: 1) over the control-flow graph of each block; the Makefile regenerates the file when this
script changes.

usage: python tools/gen_dpw_walk.py [--check] > pyrodigal_amd/csrc/dpw_walk_gfx950.inc
"""
import os
import re
import sys

EXP = os.environ.get("DPW_EXP", "")          # timing experiments (results wrong): noop | nof3 | nor5 | nor3

# ---- fixed registers -------------------------------------------------------------------------------------------------------
VB = int(os.environ.get("DPW_VBASE", "40"))          # the first of the vector registers the statements name
def _v(k, n=1): return "v%d" % (VB + k) if n == 1 else "v[%d:%d]" % (VB + k, VB + k + n - 1)
PIN = {  # C variable -> pinned VGPRs (both blocks)
    "LV": _v(0, 2), "LT": _v(2), "X0": _v(4, 2), "X1": _v(6, 2), "X2": _v(8, 2), "W0": _v(30, 2), "W1": _v(32, 2),
}
PIN_WALK = {"SVL": _v(28, 2)}
PIN_NEAR = {"NS": _v(18, 2), "NB": _v(20)}
VT = {"W": _v(10, 2), "W_lo": _v(10), "W_hi": _v(11), "TG": _v(12), "A": _v(13), "MV": _v(14, 2), "MV_lo": _v(14), "MV_hi": _v(15), "MI": _v(16), "FB": _v(17),
      "LV_lo": _v(0), "LV_hi": _v(1), "X0_lo": _v(4), "X0_hi": _v(5), "X1_lo": _v(6), "X1_hi": _v(7), "X2_lo": _v(8), "X2_hi": _v(9),
      "NS_lo": _v(18), "NS_hi": _v(19),
      "W0_lo": _v(30), "W0_hi": _v(31), "W1_lo": _v(32), "W1_hi": _v(33)}
VT.update({"SVL_lo": _v(28), "SVL_hi": _v(29)})
VT.update(PIN); VT.update(PIN_NEAR); VT.update(PIN_WALK)
VT.update({"DLM": _v(34), "DHM": _v(35)})       # per lane: the widest of the three candidate intervals (min dlo, max dhi)
V_CLOBBER = [_v(k) for k in range(10, 18)] + [_v(34), _v(35)]

def pair(n): return "s[%d:%d]" % (n, n + 1)
ST = {"SC": pair(18), "SC_lo": "s18", "SC_hi": "s19", "TAGK": "s20", "TMP": "s21", "TM": pair(22), "OK": pair(24),
      "C0": pair(26), "C1": pair(28), "C2": pair(30),
      "SX0": pair(26), "SX0_lo": "s26", "SX0_hi": "s27", "SX1_lo": "s28", "SX1_hi": "s29", "SX2_lo": "s30", "SX2_hi": "s31",
      "SV": pair(26), "SV_lo": "s26", "SV_hi": "s27", "BVS_lo": "s28", "BVS_hi": "s29", "CM": pair(30), "C2b": pair(22), "BI": "s21", "CI": "s35",
      "TBN": "s35", "LHS": "s14", "SVM": "s15",         # (s32 - s34 and s100 / s101 are the compiler's: stack and frame pointers, scratch)
      "TODO": pair(16),
      "E0": "s36", "E1": "s37", "E3": "s38", "PSL": "s39",      # the source: its lane, its position, its chain index; LDS address of the pull's scratch double
      # (MA, MC, MD, ME: one pair -- a step needs them one after the other; MB its own: a reverse stop's step holds MA and MB together)
      "MA": pair(40), "MB": pair(42), "MB_lo": "s42", "MB_hi": "s43", "MC": pair(40), "MD": pair(40), "ME": pair(40), "MF": pair(44),
      "RW": pair(46), "RW_lo": "s46", "RW_hi": "s47",
      "K0M": pair(48), "K1M": pair(50), "K2M": pair(52), "K3M": pair(54),         # the targets' kinds as lane masks
      "SK2": pair(56), "SK3": pair(58), "SK2T": pair(60),                         # the sources': plain reverse starts, reverse stops, reverse starts with a W1 (lane 63 in none)
      "TABM": pair(62), "GBM": pair(64)}                                          # sources with a W1; gene begins among the targets
S_LAST = 65
def s_clobber(near):
    return ["s14", "s15"] + ["s%d" % i for i in range(16, 32)] + ["s35"] + ["s%d" % i for i in range(36, S_LAST + 1)] + ["vcc"]


def block(near):
    """the instruction list of one block; names in {} are substituted (registers above, %[operand] for the statement's operands)"""
    out = []
    cold = []
    a = out.append
    c = cold.append
    kinfo = "%[pk]" if near else "%[kinfo]"           # the SOURCES' kind | frame << 2 | 0x80 if no node | vm << 8
    sc_lo, sc_hi = ("{NS_lo}", "{NS_hi}") if near else ("{SVL_lo}", "{SVL_hi}")          # what a gene end offers (-inf while it was never reached)
    r3_lo, r3_hi = ("{NS_lo}", "{NS_hi}") if near else ("{LV_lo}", "{LV_hi}")            # a reverse stop (a gene begin) offers its plain value
    src_ndx = "%[pndx]" if near else "%[ndx]"
    nxt = "Lnext_%="

    # ---- prologue: the lane masks of the batch.  Targets' kinds (K0M .. K3M), reverse stops with an overlapping start in frame f (R3V0 .. 2),
    #      1 << frame per lane (FB); sources' kinds and frames (the batch's own in the walk, the batch before in the near steps);
    #      the sources that reach anything (TODO) and those with a distance-term mask (TABM)
    a("v_or_b32_e32 {A}, {W0_lo}, {W0_hi}")
    a("v_cmp_ne_u32_e64 {TODO}, 0, {A}")
    a("v_or_b32_e32 {A}, {W1_lo}, {W1_hi}")
    a("v_cmp_ne_u32_e64 {TABM}, 0, {A}")
    a("s_cmp_eq_u64 {TODO}, 0")
    a("s_cbranch_scc1 Ldone_%=")
    a("v_and_b32_e32 {A}, 0x83, %[kinfo]")
    a("v_cmp_eq_u32_e64 {K0M}, 0, {A}")
    a("v_cmp_eq_u32_e64 {K1M}, 1, {A}")
    a("v_cmp_eq_u32_e64 {K2M}, 2, {A}")
    a("v_cmp_eq_u32_e64 {K3M}, 3, {A}")
    a("v_bfe_u32 {A}, %[kinfo], 2, 2")
    a("v_lshlrev_b32_e64 {FB}, {A}, 1")
    if near:
        a("v_and_b32_e32 {A}, 0x83, %[pk]")
        a("v_cmp_eq_u32_e64 {SK2}, 2, {A}")
        a("v_cmp_eq_u32_e64 {SK3}, 3, {A}")
    else:
        a("s_mov_b64 {SK2}, {K2M}")
        a("s_mov_b64 {SK3}, {K3M}")
    # A step never asks whether it was the last one: when the list is empty s_ff1 yields -1, which the dispatch reads as lane 63, and
    # lane 63 is no reverse node in the masks the dispatch tests -- so the end of the list, like a real source in lane 63, arrives at
    # the forward-stop path, which tells them apart (Llast).  SK2: the PLAIN reverse starts (no reverse stop within 3 * OPER_DIST
    # bases): they reach every gene begin behind them, so their mask is arithmetic on the lane number, not a word to read.
    a("s_bitset0_b64 {SK2}, 63")
    a("s_bitset0_b64 {SK3}, 63")
    a("s_and_b64 {SK2T}, {SK2}, {TABM}")
    a("s_andn2_b64 {SK2}, {SK2}, {TABM}")
    a("s_or_b64 {GBM}, {K0M}, {K3M}")
    a("v_min3_i32 {DLM}, %[dlo0], %[dlo1], %[dlo2]")
    a("v_max3_i32 {DHM}, %[dhi0], %[dhi1], %[dhi2]")
    if not near: a("s_add_i32 {PSL}, %[igmb], 496")      # igm[62] of the LDS table: nobody reads past igm[60]

    def commit(to, tag, where=None):
        e = where or a
        e("s_mov_b64 exec, vcc")
        e("v_mov_b64_e32 {LV}, {W}")
        if not near: e("v_mov_b64_e32 {SVL}, {W}")
        e("v_mov_b32_e32 {LT}, %s" % tag)
        e("s_mov_b64 exec, -1")
        e("s_branch " + to)

    def chain_index(e):
        # the source's chain index: i0 + lane (the walk), i0 - 64 + lane (the batch before)
        e("s_add_i32 {E3}, {E0}, %[i0]")
        if near: e("s_sub_i32 {E3}, {E3}, 64")

    def igm_lookup(e, dst, mask, label):
        # lanes of `mask` (all within 3 * OPER_DIST bases of the source): dst = igm[d] for d = ndx - s_ndx <= OPER_DIST, 0 beyond (LDS table)
        for l in ["s_mov_b64 exec, " + mask, "v_mov_b64_e32 " + dst + ", 0",
                  "v_subrev_u32_e32 {A}, {E1}, %[ndx]", "v_cmp_gt_u32_e32 vcc, 61, {A}",
                  "s_and_b64 exec, exec, vcc", "s_cbranch_execz " + label, "v_lshl_add_u32 {A}, {A}, 3, %[igmb]", "ds_read_b64 " + dst + ", {A}",
                  "s_waitcnt lgkmcnt(0)", label + ":", "s_mov_b64 exec, " + mask]:
            e(l)

    # ---- the loop: next source lane, its reach mask, dispatch on its kind
    a("Lloop_%=:")
    a("s_ff1_i32_b64 {E0}, {TODO}")
    a("s_bitset0_b64 {TODO}, {E0}")
    a("s_bitcmp1_b64 {SK2}, {E0}")
    a("s_cbranch_scc0 Lstop_%=")
    if EXP in ("noop", "nor5"): a("s_branch Lloop_%=")
    # ---- R5 (plain): a reverse start offers score + the constant intergenic term to every gene begin behind it
    a("v_readlane_b32 {SC_lo}, %s, {E0}" % sc_lo)
    a("v_readlane_b32 {SC_hi}, %s, {E0}" % sc_hi)
    if not near: a("s_lshl_b64 {RW}, -2, {E0}")          # the lanes behind it (the batch before lies behind every lane)
    a("s_nop 0" if not near else "s_nop 1")
    a("v_add_f64 {W}, {SC}, %[negc]")
    a("v_cmp_ge_f64_e32 vcc, {W}, {LV}")
    if not near: a("s_and_b64 vcc, vcc, {RW}")
    a("s_and_b64 vcc, vcc, {GBM}")
    a("s_cbranch_vccnz Lr5take_%=")              # nobody takes it (two steps in three): straight back
    a("s_branch Lloop_%=")
    a(nxt + ":")                                 # (the other kinds come back here)
    a("s_branch Lloop_%=")
    c("Lr5take_%=:")
    chain_index(c)
    commit("Lloop_%=", "{E3}", c)
    # ---- a reverse start with reverse stops within 3 * OPER_DIST bases (W1): to those the distance term instead of the constant
    c("Lr5tab_%=:")
    c("v_readlane_b32 {RW_lo}, {W0_lo}, {E0}")
    c("v_readlane_b32 {RW_hi}, {W0_hi}, {E0}")
    c("v_readlane_b32 {SC_lo}, %s, {E0}" % sc_lo)
    c("v_readlane_b32 {SC_hi}, %s, {E0}" % sc_hi)
    c("v_readlane_b32 {MB_lo}, {W1_lo}, {E0}")
    c("v_readlane_b32 {MB_hi}, {W1_hi}, {E0}")
    c("v_readlane_b32 {E1}, %s, {E0}" % src_ndx)
    c("v_add_f64 {W}, {SC}, %[negc]")
    igm_lookup(c, "{MV}", "{MB}", "Lr5t0_%=")
    c("v_add_f64 {W}, {SC}, {MV}")
    c("s_mov_b64 exec, -1")
    c("v_cmp_ge_f64_e32 vcc, {W}, {LV}")
    c("s_and_b64 vcc, vcc, {RW}")
    c("s_cbranch_vccnz Lr5take_%=")
    c("s_branch Lloop_%=")
    # ---- a stop node
    a("Lstop_%=:")
    a("s_bitcmp1_b64 {SK3}, {E0}")
    a("s_cbranch_scc1 Lr3_%=")
    a("s_bitcmp1_b64 {SK2T}, {E0}")
    a("s_cbranch_scc1 " + (nxt if EXP in ("noop", "nor5") else "Lr5tab_%="))
    a("s_cmp_lt_u32 {E0}, 63")
    a("s_cbranch_scc1 Lf3_%=")
    # the end of the list (-1), or a source in lane 63 by its true kind
    a("s_cmp_lt_i32 {E0}, 0")
    a("s_cbranch_scc1 Ldone_%=")
    a("v_readlane_b32 {SVM}, %s, {E0}" % kinfo)   # kind | frame << 2 | ...
    a("s_and_b32 {TMP}, {SVM}, 0x83")
    a("s_cmp_eq_u32 {TMP}, 3")
    a("s_cbranch_scc1 Lr3_%=")
    a("s_cmp_eq_u32 {TMP}, 2")
    a("s_cbranch_scc1 Lr5tab_%=")                 # (the path that reads the words: right for a plain reverse start as well)
    a("s_branch Lf3_%=")
    a("Lr3_%=:")
    if EXP in ("noop", "nor3"): a("s_branch " + nxt)
    a("v_readlane_b32 {RW_lo}, {W0_lo}, {E0}")
    a("v_readlane_b32 {RW_hi}, {W0_hi}, {E0}")
    # ---- R3: a reverse stop offers score + cs to the reverse starts of its frame inside its ORF (MA), score + x[frame] to the reverse
    #      stops inside it that have an overlapping start in its frame (MB & r3v[frame])
    a("v_readlane_b32 {SC_lo}, %s, {E0}" % r3_lo)
    a("v_readlane_b32 {SC_hi}, %s, {E0}" % r3_hi)
    a("v_readlane_b32 {SVM}, %s, {E0}" % kinfo)   # the source's frame: bits 2, 3
    a("s_and_b64 {MA}, {RW}, {K2M}")
    a("s_and_b64 {MB}, {RW}, {K3M}")
    a("s_bitcmp1_b32 {SVM}, 3")
    a("s_cbranch_scc1 Lr3f2_%=")
    a("s_bitcmp1_b32 {SVM}, 2")
    a("s_cbranch_scc1 Lr3f1_%=")
    for f, lab in ((0, None), (1, "Lr3f1_%="), (2, "Lr3f2_%=")):
        if lab: a(lab + ":")
        a("s_mov_b64 {TM}, {MA}")
        a("s_cmp_eq_u64 {MB}, 0")
        a("s_cbranch_scc1 Lr3n%d_%%=" % f)
        a("v_and_b32_e32 {A}, 0x%x, %%[kinfo]" % (0x100 << f))      # the reverse stops among them that have an overlapping start in its frame
        a("v_cmp_ne_u32_e32 vcc, 0, {A}")
        a("s_and_b64 {TM}, vcc, {MB}")
        a("s_or_b64 {TM}, {TM}, {MA}")
        a("Lr3n%d_%%=:" % f)
        a("s_cmp_eq_u64 {TM}, 0")
        a("s_cbranch_scc1 " + nxt)
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
    chain_index(a)
    commit(nxt, "{E3}")
    # ---- F3: a forward stop.  What it offers nearly every lane it reaches is score + the constant intergenic term, as a reverse start does:
    #      that is the straight path.  The exceptions branch off only where the source's word says they exist -- forward starts within
    #      3 * OPER_DIST bases (the distance term), forward stops whose ORF holds it (an operon through ITS overlapping start), reverse starts
    #      whose interval holds its position, reverse stops with an admissible overlapping start -- and each prices its own lanes under EXEC.
    a("Lf3_%=:")
    if EXP in ("noop", "nof3"): a("s_branch " + nxt)
    a("v_readlane_b32 {RW_lo}, {W0_lo}, {E0}")
    a("v_readlane_b32 {RW_hi}, {W0_hi}, {E0}")
    if not near:
        a("s_lshl_b64 {TM}, -1, {E0}")           # lanes from the source on: the forward starts behind it are targets, those before it are pulled
        a("v_readlane_b32 {TAGK}, {LT}, {E0}")
        a("s_and_b64 {MA}, {RW}, {K0M}")
        a("s_andn2_b64 {MF}, {MA}, {TM}")
        a("s_cbranch_scc0 Lnopull_%=")
        # pull: the forward starts of its ORF before it in the batch (final by now): (value, index) maximum, ties to the larger index.
        # Not a loop over the candidates -- the lane's own value goes to a scratch double in LDS, the candidates' lanes add theirs with
        # ONE ds_max_f64, every lane reads the maximum back; the candidates that hold it are a vote, the last of them the index (ties
        # to the larger index), and whether it beats the lane's own traceb is one more compare.
        a("v_add_f64 {MV}, {LV}, %[cs]")         # what each lane offers as a forward start
        a("v_mov_b32_e32 {A}, {PSL}")
        a("s_lshl_b64 {C0}, 1, {E0}")
        a("s_mov_b64 exec, {C0}")
        a("ds_write_b64 {A}, {LV}")
        a("s_mov_b64 exec, {MF}")
        a("ds_max_f64 {A}, {MV}")
        a("s_mov_b64 exec, -1")
        a("ds_read_b64 {W}, {A}")
        a("s_and_b32 {BI}, {TAGK}, 0xfffffff")
        a("s_cmp_lt_i32 {TAGK}, 0")
        a("s_cselect_b32 {BI}, -1, {BI}")
        a("s_waitcnt lgkmcnt(0)")
        a("v_cmp_eq_f64_e32 vcc, {W}, {MV}")     # the candidates that hold the maximum
        a("s_and_b64 {CM}, vcc, {MF}")
        a("s_cbranch_scc0 LpullE_%=")             # none: the lane's own value is larger than every candidate's
        a("s_flbit_i32_b64 {CI}, {CM}")
        a("s_sub_i32 {CI}, 63, {CI}")             # the last of them
        a("v_cmp_gt_f64_e32 vcc, {W}, {LV}")      # (in lane E0: the maximum beats the lane's own value)
        a("s_add_i32 {TMP}, {CI}, %[i0]")
        a("s_bitcmp1_b64 vcc, {E0}")
        a("s_cbranch_scc1 Lptake_%=")
        a("s_cmp_gt_i32 {TMP}, {BI}")             # equal values: the larger index
        a("s_cbranch_scc0 LpullE_%=")
        a("Lptake_%=:")
        a("s_mov_b32 {TAGK}, {TMP}")
        a("v_readlane_b32 {SV_lo}, {MV_lo}, {CI}")      # the winner's own bits
        a("v_readlane_b32 {SV_hi}, {MV_hi}, {CI}")
        a("s_lshl_b64 {C2}, 1, {E0}")
        a("s_mov_b64 exec, {C2}")
        a("v_mov_b32_e32 {LV_lo}, {SV_lo}")
        a("v_mov_b32_e32 {LV_hi}, {SV_hi}")
        a("v_mov_b32_e32 {SVL_lo}, {SV_lo}")
        a("v_mov_b32_e32 {SVL_hi}, {SV_hi}")
        a("v_mov_b32_e32 {LT}, {TAGK}")
        a("s_mov_b64 exec, -1")
        a("LpullE_%=:")
        a("s_cmp_lt_i32 {TAGK}, 0")              # a gene end that was never reached connects to nothing
        a("s_cbranch_scc1 " + nxt)
        a("Lnopull_%=:")
    a("v_readlane_b32 {SC_lo}, %s, {E0}" % sc_lo)
    a("v_readlane_b32 {SC_hi}, %s, {E0}" % sc_hi)
    a("v_readlane_b32 {E1}, %s, {E0}" % src_ndx)
    a("s_and_b64 {OK}, {RW}, {GBM}")             # the gene begins it reaches
    if not near: a("s_and_b64 {OK}, {OK}, {TM}")  # (behind it; the forward starts before it were the pulled ones)
    a("v_add_f64 {W}, {SC}, %[negc]")
    a("s_bitcmp1_b64 {TABM}, {E0}")
    a("s_cbranch_scc1 Lf3tab_%=")
    a("Lf3a_%=:")
    a("s_and_b64 {MC}, {RW}, {K1M}")
    a("s_cbranch_scc1 Lf3op_%=")
    a("Lf3b_%=:")
    a("s_and_b64 {MD}, {RW}, {K2M}")
    a("s_cbranch_scc1 Lf3md_%=")
    a("Lf3c_%=:")
    a("s_and_b64 {ME}, {RW}, {K3M}")
    a("s_cbranch_scc0 Lf3d_%=")
    # reverse stops: directly (the constant term, W holds it), or through the best admissible overlapping start of the LANE -- admissible
    # only for dlo < s_ndx < dhi: where no lane's widest interval holds s_ndx (DLM / DHM: per lane, once per batch) nothing is to do
    a("v_cmp_gt_i32_e64 {C0}, {E1}, {DLM}")
    a("v_cmp_lt_i32_e32 vcc, {E1}, {DHM}")
    a("s_and_b64 {C0}, {C0}, vcc")
    a("s_and_b64 {C0}, {C0}, {ME}")
    a("s_cbranch_scc1 Lf3cand_%=")
    a("Lf3d_%=:")
    a("v_cmp_ge_f64_e32 vcc, {W}, {LV}")
    a("s_and_b64 vcc, vcc, {OK}")
    a("s_cbranch_vccz " + nxt)
    chain_index(a)
    commit(nxt, "{E3}")

    def tbn_of(e):
        # the position of the source's own traceb node
        if near:
            e("v_readlane_b32 {TBN}, {NB}, {E0}")
        else:
            e("s_and_b32 {TMP}, {TAGK}, 0xfffffff")
            e("s_cmp_lt_u32 {TMP}, %[i0]")
            e("s_cbranch_scc1 Lf3pre%d_%%=" % tbn_of.n)
            e("v_readlane_b32 {TBN}, %[ndx], {TMP}")         # inside the batch: lane = index & 63
            e("s_branch Lf3q%d_%%=" % tbn_of.n)
            e("Lf3pre%d_%%=:" % tbn_of.n)
            e("v_readlane_b32 {TBN}, %[tbnpre], {E0}")
            e("Lf3q%d_%%=:" % tbn_of.n)
            tbn_of.n += 1
        e("s_add_i32 {LHS}, {TBN}, {E1}")
        e("s_add_i32 {LHS}, {LHS}, 7")
    tbn_of.n = 0

    # forward starts within 3 * OPER_DIST bases: igm[d] up to OPER_DIST, 0 beyond
    c("Lf3tab_%=:")
    c("v_readlane_b32 {MB_lo}, {W1_lo}, {E0}")
    c("v_readlane_b32 {MB_hi}, {W1_hi}, {E0}")
    igm_lookup(c, "{MV}", "{MB}", "Lf3t0_%=")
    c("v_add_f64 {W}, {SC}, {MV}")
    c("s_mov_b64 exec, -1")
    c("s_branch Lf3a_%=")
    # forward stops whose ORF holds it: through the SOURCE's overlapping start of the lane's frame
    c("Lf3op_%=:")
    c("v_readlane_b32 {SVM}, %s, {E0}" % kinfo)
    c("s_lshr_b32 {SVM}, {SVM}, 8")              # vm sits in bits 8 .. 10
    c("s_nop 0")
    c("v_and_b32_e32 {A}, {SVM}, {FB}")
    c("v_cmp_ne_u32_e32 vcc, 0, {A}")
    c("s_and_b64 {TM}, vcc, {MC}")
    c("s_cbranch_scc0 Lf3b_%=")
    c("s_or_b64 {OK}, {OK}, {TM}")
    xs = ("{X0}", "{X1}", "{X2}")
    for f in range(3):
        # the lanes of frame f among them, if the source has an overlapping start of that frame at all
        c("s_bitcmp1_b32 {SVM}, %d" % f)
        c("s_cbranch_scc0 Lf3o%d_%%=" % f)
        c("v_cmp_eq_u32_e32 vcc, %d, {FB}" % (1 << f))
        c("s_and_b64 {C1}, vcc, {TM}")
        c("s_cbranch_scc0 Lf3o%d_%%=" % f)
        if near:
            # (the x[] of a node of the batch before: in LDS, s_px[lane][3])
            c("s_mul_i32 {TMP}, {E0}, 24")
            c("s_add_i32 {TMP}, {TMP}, %[pxb]")
            c("v_mov_b32_e32 {A}, {TMP}")
            c("s_mov_b64 exec, {C1}")
            c("ds_read_b64 {W}, {A} offset:%d" % (8 * f))
            c("s_waitcnt lgkmcnt(0)")
        else:
            c("v_readlane_b32 {SX0_lo}, %s, {E0}" % xs[f].replace("}", "_lo}"))
            c("v_readlane_b32 {SX0_hi}, %s, {E0}" % xs[f].replace("}", "_hi}"))
            c("s_mov_b64 exec, {C1}")
            c("s_nop 0")
            c("v_mov_b64_e32 {W}, {SX0}")            # (one scalar operand per instruction)
        c("v_add_f64 {W}, {SC}, {W}")
        c("s_mov_b64 exec, -1")
        c("Lf3o%d_%%=:" % f)
    if not near: c("s_lshl_b64 {TM}, -1, {E0}")
    c("s_branch Lf3b_%=")
    # reverse starts whose static interval holds s_ndx: tbn + s_ndx + 7 < drhs0
    c("Lf3md_%=:")
    tbn_of(c)
    c("v_cmp_lt_i32_e32 vcc, {LHS}, %[drhs0]")
    c("s_and_b64 {C0}, vcc, {MD}")
    c("s_or_b64 {OK}, {OK}, {C0}")
    c("s_mov_b64 exec, {MD}")
    c("v_add_f64 {W}, %[cs], %[negc]")          # cs + negc
    c("v_add_f64 {W}, {SC}, {W}")
    c("s_mov_b64 exec, -1")
    c("s_branch Lf3c_%=")
    # reverse stops with an interval that may hold s_ndx: the three candidates of each lane
    c("Lf3cand_%=:")
    tbn_of(c)
    c("s_mov_b64 exec, {ME}")
    for q in range(3):
        c("v_cmp_gt_i32_e64 {C%d}, {E1}, %%[dlo%d]" % (q, q))
        c("v_cmp_lt_i32_e32 vcc, {E1}, %%[dhi%d]" % q)
        c("s_and_b64 {C%d}, {C%d}, vcc" % (q, q))
        c("v_cmp_lt_i32_e32 vcc, {LHS}, %%[drhs%d]" % q)
        c("s_and_b64 {C%d}, {C%d}, vcc" % (q, q))
    c("v_mov_b64_e32 {MV}, 0")
    c("v_mov_b32_e32 {MI}, 0")
    for q in range(3):
        c("v_cmp_gt_f64_e32 vcc, {X%d}, {MV}" % q)
        c("s_and_b64 vcc, vcc, {C%d}" % q)
        c("v_cndmask_b32_e32 {MV_lo}, {MV_lo}, {X%d_lo}, vcc" % q)
        c("v_cndmask_b32_e32 {MV_hi}, {MV_hi}, {X%d_hi}, vcc" % q)
        c("v_cndmask_b32_e64 {MI}, {MI}, %d, vcc" % (q + 1))
    c("s_mov_b64 exec, -1")
    chain_index(c)
    c("v_lshlrev_b32_e32 {TG}, 28, {MI}")
    c("v_or_b32_e32 {TG}, {E3}, {TG}")           # the tag of a lane that takes it: index | (candidate + 1) << 28 (lanes outside ME: MI is stale, mask it)
    c("v_cmp_ne_u32_e32 vcc, 0, {MI}")
    c("s_and_b64 {C0}, vcc, {ME}")               # the reverse stops that go through a candidate
    c("s_andn2_b64 {C1}, -1, {C0}")
    c("s_mov_b64 exec, {C1}")
    c("v_mov_b32_e32 {TG}, {E3}")                # everyone else: the plain index
    c("s_mov_b64 exec, {C0}")
    c("v_add_f64 {W}, {SC}, {MV}")
    c("s_mov_b64 exec, -1")
    c("v_cmp_ge_f64_e32 vcc, {W}, {LV}")
    c("s_and_b64 vcc, vcc, {OK}")
    c("s_cbranch_vccz " + nxt)
    commit(nxt, "{TG}", c)
    out += cold
    out.append("Ldone_%=:")
    return out


def subst(line, near):
    """register names for the {NAME} placeholders of a line"""
    def rep(m):
        k = m.group(1)
        if k in VT: return VT[k]
        if k in ST: return ST[k]
        raise KeyError(k)
    return re.sub(r"\{(\w+)\}", rep, line)


# ---- wait-state check over the control-flow graph -------------------------------------------------------------------------------
def regs_of(tok):
    """registers named by an operand token: 's40', 's[40:41]', 'vcc', 'v64', 'v[64:65]', '%[x]' (named operands count as opaque)"""
    tok = tok.strip()
    m = re.fullmatch(r"([sv])\[(\d+):(\d+)\]", tok)
    if m: return ["%s%d" % (m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    if re.fullmatch(r"[sv]\d+", tok): return [tok]
    if tok == "vcc": return ["vcc"]
    if tok == "exec": return ["exec"]
    m = re.fullmatch(r"%\[(\w+)\]", tok)
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
    outs = ['"+{%s}"(a_lv)' % PIN["LV"], '"+{%s}"(a_lt)' % PIN["LT"]]
    if not near: outs.append('"+{%s}"(a_sv)' % PIN_WALK["SVL"])
    ins = ['"{%s}"(a_x0)' % PIN["X0"], '"{%s}"(a_x1)' % PIN["X1"], '"{%s}"(a_x2)' % PIN["X2"], '"{%s}"(a_w0)' % PIN["W0"], '"{%s}"(a_w1)' % PIN["W1"]]
    if near:
        ins += ['"{%s}"(a_ns)' % PIN_NEAR["NS"], '"{%s}"(a_nb)' % PIN_NEAR["NB"], '[pk] "v"(a_pk)', '[pndx] "v"(a_pndx)', '[pxb] "s"(a_pxb)']
    else:
        ins += ['[tbnpre] "v"(a_tbnpre)']
    ins += ['[kinfo] "v"(a_kinfo)', '[i0] "s"(a_i0)', '[ndx] "v"(a_ndx)', '[cs] "v"(a_cs)', '[negc] "v"(a_negc)', '[igmb] "s"(a_igmb)']
    for q in range(3):
        ins += ['[drhs%d] "v"(a_drhs%d)' % (q, q), '[dlo%d] "v"(a_dlo%d)' % (q, q), '[dhi%d] "v"(a_dhi%d)' % (q, q)]
    assert len(outs) + len(ins) <= 30, "an asm statement takes at most 30 operands"
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
