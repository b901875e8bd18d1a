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
import re
import sys

# ---- fixed registers -------------------------------------------------------------------------------------------------------
PIN = {  # C variable -> pinned VGPRs (both blocks)
    "LV": "v[64:65]", "LT": "v66", "X0": "v[68:69]", "X1": "v[70:71]", "X2": "v[72:73]",
}
PIN_NEAR = {"NS": "v[82:83]", "NB": "v84", "NVM": "v85", "NX0": "v[86:87]", "NX1": "v[88:89]", "NX2": "v[90:91]"}
VT = {"W": "v[74:75]", "W_lo": "v74", "W_hi": "v75", "TG": "v76", "A": "v77", "MV": "v[78:79]", "MV_lo": "v78", "MV_hi": "v79", "MI": "v80",
      "LV_lo": "v64", "LV_hi": "v65", "X0_lo": "v68", "X0_hi": "v69", "X1_lo": "v70", "X1_hi": "v71", "X2_lo": "v72", "X2_hi": "v73",
      "NS_lo": "v82", "NS_hi": "v83", "NX0_lo": "v86", "NX0_hi": "v87", "NX1_lo": "v88", "NX1_hi": "v89", "NX2_lo": "v90", "NX2_hi": "v91"}
VT.update(PIN); VT.update(PIN_NEAR)
V_CLOBBER = ["v74", "v75", "v76", "v77", "v78", "v79", "v80"]
E = ["s%d" % (36 + i) for i in range(16)]           # the entry: lane, s_ndx, code, j, six masks
ST = {"E0": E[0], "E1": E[1], "E2": E[2], "E3": E[3],
      "EP": "s[34:35]", "EP_lo": "s34", "EP_hi": "s35",
      "MA": "s[40:41]", "MB": "s[42:43]", "MC": "s[44:45]", "MD": "s[46:47]", "ME": "s[48:49]", "MF": "s[50:51]", "ENT": "s[36:51]",
      "SC": "s[52:53]", "SC_lo": "s52", "SC_hi": "s53", "TAGK": "s54", "TMP": "s55", "TM": "s[56:57]", "OK": "s[58:59]",
      "C0": "s[60:61]", "C1": "s[62:63]", "C2": "s[64:65]",
      "SX0": "s[60:61]", "SX0_lo": "s60", "SX0_hi": "s61", "SX1_lo": "s62", "SX1_hi": "s63", "SX2_lo": "s64", "SX2_hi": "s65",
      "SV": "s[60:61]", "SV_lo": "s60", "SV_hi": "s61", "BVS_lo": "s62", "BVS_hi": "s63", "CM": "s[64:65]", "BI": "s55", "CI": "s66",
      "TBN": "s66", "LHS": "s67", "SVM": "s68"}
S_CLOBBER = ["s%d" % i for i in range(36, 69)] + ["vcc"]


def block(near):
    """the instruction list of one block; names in {} are substituted (registers above, %[operand] for the statement's operands)"""
    L = []
    a = L.append
    src_score_lo, src_score_hi = ("{NS_lo}", "{NS_hi}") if near else ("{LV_lo}", "{LV_hi}")
    # ---- loop head
    a("s_cmp_lt_i32 %[left], 1")
    a("s_cbranch_scc1 Lend_%=")
    a("Lloop_%=:")
    a("s_load_dwordx16 {ENT}, {EP}, 0x0")
    a("s_waitcnt lgkmcnt(0)")
    if near:
        a("s_cmp_ge_i32 {E3}, %[tend]")          # the entry's source sits in the next tile: leave it for the next call
        a("s_cbranch_scc1 Lend_%=")
    a("s_bitcmp0_b32 {E2}, 1")                   # word 2 = kind | frame << 2; kinds: 1 = F3 (01), 2 = R5 (10), 3 = R3 (11); word 0 = the lane
    a("s_cbranch_scc1 Lf3_%=")
    a("s_bitcmp1_b32 {E2}, 0")
    a("s_cbranch_scc1 Lr3_%=")
    # ---- R5: a reverse start offers score + the intergenic term to the gene begins behind it
    a("s_add_u32 {EP_lo}, {EP_lo}, 32")
    a("s_addc_u32 {EP_hi}, {EP_hi}, 0")
    if near:
        a("v_readlane_b32 {TAGK}, {NB}, {E0}")   # position of the source's traceb node; -1: never reached, no source
    else:
        a("v_readlane_b32 {TAGK}, {LT}, {E0}")   # the source's tag; < 0: never reached
    a("v_readlane_b32 {SC_lo}, %s, {E0}" % src_score_lo)
    a("v_readlane_b32 {SC_hi}, %s, {E0}" % src_score_hi)
    a("s_cmp_lt_i32 {TAGK}, 0")
    a("s_cbranch_scc1 Lnext_%=")
    a("v_add_f64 {W}, {SC}, %[negc]")
    a("s_cmp_eq_u64 {MB}, 0")
    a("s_cbranch_scc1 Lr5c_%=")
    a("s_mov_b64 exec, {MB}")                    # reverse stops within 3 * OPER_DIST bases: igm[d] up to OPER_DIST, 0 beyond
    a("v_subrev_u32_e32 {A}, {E1}, %[ndx]")
    a("v_mov_b64_e32 {MV}, 0")
    a("v_cmp_gt_u32_e32 vcc, 61, {A}")
    a("s_and_b64 exec, exec, vcc")
    a("v_lshl_add_u32 {A}, {A}, 3, %[igmb]")
    a("ds_read_b64 {MV}, {A}")
    a("s_mov_b64 exec, {MB}")
    a("s_waitcnt lgkmcnt(0)")
    a("v_add_f64 {W}, {SC}, {MV}")
    a("s_mov_b64 exec, -1")
    a("Lr5c_%=:")
    a("v_cmp_ge_f64_e32 vcc, {W}, {LV}")
    a("s_and_b64 vcc, vcc, {MA}")
    a("s_cbranch_vccz Lnext_%=")
    a("s_mov_b64 exec, vcc")
    a("v_mov_b64_e32 {LV}, {W}")
    a("v_mov_b32_e32 {LT}, {E3}")
    a("s_mov_b64 exec, -1")
    a("s_branch Lnext_%=")
    # ---- R3: a reverse stop offers score + cs to the reverse starts of its ORF (MA), score + x[frame] to the reverse stops inside it
    #      that have an overlapping start in its frame (MB & r3v[frame])
    a("Lr3_%=:")
    a("s_add_u32 {EP_lo}, {EP_lo}, 32")
    a("s_addc_u32 {EP_hi}, {EP_hi}, 0")
    a("v_readlane_b32 {SC_lo}, %s, {E0}" % src_score_lo)
    a("v_readlane_b32 {SC_hi}, %s, {E0}" % src_score_hi)
    a("s_bitcmp1_b32 {E2}, 3")
    a("s_cbranch_scc1 Lr3f2_%=")
    a("s_bitcmp1_b32 {E2}, 2")
    a("s_cbranch_scc1 Lr3f1_%=")
    for f, lab in ((0, None), (1, "Lr3f1_%="), (2, "Lr3f2_%=")):
        if lab:
            a(lab + ":")
        a("s_and_b64 {TM}, {MB}, %%[r3v%d]" % f)
        a("s_or_b64 {TM}, {TM}, {MA}")
        a("s_cbranch_scc0 Lnext_%=")
        a("v_add_f64 {W}, {SC}, {X%d}" % f)
        if f < 2:
            a("s_branch Lr3j_%=")
    a("Lr3j_%=:")
    a("s_cmp_eq_u64 {MA}, 0")
    a("s_cbranch_scc1 Lr3c_%=")
    a("s_mov_b64 exec, {MA}")
    a("v_add_f64 {W}, {SC}, %[cs]")
    a("s_mov_b64 exec, -1")
    a("Lr3c_%=:")
    a("v_cmp_ge_f64_e32 vcc, {W}, {LV}")
    a("s_and_b64 vcc, vcc, {TM}")
    a("s_cbranch_vccz Lnext_%=")
    a("s_mov_b64 exec, vcc")
    a("v_mov_b64_e32 {LV}, {W}")
    a("v_mov_b32_e32 {LT}, {E3}")
    a("s_mov_b64 exec, -1")
    a("s_branch Lnext_%=")
    # ---- F3: a forward stop; all four kinds of targets
    a("Lf3_%=:")
    a("s_add_u32 {EP_lo}, {EP_lo}, 64")
    a("s_addc_u32 {EP_hi}, {EP_hi}, 0")
    if near:
        a("v_readlane_b32 {TBN}, {NB}, {E0}")
        a("s_cmp_eq_u32 {TBN}, -1")
        a("s_cbranch_scc1 Lnext_%=")
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
        a("Lpull_%=:")
        a("s_ff1_i32_b64 {CI}, {CM}")
        a("s_bitset0_b64 {CM}, {CI}")
        a("v_readlane_b32 {SV_lo}, {MV_lo}, {CI}")
        a("v_readlane_b32 {SV_hi}, {MV_hi}, {CI}")
        a("s_add_i32 {CI}, {CI}, %[i0]")
        a("s_nop 0")
        a("v_cmp_gt_f64_e32 vcc, {SV}, {W}")
        a("s_cbranch_vccnz Lptake_%=")
        a("v_cmp_eq_f64_e32 vcc, {SV}, {W}")
        a("s_cbranch_vccz Lpnext_%=")
        a("s_cmp_gt_i32 {CI}, {BI}")
        a("s_cbranch_scc0 Lpnext_%=")
        a("Lptake_%=:")
        a("v_mov_b32_e32 {W_lo}, {SV_lo}")
        a("v_mov_b32_e32 {W_hi}, {SV_hi}")
        a("s_mov_b32 {BI}, {CI}")
        a("s_mov_b32 {TAGK}, {CI}")
        a("Lpnext_%=:")
        a("s_cmp_lg_u64 {CM}, 0")
        a("s_cbranch_scc1 Lpull_%=")
        a("s_lshl_b64 {TM}, 1, {E0}")             # (v_writelane with a scalar value AND a scalar lane select is over the constant-bus limit)
        a("s_mov_b64 exec, {TM}")
        a("v_mov_b64_e32 {LV}, {W}")
        a("v_mov_b32_e32 {LT}, {TAGK}")
        a("s_mov_b64 exec, -1")
        a("Lnopull_%=:")
        a("s_cmp_lt_i32 {TAGK}, 0")              # a gene end that was never reached connects to nothing
        a("s_cbranch_scc1 Lnext_%=")
    a("s_or_b64 {TM}, {MA}, {MC}")
    a("s_or_b64 {TM}, {TM}, {MD}")
    a("s_or_b64 {TM}, {TM}, {ME}")
    a("s_cbranch_scc0 Lnext_%=")
    a("v_readlane_b32 {SC_lo}, %s, {E0}" % src_score_lo)
    a("v_readlane_b32 {SC_hi}, %s, {E0}" % src_score_hi)
    a("s_mov_b64 {OK}, {MA}")                    # forward starts behind it: always admissible
    a("v_mov_b64_e32 {W}, %[negc]")
    a("v_mov_b32_e32 {TG}, {E3}")
    a("s_cmp_eq_u64 {MB}, 0")                    # ... those within 3 * OPER_DIST bases: igm[d] up to OPER_DIST, 0 beyond
    a("s_cbranch_scc1 Lf3a_%=")
    a("s_mov_b64 exec, {MB}")
    a("v_subrev_u32_e32 {A}, {E1}, %[ndx]")
    a("v_mov_b64_e32 {W}, 0")
    a("v_cmp_gt_u32_e32 vcc, 61, {A}")
    a("s_and_b64 exec, exec, vcc")
    a("v_lshl_add_u32 {A}, {A}, 3, %[igmb]")
    a("ds_read_b64 {W}, {A}")
    a("s_waitcnt lgkmcnt(0)")
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
    a("s_cbranch_vccz Lnext_%=")
    a("s_mov_b64 exec, vcc")
    a("v_mov_b64_e32 {LV}, {W}")
    a("v_mov_b32_e32 {LT}, {TG}")
    a("s_mov_b64 exec, -1")
    # ---- next entry
    a("Lnext_%=:")
    a("s_add_i32 %[left], %[left], -1")
    a("s_cmp_lg_u32 %[left], 0")
    a("s_cbranch_scc1 Lloop_%=")
    a("Lend_%=:")
    return L


def subst(line):
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
    m = re.fullmatch(r"%\[(\w+)\](_hi)?", tok)
    if m: return ["%" + m.group(1)]
    return []


def check(lines, name):
    ins = []
    labels = {}
    for l in lines:
        l = subst(l)
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
    return [subst(l) for l in block(near)]


def c_statement(near):
    body = statement(near)
    nm = "DPW_ASM_NEAR" if near else "DPW_ASM_WALK"
    s = ["#define %s() \\" % nm, "    asm volatile( \\"]
    for l in body:
        s.append('        "%s\\n\\t" \\' % l)
    outs = ['"+{v[64:65]}"(a_lv)', '"+{v66}"(a_lt)', '"+{s[34:35]}"(a_ep)', '[left] "+s"(a_left)']
    ins = ['"{v[68:69]}"(a_x0)', '"{v[70:71]}"(a_x1)', '"{v[72:73]}"(a_x2)']
    if near:
        ins += ['"{v[82:83]}"(a_ns)', '"{v84}"(a_nb)', '"{v85}"(a_nvm)', '"{v[86:87]}"(a_nx0)', '"{v[88:89]}"(a_nx1)', '"{v[90:91]}"(a_nx2)', '[tend] "s"(a_tend)']
    else:
        ins += ['[vm] "v"(a_vm)', '[tbnpre] "v"(a_tbnpre)', '[i0] "s"(a_i0)']
    ins += ['[ndx] "v"(a_ndx)', '[fbit] "v"(a_fbit)', '[cs] "v"(a_cs)', '[csd] "v"(a_csd)', '[negc] "v"(a_negc)', '[igmb] "s"(a_igmb)']
    for q in range(3):
        ins += ['[drhs%d] "v"(a_drhs%d)' % (q, q), '[dlo%d] "v"(a_dlo%d)' % (q, q), '[dhi%d] "v"(a_dhi%d)' % (q, q), '[r3v%d] "s"(a_r3v%d)' % (q, q)]
    s.append("        : " + ", ".join(outs) + " \\")
    s.append("        : " + ", ".join(ins) + " \\")
    s.append("        : " + ", ".join('"%s"' % r for r in V_CLOBBER + S_CLOBBER) + ', "memory")')
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
