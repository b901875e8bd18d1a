#!/usr/bin/env python3
"""Generates pyrodigal_amd/csrc/dpc_walk_gfx950.inc: the fast walk of the contig-per-wavefront connection scorer (dp_contig.hip) as
one block of gfx950 assembly.

Why assembly: a wavefront issues at most one instruction every four cycles, whatever its kind, so what a node costs is its
instruction COUNT, scalar ones included; the compiler's rendering of dpc_core.h's fast routines is 320 instructions per node
(108 vector, 164 scalar, 39 branches), three times what the routines need.  The block below is those routines written out by hand:
dpc_cand_f5 / _f3 / _r5 / _r3 and dpc_finish_* of dpc_core.h, node after node, driven by the compiled records (DpcProg).  It stops
in front of every node the fast routines do not cover (the caller walks that one with the C++ routines and comes back) and at the
end of the run of nodes it was given (the caller flushes results and asks for the next group of cs between runs).

    python tools/gen_dpc_walk.py > pyrodigal_amd/csrc/dpc_walk_gfx950.inc

Registers (fixed; the block clobbers them): see the tables below.  LDS layout: struct DpcLds of dp_contig.hip (offsets checked there
by static_asserts against the constants below)."""

HIST, CAND, CARRY, L3V, CS, EXT, T2 = 0, 4096, 6400, 6784, 6976, 8000, 10048
DPC_CAND = 6

# ---- scalar registers
P = lambda k: "s%d" % (36 + k)            # the node's record: w0..w7 in s36-s43, w8..w15 (reverse stops) in s44-s51
KIND, FRAME = "s60", "s61"
IM32 = "s80"                              # i - 32: bit k of a mask is node i - 1 - k = IM32 + (leading zeros of the bit)
PROG = "s[84:85]"; WMASK = "s[88:89]"; EXECSAVE = "s[82:83]"
# ---- vector registers
R5A, R5A_I, R5F, R5F_I, F3F, F3F_I = (28, 29), 30, (32, 33), 31, (34, 35), 36
END, END_I, END_TB = (38, 39), 37, 40
CIDX, CNDX, TID, L16, L8 = 41, 42, 43, 44, 45
NEGC, STWT, EXTP, NI = (46, 47), (48, 49), (50, 51), (52, 53)
B, BOV, BTB = (54, 55), 56, 57            # v[54:57] is a history entry {val, tbn / ov, tb}
SV, SVTBN, SVTAG = (58, 59), 60, 61       # v[58:61]: a history / list entry {sv, tbn, tag}
CARRYR = (62, 65)
CSR, TA, TB_, RD = (66, 67), (68, 69), (70, 71), (72, 73)    # RD: v[72:75] holds a 16-byte entry read back, v[72:73] its score
X = [(76, 77), (78, 79), (80, 81)]; N3N = [82, 83, 84]; N3S = [85, 86, 87]; CQ = [88, 89, 90]; VM = 91
VI, VA, VJ, VC = 92, 93, 94, 95
VDUMMY = 27


def v2(p): return "v[%d:%d]" % (p[0], p[1])
def v4(a): return "v[%d:%d]" % (a, a + 3)


class Gen:
    def __init__(self):
        self.lines = []
        self.n = 0
        self.smem_free = True          # no scalar load is in flight besides the ones the code at hand waits for anyway

    def e(self, s=""):
        for ln in s.strip("\n").split("\n"):
            ln = ln.strip()
            if ln:
                self.lines.append(ln)

    def label(self, stem):
        self.n += 1
        return "L%s%d_%%=" % (stem, self.n)

    # ---- snippets ------------------------------------------------------------------------------------------------------
    def hist_read64(self, sj, dst):
        """sv of node sj (scalar register) into dst (pair)"""
        self.e(f"""
            s_and_b32 s65, {sj}, 31
            s_lshl_b32 s65, s65, 7
            v_add_u32_e32 v{VA}, s65, v{L16}
            ds_read_b64 {v2(dst)}, v{VA} offset:{HIST}""")

    def bit_desc(self, mask):
        """highest set bit of `mask`: s63 = its index, s64 = the node it names; the bit is cleared"""
        self.e(f"""
            s_flbit_i32_b32 s62, {mask}
            s_sub_i32 s63, 31, s62
            s_bitset0_b32 {mask}, s63
            s_add_i32 s64, s62, {IM32}""")

    def fold(self, mask, mx, mx_i):
        """the gene ends named by `mask` join the running maximum (mx, mx_i): dpc_fold"""
        done, loop = self.label("fd"), self.label("fl")
        self.e(f"s_cmp_eq_u32 {mask}, 0\ns_cbranch_scc1 {done}\n{loop}:")
        self.bit_desc(mask)
        self.hist_read64("s64", RD)
        self.e(f"v_mov_b32_e32 v{VJ}, s64")
        self.wait_lds()
        self.e(f"""
            v_cmp_lg_f64_e32 vcc, {v2(RD)}, {v2(NI)}
            v_add_f64 {v2(TA)}, {v2(RD)}, {v2(NEGC)}
            v_cmp_ge_f64_e64 s[66:67], {v2(TA)}, {v2(mx)}
            s_and_b64 vcc, vcc, s[66:67]
            v_cndmask_b32_e32 v{mx[0]}, v{mx[0]}, v{TA[0]}, vcc
            v_cndmask_b32_e32 v{mx[1]}, v{mx[1]}, v{TA[1]}, vcc
            v_cndmask_b32_e32 v{mx_i}, v{mx_i}, v{VJ}, vcc
            s_cmp_lg_u32 {mask}, 0
            s_cbranch_scc1 {loop}
            {done}:""")

    def wait_lds(self):
        """the LDS reads issued so far have arrived.  The next node's record (ONE scalar load) is in flight through the whole node, and
        scalar loads return out of order with LDS operations, so lgkmcnt(0) would wait for it as well -- its whole trip to the L2, at every
        node.  With one more (dummy) LDS read behind the ones that matter, lgkmcnt(1) leaves either that read or the scalar load
        outstanding: LDS operations complete in order, so the reads that matter are done in both cases."""
        if self.smem_free:
            self.e("s_waitcnt lgkmcnt(0)")
        else:
            self.e(f"ds_read_b32 v{VDUMMY}, v{L16} offset:{T2}\ns_waitcnt lgkmcnt(1)")

    def take_asc(self, val, vj, extra=None):
        """B takes (val, vj) when val >= B.val [and extra]: dpc_take_asc (the candidate is later in the chain than what B holds)"""
        self.e(f"v_cmp_ge_f64_e32 vcc, {v2(val)}, {v2(B)}")
        if extra:
            self.e(f"s_and_b64 vcc, vcc, {extra}")
        self.e(f"""
            v_cndmask_b32_e32 v{B[0]}, v{B[0]}, v{val[0]}, vcc
            v_cndmask_b32_e32 v{B[1]}, v{B[1]}, v{val[1]}, vcc
            v_cndmask_b32_e32 v{BTB}, v{BTB}, v{vj}, vcc""")

    def lex_cond(self, val, vj, extra=None):
        """vcc = (val, vj) > (B.val, B.tb) lexicographically [and extra]"""
        self.e(f"""
            v_cmp_gt_f64_e32 vcc, {v2(val)}, {v2(B)}
            v_cmp_eq_f64_e64 s[66:67], {v2(val)}, {v2(B)}
            v_cmp_gt_i32_e64 s[68:69], v{vj}, v{BTB}
            s_and_b64 s[66:67], s[66:67], s[68:69]
            s_or_b64 vcc, vcc, s[66:67]""")
        if extra:
            self.e(f"s_and_b64 vcc, vcc, {extra}")

    def take_lex(self, val, vj, extra=None):
        self.lex_cond(val, vj, extra)
        self.e(f"""
            v_cndmask_b32_e32 v{B[0]}, v{B[0]}, v{val[0]}, vcc
            v_cndmask_b32_e32 v{B[1]}, v{B[1]}, v{val[1]}, vcc
            v_cndmask_b32_e32 v{BTB}, v{BTB}, v{vj}, vcc""")

    def table_term(self, dst):
        """dst = igm(ndx - ndx of node s64) = t2[d] * st_wt (x.igm(P.w[1] - x.ndx_of(j)))"""
        self.e(f"""
            s_lshl_b32 s66, s64, 6
            s_add_u32 s70, s84, s66
            s_addc_u32 s71, s85, 0
            s_load_dword s66, s[70:71], 0x4
            s_waitcnt lgkmcnt(0)
            s_sub_i32 s66, {P(1)}, s66
            s_lshl_b32 s66, s66, 3
            s_add_i32 s66, s66, %[base]
            v_mov_b32_e32 v{VA}, s66
            ds_read_b64 {v2(dst)}, v{VA} offset:{T2}
            s_waitcnt lgkmcnt(0)
            v_mul_f64 {v2(dst)}, {v2(dst)}, {v2(STWT)}""")

    def near_loop(self, mask, tab, lex, plain_negc=False):
        """the near gene ends named by `mask`, pair by pair; those of `tab` with the distance term of the table"""
        done, loop, notab, go = self.label("nd"), self.label("nl"), self.label("nt"), self.label("ng")
        self.e(f"s_cmp_eq_u32 {mask}, 0\ns_cbranch_scc1 {done}\n{loop}:")
        self.bit_desc(mask)
        self.hist_read64("s64", RD)
        self.e(f"v_mov_b32_e32 v{VJ}, s64")
        if plain_negc:
            self.wait_lds()
            self.e(f"v_add_f64 {v2(TA)}, {v2(RD)}, {v2(NEGC)}")
        else:
            self.e(f"s_bitcmp1_b32 {tab}, s63\ns_cbranch_scc0 {notab}")
            self.table_term(TB_)
            self.e(f"v_add_f64 {v2(TA)}, {v2(RD)}, {v2(TB_)}\ns_branch {go}\n{notab}:")
            self.wait_lds()
            self.e(f"v_add_f64 {v2(TA)}, {v2(RD)}, 0\n{go}:")
        if lex:
            self.take_lex(TA, VJ)
        else:
            self.take_asc(TA, VJ)
        self.e(f"s_cmp_lg_u32 {mask}, 0\ns_cbranch_scc1 {loop}\n{done}:")

    def cs_read(self):
        """cs of node i for this lane's model into CSR: s_cs[((i & 15) >> 1) * 128 + (i & 1) * 8 + ml * 16]"""
        self.e(f"""
            s_bfe_u32 s62, %[i], 0x30001
            s_lshl_b32 s62, s62, 7
            s_bitcmp1_b32 %[i], 0
            s_cselect_b32 s63, 8, 0
            s_add_i32 s62, s62, s63
            v_add_u32_e32 v{VA}, s62, v{L16}
            ds_read_b64 {v2(CSR)}, v{VA} offset:{CS}""")

    def hist_write(self, entry):
        self.e(f"""
            s_and_b32 s62, %[i], 31
            s_lshl_b32 s62, s62, 7
            v_add_u32_e32 v{VA}, s62, v{L16}
            ds_write_b128 v{VA}, {v4(entry)} offset:{HIST}""")

    def writers(self):
        self.e(f"s_mov_b64 {EXECSAVE}, exec\ns_mov_b64 exec, {WMASK}")

    def everyone(self):
        self.e(f"s_mov_b64 exec, {EXECSAVE}")

    def note_end(self):
        """dpc_note_end: a gene end at least as good as the best so far becomes the best"""
        self.e(f"""
            v_cmp_ge_f64_e32 vcc, {v2(B)}, {v2(END)}
            v_cndmask_b32_e32 v{END[0]}, v{END[0]}, v{B[0]}, vcc
            v_cndmask_b32_e32 v{END[1]}, v{END[1]}, v{B[1]}, vcc
            v_cndmask_b32_e32 v{END_I}, v{END_I}, v{VI}, vcc
            v_cndmask_b32_e32 v{END_TB}, v{END_TB}, v{BTB}, vcc""")

    def ext_issue(self):
        """the extras of stop nstop + 2 are asked for, straight into LDS slot (nstop + 2) & 3 (after nstop was advanced): four 16-byte
        parts per lane; the instruction offset moves the LDS address as well, hence m0 = slot + 112 part"""
        self.e(f"""
            s_add_i32 s62, %[nstop], 2
            s_and_b32 s63, s62, 3
            s_lshl_b32 s63, s63, 9
            s_add_i32 s63, s63, {EXT}
            s_add_i32 s63, s63, %[base]
            s_lshl_b32 s62, s62, 6
            v_add_co_u32_e32 v{TB_[0]}, vcc, s62, v{EXTP[0]}
            v_addc_co_u32_e32 v{TB_[1]}, vcc, 0, v{EXTP[1]}, vcc""")
        self.writers()
        self.e(f"""
            s_mov_b32 m0, s63
            s_nop 0
            global_load_lds_dwordx4 {v2(TB_)}, off
            s_add_i32 m0, m0, 112
            s_nop 0
            global_load_lds_dwordx4 {v2(TB_)}, off offset:16
            s_add_i32 m0, m0, 112
            s_nop 0
            global_load_lds_dwordx4 {v2(TB_)}, off offset:32
            s_add_i32 m0, m0, 112
            s_nop 0
            global_load_lds_dwordx4 {v2(TB_)}, off offset:48""")
        self.everyone()

    def ext_read(self, all_parts):
        """the record of the stop being walked (asked for three stops ago: the two records asked for since, four loads each, may still
        be on their way)"""
        self.e(f"""
            s_waitcnt vmcnt(8)
            s_and_b32 s62, %[nstop], 3
            s_lshl_b32 s62, s62, 9
            v_add_u32_e32 v{VA}, s62, v{L16}
            ds_read_b128 {v4(76)}, v{VA} offset:{EXT}
            ds_read_b128 {v4(80)}, v{VA} offset:{EXT + 128}
            ds_read_b128 {v4(88)}, v{VA} offset:{EXT + 384}""")
        if all_parts:
            self.e(f"ds_read_b128 {v4(84)}, v{VA} offset:{EXT + 256}")

    def window_check(self, far_i, exit_label):
        """dpc_need_slow_begin, the part the record cannot know: the window start has passed the argmax of a running maximum"""
        skip = self.label("nw")
        self.e(f"""
            s_bitcmp1_b32 {P(0)}, 10
            s_cbranch_scc0 {skip}
            v_cmp_gt_i32_e32 vcc, {P(3)}, v{far_i}
            v_cmp_le_i32_e64 s[66:67], 0, v{far_i}
            s_and_b64 s[66:67], s[66:67], vcc
            v_cmp_gt_i32_e32 vcc, {P(3)}, v{F3F_I}
            v_cmp_le_i32_e64 s[68:69], 0, v{F3F_I}
            s_and_b64 vcc, vcc, s[68:69]
            s_or_b64 vcc, vcc, s[66:67]
            s_cmp_lg_u64 vcc, 0
            s_cbranch_scc1 {exit_label}
            {skip}:""")


def generate():
    g = Gen()
    e = g.e
    EXIT, NEXT, NODE = "Lexit_%=", "Lnext_%=", "Lnode_%="
    LF5, LR5, LF3 = "Lf5_%=", "Lr5_%=", "Lf3_%="

    # ---- entry: the caller's values into the block's registers
    ins = [(R5A[0], "r5a_lo"), (R5A[1], "r5a_hi"), (R5A_I, "r5a_i"), (R5F[0], "r5f_lo"), (R5F[1], "r5f_hi"), (R5F_I, "r5f_i"),
           (F3F[0], "f3f_lo"), (F3F[1], "f3f_hi"), (F3F_I, "f3f_i"), (END[0], "end_lo"), (END[1], "end_hi"), (END_I, "end_i"),
           (END_TB, "end_tb"), (CIDX, "cidx"), (CNDX, "cndx")]
    consts = [(TID, "tid"), (NEGC[0], "negc_lo"), (NEGC[1], "negc_hi"), (STWT[0], "stwt_lo"), (STWT[1], "stwt_hi"), (EXTP[0], "extp_lo"),
              (EXTP[1], "extp_hi")]
    for r, name in ins + consts:
        e(f"v_mov_b32_e32 v{r}, %[{name}]")
    e(f"""
        v_and_b32_e32 v{L16}, 7, v{TID}
        v_lshlrev_b32_e32 v{L8}, 3, v{L16}
        v_lshlrev_b32_e32 v{L16}, 4, v{L16}
        v_add_u32_e32 v{L8}, %[base], v{L8}
        v_add_u32_e32 v{L16}, %[base], v{L16}
        v_mov_b32_e32 v{NI[0]}, 0
        v_mov_b32_e32 v{NI[1]}, 0xfff00000
        s_mov_b32 s84, %[prog_lo]
        s_mov_b32 s85, %[prog_hi]
        s_mov_b64 {WMASK}, 0xff
        s_cmp_ge_i32 %[i], %[end]
        s_cbranch_scc1 {EXIT}
        s_lshl_b32 s62, %[i], 6
        s_load_dwordx8 s[36:43], {PROG}, s62
        s_waitcnt lgkmcnt(0)""")

    # ---- a node: its record is in s36-s43; the next one's is asked for
    nopf = g.label("np")
    e(f"""
        {NODE}:
        s_add_i32 s62, %[i], 1
        s_cmp_lt_i32 s62, %[end]
        s_cbranch_scc0 {nopf}
        s_lshl_b32 s62, s62, 6
        s_load_dwordx8 s[52:59], {PROG}, s62
        {nopf}:
        s_bitcmp1_b32 {P(0)}, 8
        s_cbranch_scc1 {EXIT}
        s_and_b32 {KIND}, {P(0)}, 3
        s_bfe_u32 {FRAME}, {P(0)}, 0x20002
        s_sub_i32 {IM32}, %[i], 32
        v_mov_b32_e32 v{VI}, %[i]
        s_cmp_eq_u32 {KIND}, 0
        s_cbranch_scc1 {LF5}
        s_cmp_eq_u32 {KIND}, 2
        s_cbranch_scc1 {LR5}
        s_cmp_eq_u32 {KIND}, 1
        s_cbranch_scc1 {LF3}""")

    # ================================================================ reverse stop (falls through from the dispatch)
    e(f"""
        s_lshl_b32 s62, %[i], 6
        s_add_u32 s70, s84, s62
        s_addc_u32 s71, s85, 0
        s_load_dwordx8 s[44:51], s[70:71], 0x20""")
    g.ext_read(True)
    e("s_waitcnt lgkmcnt(0)")
    g.window_check(R5F_I, EXIT)
    # dpc_need_slow_r3: a lane's overlapping start whose gene's stop is not the frame's last reverse stop (or whose list is incomplete)
    # while its static chain of candidates is not empty
    e("s_mov_b64 s[86:87], 0")
    for q in range(3):
        nolist, acc = g.label("rl"), g.label("ra")
        e(f"""
            v_and_b32_e32 v{VJ}, {1 << q}, v{VM}
            v_cmp_ne_u32_e32 vcc, 0, v{VJ}
            v_cmp_gt_i32_e64 s[66:67], %[i], v{CQ[q]}
            s_and_b64 vcc, vcc, s[66:67]
            s_cmp_lt_i32 {P(9 + q)}, 0
            s_cbranch_scc1 {acc}
            s_bitcmp1_b32 {P(15)}, {3 + q}
            s_cbranch_scc1 {acc}
            v_cmp_ne_u32_e64 s[66:67], {P(12 + q)}, v{N3S[q]}
            s_and_b64 vcc, vcc, s[66:67]
            {acc}:
            s_or_b64 s[86:87], s[86:87], vcc""")
    e(f"s_cmp_lg_u64 s[86:87], 0\ns_cbranch_scc1 {EXIT}")
    # from here on the node is walked: its stop rank is taken, the extras three stops ahead are asked for
    e("s_add_i32 %[nstop], %[nstop], 1")
    g.ext_issue()
    g.fold(P(4), R5F, R5F_I)
    g.fold(P(5), F3F, F3F_I)
    # B: the far gene ends (either order), then the near ones pair by pair
    e(f"""
        v_cmp_le_f64_e32 vcc, 0, {v2(R5F)}
        v_cndmask_b32_e32 v{B[0]}, 0, v{R5F[0]}, vcc
        v_cndmask_b32_e32 v{B[1]}, 0, v{R5F[1]}, vcc
        v_cndmask_b32_e32 v{BTB}, -1, v{R5F_I}, vcc
        v_mov_b32_e32 v{BOV}, -1""")
    g.take_lex(F3F, F3F_I)
    g.near_loop(P(7), P(8), lex=True)
    g.near_loop(P(6), None, lex=True, plain_negc=True)
    # the reverse stop whose ORF covers this one, per frame of an overlapping start: an operon
    for q in range(3):
        skip = g.label("op")
        e(f"""
            s_bitcmp1_b32 {P(15)}, {q}
            s_cbranch_scc0 {skip}
            ds_read_b64 {v2(TA)}, v{L8} offset:{L3V + 64 * q}
            v_and_b32_e32 v{VC}, {1 << q}, v{VM}
            v_cmp_ne_u32_e64 s[72:73], 0, v{VC}
            v_mov_b32_e32 v{VJ}, {P(9 + q)}
            s_waitcnt lgkmcnt(0)
            v_add_f64 {v2(TA)}, {v2(TA)}, {v2(X[q])}""")
        g.take_lex(TA, VJ, "s[72:73]")
        e(f"{skip}:")
    # forward stops that overlap the 3' end of the gene of an overlapping start: the (stop, start) pairs of the frame's list
    for q in range(3):
        skip, loop = g.label("ls"), g.label("ll")
        e(f"""
            s_bfe_u32 s76, {P(15)}, {hex((6 << 16) | (6 + 6 * q))}
            s_cmp_eq_u32 s76, 0
            s_cbranch_scc1 {skip}
            v_and_b32_e32 v{VC}, {1 << q}, v{VM}
            v_cmp_ne_u32_e32 vcc, 0, v{VC}
            v_cmp_eq_u32_e64 s[66:67], {P(12 + q)}, v{N3S[q]}
            s_and_b64 vcc, vcc, s[66:67]
            v_cmp_lt_f64_e64 s[66:67], 0, {v2(X[q])}
            s_and_b64 s[78:79], vcc, s[66:67]
            s_cmp_eq_u64 s[78:79], 0
            s_cbranch_scc1 {skip}
            {loop}:
            s_ff1_i32_b32 s62, s76
            s_bitset0_b32 s76, s62
            s_add_i32 s63, s62, {8 * q}
            v_readlane_b32 s64, v{CNDX}, s63
            v_readlane_b32 s65, v{CIDX}, s63
            s_lshl_b32 s66, s62, 7
            v_add_u32_e32 v{VA}, s66, v{L16}
            ds_read_b128 {v4(RD[0])}, v{VA} offset:{CAND + q * DPC_CAND * 128}
            s_add_i32 s66, s64, 5
            s_sub_i32 s66, s66, {P(12 + q)}
            s_add_i32 s67, s66, s64
            s_add_i32 s67, s67, 2
            s_sub_i32 s68, {P(12 + q)}, 2
            s_sub_i32 s68, s68, s66
            v_mov_b32_e32 v{VJ}, s65
            s_waitcnt lgkmcnt(0)
            v_add_f64 {v2(TA)}, {v2(RD)}, {v2(X[q])}
            v_cmp_lt_i32_e32 vcc, s67, v{N3N[q]}
            v_cmp_gt_i32_e64 s[70:71], s68, v{RD[0] + 2}
            s_and_b64 s[70:71], s[70:71], vcc
            s_and_b64 s[70:71], s[70:71], s[78:79]""")
        g.lex_cond(TA, VJ, "s[70:71]")
        e(f"""
            v_cndmask_b32_e32 v{B[0]}, v{B[0]}, v{TA[0]}, vcc
            v_cndmask_b32_e32 v{B[1]}, v{B[1]}, v{TA[1]}, vcc
            v_cndmask_b32_e32 v{BTB}, v{BTB}, v{VJ}, vcc
            v_cndmask_b32_e64 v{BOV}, v{BOV}, {q}, vcc
            s_cmp_lg_u32 s76, 0
            s_cbranch_scc1 {loop}
            {skip}:""")
    # dpc_finish_r3: it becomes the last reverse stop of its frame; the frame's list starts over with the forward stops up to four
    # bases before it
    e(f"""
        s_lshl_b32 s62, {FRAME}, 6
        v_add_u32_e32 v{VC}, s62, v{L8}
        v_add_u32_e32 v{VJ}, 1, v{BOV}
        v_lshlrev_b32_e32 v{VJ}, 28, v{VJ}
        v_or_b32_e32 v{VJ}, v{VJ}, v{BTB}
        v_cmp_gt_i32_e32 vcc, 0, v{BTB}
        v_cndmask_b32_e64 v{SVTAG}, v{VJ}, -1, vcc
        v_mov_b32_e32 v{SV[0]}, v{B[0]}
        v_mov_b32_e32 v{SV[1]}, v{B[1]}
        v_mov_b32_e32 v{SVTBN}, -1""")
    g.writers()
    e(f"ds_write_b64 v{VC}, {v2(B)} offset:{L3V}")
    g.hist_write(SV[0])
    g.everyone()
    rdone, rloop, rfull = g.label("rd"), g.label("rp"), g.label("rf")
    e(f"""
        s_lshr_b32 s76, {P(0)}, 16
        s_cmp_eq_u32 s76, 0
        s_cbranch_scc1 {rdone}
        s_mov_b32 s77, 0
        s_lshl_b32 s78, {FRAME}, 3
        s_mul_i32 s79, {FRAME}, {DPC_CAND * 128}
        {rloop}:""")
    g.bit_desc("s76")
    e(f"""
        s_cmp_ge_u32 s77, {DPC_CAND}
        s_cbranch_scc1 {rfull}
        s_and_b32 s65, s64, 31
        s_lshl_b32 s65, s65, 7
        v_add_u32_e32 v{VA}, s65, v{L16}
        ds_read_b128 {v4(RD[0])}, v{VA} offset:{HIST}
        s_lshl_b32 s66, s64, 6
        s_add_u32 s70, s84, s66
        s_addc_u32 s71, s85, 0
        s_load_dword s66, s[70:71], 0x4
        s_add_i32 s67, s78, s77
        v_cmp_eq_u32_e32 vcc, s67, v{TID}
        v_mov_b32_e32 v{VJ}, s64
        v_cndmask_b32_e32 v{CIDX}, v{CIDX}, v{VJ}, vcc
        s_lshl_b32 s68, s77, 7
        s_add_i32 s68, s68, s79
        v_add_u32_e32 v{VA}, s68, v{L16}
        s_waitcnt lgkmcnt(0)
        v_mov_b32_e32 v{VJ}, s66
        v_cndmask_b32_e32 v{CNDX}, v{CNDX}, v{VJ}, vcc""")
    g.writers()
    e(f"ds_write_b128 v{VA}, {v4(RD[0])} offset:{CAND}")
    g.everyone()
    e(f"""
        {rfull}:
        s_add_i32 s77, s77, 1
        s_cmp_lg_u32 s76, 0
        s_cbranch_scc1 {rloop}
        {rdone}:
        s_branch {NEXT}""")

    # ================================================================ forward start
    e(f"{LF5}:")
    g.smem_free = False
    g.window_check(R5A_I, EXIT)
    g.cs_read()
    e(f"""
        s_lshl_b32 s62, {FRAME}, 7
        v_add_u32_e32 v{VC}, s62, v{L16}
        ds_read_b128 {v4(CARRYR[0])}, v{VC} offset:{CARRY}""")
    g.fold(P(4), R5F, R5F_I)
    g.fold(P(5), F3F, F3F_I)
    e(f"""
        v_cmp_le_f64_e32 vcc, 0, {v2(F3F)}
        v_cndmask_b32_e32 v{B[0]}, 0, v{F3F[0]}, vcc
        v_cndmask_b32_e32 v{B[1]}, 0, v{F3F[1]}, vcc
        v_cndmask_b32_e32 v{BTB}, -1, v{F3F_I}, vcc""")
    g.near_loop(P(6), P(7), lex=False)
    g.take_lex(R5A, R5A_I)
    # dpc_finish_f5: it offers score + cs to the stop of its ORF (a later node wins a tie)
    g.wait_lds()
    e(f"""
        v_add_f64 {v2(TA)}, {v2(B)}, {v2(CSR)}
        v_mov_b32_e32 v{VJ}, {P(1)}
        v_mov_b32_e32 v{BOV}, -1
        v_cmp_ge_f64_e32 vcc, {v2(TA)}, v[{CARRYR[0]}:{CARRYR[0] + 1}]
        v_cndmask_b32_e32 v{CARRYR[0]}, v{CARRYR[0]}, v{TA[0]}, vcc
        v_cndmask_b32_e32 v{CARRYR[0] + 1}, v{CARRYR[0] + 1}, v{TA[1]}, vcc
        v_cndmask_b32_e32 v{CARRYR[0] + 2}, v{CARRYR[0] + 2}, v{VI}, vcc
        v_cndmask_b32_e32 v{CARRYR[0] + 3}, v{CARRYR[0] + 3}, v{VJ}, vcc""")
    g.writers()
    e(f"ds_write_b128 v{VC}, {v4(CARRYR[0])} offset:{CARRY}")
    g.hist_write(B[0])
    g.everyone()
    e(f"s_branch {NEXT}")
    g.smem_free = True

    # ================================================================ reverse start
    e(f"{LR5}:")
    g.cs_read()
    noown, fin, loop = g.label("no"), g.label("rf"), g.label("rc")
    e(f"""
        v_mov_b32_e32 v{B[0]}, 0
        v_mov_b32_e32 v{B[1]}, 0
        v_mov_b32_e32 v{BTB}, -1
        s_bitcmp1_b32 {P(0)}, 9
        s_cbranch_scc0 {noown}
        s_lshl_b32 s62, {FRAME}, 6
        v_add_u32_e32 v{VC}, s62, v{L8}
        ds_read_b64 {v2(TA)}, v{VC} offset:{L3V}
        v_mov_b32_e32 v{VJ}, {P(4)}
        ds_read_b32 v{VDUMMY}, v{L16} offset:{T2}
        s_waitcnt lgkmcnt(1)
        v_add_f64 {v2(TA)}, {v2(TA)}, {v2(CSR)}
        v_cmp_le_f64_e32 vcc, 0, {v2(TA)}
        v_cndmask_b32_e32 v{B[0]}, 0, v{TA[0]}, vcc
        v_cndmask_b32_e32 v{B[1]}, 0, v{TA[1]}, vcc
        v_cndmask_b32_e32 v{BTB}, -1, v{VJ}, vcc
        {noown}:
        s_cmp_eq_u32 {P(5)}, 0
        s_cbranch_scc1 {fin}
        ds_read_b32 v{VDUMMY}, v{L16} offset:{T2}
        s_waitcnt lgkmcnt(1)
        v_add_f64 {v2(TB_)}, {v2(CSR)}, {v2(NEGC)}
        s_lshl_b32 s76, {FRAME}, 3
        s_mul_i32 s77, {FRAME}, {DPC_CAND * 128}
        s_lshl_b32 s78, {P(2)}, 1
        s_sub_i32 s78, s78, 3
        {loop}:
        s_ff1_i32_b32 s62, {P(5)}
        s_bitset0_b32 {P(5)}, s62
        s_add_i32 s63, s76, s62
        v_readlane_b32 s64, v{CNDX}, s63
        v_readlane_b32 s65, v{CIDX}, s63
        s_lshl_b32 s66, s62, 7
        s_add_i32 s66, s66, s77
        v_add_u32_e32 v{VA}, s66, v{L16}
        ds_read_b128 {v4(RD[0])}, v{VA} offset:{CAND}
        s_sub_i32 s67, s78, s64
        v_mov_b32_e32 v{VJ}, s65
        ds_read_b32 v{VDUMMY}, v{L16} offset:{T2}
        s_waitcnt lgkmcnt(1)
        v_cmp_gt_i32_e64 s[72:73], s67, v{RD[0] + 2}
        v_add_f64 {v2(TA)}, {v2(RD)}, {v2(TB_)}""")
    g.take_asc(TA, VJ, "s[72:73]")
    e(f"""
        s_cmp_lg_u32 {P(5)}, 0
        s_cbranch_scc1 {loop}
        {fin}:
        v_cmp_ne_u32_e64 s[74:75], -1, v{BTB}
        v_cndmask_b32_e64 v{SV[0]}, v{NI[0]}, v{B[0]}, s[74:75]
        v_cndmask_b32_e64 v{SV[1]}, v{NI[1]}, v{B[1]}, s[74:75]
        v_mov_b32_e32 v{SVTBN}, -1
        v_mov_b32_e32 v{SVTAG}, v{BTB}""")
    g.note_end()
    e(f"""
        v_add_f64 {v2(TA)}, {v2(B)}, {v2(NEGC)}
        v_cmp_ge_f64_e32 vcc, {v2(TA)}, {v2(R5A)}
        s_and_b64 vcc, vcc, s[74:75]
        v_cndmask_b32_e32 v{R5A[0]}, v{R5A[0]}, v{TA[0]}, vcc
        v_cndmask_b32_e32 v{R5A[1]}, v{R5A[1]}, v{TA[1]}, vcc
        v_cndmask_b32_e32 v{R5A_I}, v{R5A_I}, v{VI}, vcc""")
    g.writers()
    g.hist_write(SV[0])
    g.everyone()
    e(f"s_branch {NEXT}")

    # ================================================================ forward stop
    e(f"{LF3}:")
    g.ext_read(False)
    e(f"""
        s_lshl_b32 s62, {FRAME}, 7
        v_add_u32_e32 v{VC}, s62, v{L16}
        ds_read_b128 {v4(CARRYR[0])}, v{VC} offset:{CARRY}
        s_add_i32 %[nstop], %[nstop], 1""")
    g.ext_issue()
    C0 = CARRYR[0]
    e(f"""
        ds_read_b32 v{VDUMMY}, v{L16} offset:{T2}
        s_waitcnt lgkmcnt(1)
        v_cmp_le_i32_e32 vcc, 0, v{C0 + 2}
        v_cmp_le_f64_e64 s[66:67], 0, v[{C0}:{C0 + 1}]
        s_and_b64 s[74:75], vcc, s[66:67]
        v_cndmask_b32_e64 v{B[0]}, 0, v{C0}, s[74:75]
        v_cndmask_b32_e64 v{B[1]}, 0, v{C0 + 1}, s[74:75]
        v_cndmask_b32_e64 v{BTB}, -1, v{C0 + 2}, s[74:75]
        v_cndmask_b32_e64 v{SVTBN}, -1, v{C0 + 3}, s[74:75]
        v_cndmask_b32_e64 v{SV[0]}, v{NI[0]}, v{C0}, s[74:75]
        v_cndmask_b32_e64 v{SV[1]}, v{NI[1]}, v{C0 + 1}, s[74:75]
        v_mov_b32_e32 v{SVTAG}, v{BTB}""")
    g.note_end()
    # the running maximum of its own frame starts over
    e(f"""
        v_mov_b32_e32 v{C0}, v{NI[0]}
        v_mov_b32_e32 v{C0 + 1}, v{NI[1]}
        v_mov_b32_e32 v{C0 + 2}, -1
        v_mov_b32_e32 v{C0 + 3}, -1""")
    g.writers()
    e(f"ds_write_b128 v{VC}, {v4(C0)} offset:{CARRY}")
    g.everyone()
    # when reached it offers score + x to the frames whose next stop's ORF holds it (operon partners)
    for q in range(3):
        skip = g.label("of")
        e(f"""
            s_bitcmp1_b32 {P(0)}, {4 + q}
            s_cbranch_scc0 {skip}
            ds_read_b128 {v4(RD[0])}, v{L16} offset:{CARRY + 128 * q}
            v_and_b32_e32 v{VA}, {1 << q}, v{VM}
            v_cmp_ne_u32_e32 vcc, 0, v{VA}
            s_and_b64 s[72:73], vcc, s[74:75]
            v_add_f64 {v2(TA)}, {v2(B)}, {v2(X[q])}
            v_mov_b32_e32 v{VJ}, {P(1)}
            ds_read_b32 v{VDUMMY}, v{L16} offset:{T2}
            s_waitcnt lgkmcnt(1)
            v_cmp_ge_f64_e32 vcc, {v2(TA)}, {v2(RD)}
            s_and_b64 vcc, vcc, s[72:73]
            v_cndmask_b32_e32 v{RD[0]}, v{RD[0]}, v{TA[0]}, vcc
            v_cndmask_b32_e32 v{RD[0] + 1}, v{RD[0] + 1}, v{TA[1]}, vcc
            v_cndmask_b32_e32 v{RD[0] + 2}, v{RD[0] + 2}, v{VI}, vcc
            v_cndmask_b32_e32 v{RD[0] + 3}, v{RD[0] + 3}, v{VJ}, vcc""")
        g.writers()
        e(f"ds_write_b128 v{L16}, {v4(RD[0])} offset:{CARRY + 128 * q}")
        g.everyone()
        e(f"{skip}:")
    # it enters the candidate lists of the reverse stops whose genes' 3' ends it can overlap
    for q in range(3):
        skip = g.label("pu")
        e(f"""
            s_bitcmp1_b32 {P(4)}, {q}
            s_cbranch_scc0 {skip}
            s_bfe_u32 s62, {P(4)}, {hex((3 << 16) | (3 + 3 * q))}
            s_add_i32 s63, s62, {8 * q}
            v_cmp_eq_u32_e32 vcc, s63, v{TID}
            v_cndmask_b32_e32 v{CIDX}, v{CIDX}, v{VI}, vcc
            v_mov_b32_e32 v{VJ}, {P(1)}
            v_cndmask_b32_e32 v{CNDX}, v{CNDX}, v{VJ}, vcc
            s_lshl_b32 s62, s62, 7
            v_add_u32_e32 v{VA}, s62, v{L16}""")
        g.writers()
        e(f"ds_write_b128 v{VA}, {v4(SV[0])} offset:{CAND + q * DPC_CAND * 128}")
        g.everyone()
        e(f"{skip}:")
    g.writers()
    g.hist_write(SV[0])
    g.everyone()

    # ---- next node
    e(f"""
        {NEXT}:
        s_add_i32 %[i], %[i], 1
        s_cmp_ge_i32 %[i], %[end]
        s_cbranch_scc1 {EXIT}
        s_waitcnt lgkmcnt(0)
        s_mov_b64 s[36:37], s[52:53]
        s_mov_b64 s[38:39], s[54:55]
        s_mov_b64 s[40:41], s[56:57]
        s_mov_b64 s[42:43], s[58:59]
        s_branch {NODE}
        {EXIT}:
        s_waitcnt lgkmcnt(0)""")
    for r, name in ins:
        e(f"v_mov_b32_e32 %[{name}], v{r}")

    return g.lines


def main():
    lines = generate()
    print("// GENERATED by tools/gen_dpc_walk.py -- do not edit; see that script for what the block does and its register tables.")
    print("asm volatile(")
    for ln in lines:
        print('    "%s\\n\\t"' % ln)
    outs = ["i", "nstop"]
    vio = ["r5a_lo", "r5a_hi", "r5a_i", "r5f_lo", "r5f_hi", "r5f_i", "f3f_lo", "f3f_hi", "f3f_i", "end_lo", "end_hi", "end_i", "end_tb", "cidx", "cndx"]
    print("    : " + ", ".join(['[%s] "+s"(w_%s)' % (n, n) for n in outs] + ['[%s] "+v"(w_%s)' % (n, n) for n in vio]))
    vin = ["tid", "negc_lo", "negc_hi", "stwt_lo", "stwt_hi", "extp_lo", "extp_hi"]
    sin = ["end", "prog_lo", "prog_hi", "base"]
    print("    : " + ", ".join(['[%s] "v"(w_%s)' % (n, n) for n in vin] + ['[%s] "s"(w_%s)' % (n, n) for n in sin]))
    clob = ['"memory"', '"vcc"'] + ['"s%d"' % k for k in range(36, 90)] + ['"v%d"' % k for k in range(27, 96)]
    print("    : " + ", ".join(clob) + ");")


if __name__ == "__main__":
    main()
