"""ctypes binding of the CPU oracle (``libprodigal_oracle.so``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` -- never by ``pyrodigal_amd``.
"""
import ctypes
import gzip
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libprodigal_oracle.so")

TRAINING_SIZE = 558392

# field order / alignment of `po_node` in prodigal_oracle.h
NODE_DTYPE = np.dtype(
    [
        ("cscore", "f8"), ("uscore", "f8"), ("tscore", "f8"), ("rscore", "f8"),
        ("sscore", "f8"), ("score", "f8"), ("gc_score", "f8", (3,)), ("mot_score", "f8"),
        ("gc_cont", "f4"), ("star_ptr", "i4", (3,)), ("traceb", "i4"), ("tracef", "i4"),
        ("ndx", "i4"), ("stop_val", "i4"), ("mot_ndx", "i4"), ("ov_mark", "i1"),
        ("strand", "i1"), ("rbs", "u1", (2,)), ("edge", "u1"), ("elim", "u1"),
        ("gc_bias", "u1"), ("type", "u1"), ("mot_len", "u1"), ("mot_spacer", "u1"),
        ("mot_spacendx", "u1"), ("_pad", "u1"),
    ],
    align=True,
)
GENE_DTYPE = np.dtype([("begin", "i4"), ("end", "i4"), ("start_ndx", "i4"), ("stop_ndx", "i4")])


class Params(ctypes.Structure):
    _fields_ = [("closed", ctypes.c_int32), ("min_gene", ctypes.c_int32),
                ("min_edge_gene", ctypes.c_int32), ("max_overlap", ctypes.c_int32)]

    def __init__(self, closed=False, min_gene=90, min_edge_gene=60, max_overlap=60):
        super().__init__(int(closed), min_gene, min_edge_gene, max_overlap)


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "prodigal_oracle.c"))
    ):
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        vp, i32, f64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
        L.po_new.restype = vp
        L.po_new.argtypes = [ctypes.c_char_p, ctypes.c_int64, i32, i32]
        L.po_free.argtypes = [vp]
        L.po_record_gc_bias.restype = None; L.po_record_gc_bias.argtypes = [vp, vp]
        L.po_masks.restype = i32
        L.po_masks.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), i32]
        for name in ("po_slen", "po_unknown", "po_num_nodes", "po_num_genes", "po_last_ipath"):
            getattr(L, name).restype = i32
            getattr(L, name).argtypes = [vp]
        L.po_gc.restype = f64; L.po_gc.argtypes = [vp]
        L.po_last_path_score.restype = f64; L.po_last_path_score.argtypes = [vp]
        L.po_digits.restype = vp; L.po_digits.argtypes = [vp]
        L.po_nodes.restype = vp; L.po_nodes.argtypes = [vp]
        L.po_genes.restype = vp; L.po_genes.argtypes = [vp]
        L.po_node_size.restype = i32
        L.po_extract.restype = i32; L.po_extract.argtypes = [vp, i32, vp]
        L.po_sort.argtypes = [vp]
        L.po_reset_scores.argtypes = [vp]
        L.po_score_nodes.argtypes = [vp, vp, i32, i32]
        L.po_overlapping_starts.argtypes = [vp, vp, i32, i32]
        L.po_dprog.restype = i32; L.po_dprog.argtypes = [vp, vp, i32, i32]
        L.po_dprog_raw.restype = None; L.po_dprog_raw.argtypes = [vp, vp, i32]
        L.po_find_max_index.restype = i32; L.po_find_max_index.argtypes = [vp]
        L.po_eliminate_bad_genes.argtypes = [vp, i32, vp]
        L.po_extract_genes.restype = i32; L.po_extract_genes.argtypes = [vp, i32]
        L.po_tweak_final_starts.argtypes = [vp, vp, i32]
        L.po_find_genes_single.restype = i32; L.po_find_genes_single.argtypes = [vp, vp, vp]
        L.po_find_genes_meta.restype = i32; L.po_find_genes_meta.argtypes = [vp, vp, i32, vp]
        L.po_find_genes_meta_pool.restype = ctypes.c_int64; L.po_find_genes_meta_pool.argtypes = [vp, vp, i32, vp, i32, vp, i32]
        L.po_find_genes_meta_pool_pinned.restype = ctypes.c_int64
        L.po_find_genes_meta_pool_pinned.argtypes = [vp, vp, i32, vp, i32, vp, i32, i32, vp, i32, vp]
        L.po_train.restype = i32; L.po_train.argtypes = [vp, vp, vp, i32, f64, i32]
        L.po_train_upto.restype = i32; L.po_train_upto.argtypes = [vp, vp, vp, i32, f64, i32, i32]
        assert L.po_node_size() == NODE_DTYPE.itemsize, (L.po_node_size(), NODE_DTYPE.itemsize)
        _lib = L
    return _lib


class Training:
    """A 558 392-byte ``struct _training`` blob (reference byte layout)."""

    def __init__(self, raw=None):
        self.buf = np.zeros(TRAINING_SIZE, dtype=np.uint8)
        if raw is not None:
            assert len(raw) == TRAINING_SIZE, len(raw)
            self.buf[:] = np.frombuffer(raw, dtype=np.uint8)

    @classmethod
    def load(cls, path):
        opener = gzip.open if str(path).endswith(".gz") else open
        with opener(path, "rb") as f:
            return cls(f.read())

    @property
    def ptr(self):
        return self.buf.ctypes.data

    def tobytes(self):
        return self.buf.tobytes()

    def _f64(self, off, n=1):
        return self.buf[off:off + 8 * n].view(np.float64)

    def _i32(self, off):
        return self.buf[off:off + 4].view(np.int32)

    gc = property(lambda s: float(s._f64(0)[0]))
    trans_table = property(lambda s: int(s._i32(8)[0]))
    st_wt = property(lambda s: float(s._f64(16)[0]))
    uses_sd = property(lambda s: int(s._i32(72)[0]))
    no_mot = property(lambda s: float(s._f64(525616)[0]))
    bias = property(lambda s: s._f64(24, 3))
    type_wt = property(lambda s: s._f64(48, 3))
    rbs_wt = property(lambda s: s._f64(80, 28))

    def set_gc(self, v):
        self._f64(0)[0] = v

    def set_trans_table(self, v):
        self._i32(8)[0] = v

    def copy(self):
        return Training(self.tobytes())


class Oracle:
    """One sequence worth of oracle state (Sequence + Nodes + Genes)."""

    def __init__(self, seq, mask=False, mask_size=50):
        if isinstance(seq, str):
            seq = seq.encode("ascii")
        self._seq = bytes(seq)
        self.L = lib()
        self.h = self.L.po_new(self._seq, len(self._seq), int(mask), mask_size)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.po_free(self.h)
            self.h = None

    slen = property(lambda s: s.L.po_slen(s.h))
    gc = property(lambda s: s.L.po_gc(s.h))
    num_nodes = property(lambda s: s.L.po_num_nodes(s.h))
    num_genes = property(lambda s: s.L.po_num_genes(s.h))
    path_score = property(lambda s: s.L.po_last_path_score(s.h))
    ipath = property(lambda s: s.L.po_last_ipath(s.h))

    def masks(self):
        """Masked regions as an (k, 2) array of [begin, end) (ref: lib.pyx:699-713)."""
        k = self.L.po_masks(self.h, None, 0)
        out = np.zeros((k, 2), np.int32)
        if k:
            self.L.po_masks(self.h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), k)
        return out

    def digits(self):
        n = self.slen
        return np.ctypeslib.as_array(ctypes.cast(self.L.po_digits(self.h), ctypes.POINTER(ctypes.c_uint8)), (n,)).copy() if n else np.zeros(0, np.uint8)

    def nodes(self, copy=True):
        n = self.num_nodes
        if n == 0:
            return np.zeros(0, NODE_DTYPE)
        raw = ctypes.cast(self.L.po_nodes(self.h), ctypes.POINTER(ctypes.c_uint8))
        a = np.ctypeslib.as_array(raw, (n * NODE_DTYPE.itemsize,)).view(NODE_DTYPE)
        return a.copy() if copy else a

    def genes(self):
        n = self.num_genes
        if n == 0:
            return np.zeros(0, GENE_DTYPE)
        raw = ctypes.cast(self.L.po_genes(self.h), ctypes.POINTER(ctypes.c_uint8))
        return np.ctypeslib.as_array(raw, (n * GENE_DTYPE.itemsize,)).view(GENE_DTYPE).copy()

    # stage-level
    def extract(self, tt=11, params=None):
        p = params or Params()
        return self.L.po_extract(self.h, tt, ctypes.addressof(p))

    def sort(self):
        self.L.po_sort(self.h)

    def reset_scores(self):
        self.L.po_reset_scores(self.h)

    def score_nodes(self, tinf, closed=False, is_meta=False):
        self.L.po_score_nodes(self.h, tinf.ptr, int(closed), int(is_meta))

    def overlapping_starts(self, tinf, flag=1, max_overlap=60):
        self.L.po_overlapping_starts(self.h, tinf.ptr, flag, max_overlap)

    def dprog(self, tinf, final=True):
        return self.L.po_dprog(self.h, tinf.ptr, int(final), 1)

    def record_gc_bias(self, tinf):
        """Training: fill gc_score / gc_bias of the nodes and ``tinf.bias`` (Prodigal record_gc_bias)."""
        self.L.po_record_gc_bias(self.h, tinf.ptr)

    def dprog_raw(self, tinf, final=True):
        """Connection loop only (no traceback fix-ups): nodes keep the raw score/traceb/ov_mark."""
        self.L.po_dprog_raw(self.h, tinf.ptr, int(final))

    def find_max_index(self):
        return self.L.po_find_max_index(self.h)

    def eliminate_bad_genes(self, ipath, tinf):
        self.L.po_eliminate_bad_genes(self.h, ipath, tinf.ptr)

    def extract_genes(self, ipath):
        return self.L.po_extract_genes(self.h, ipath)

    def tweak_final_starts(self, tinf, max_overlap=60):
        self.L.po_tweak_final_starts(self.h, tinf.ptr, max_overlap)

    # drivers
    def find_genes_single(self, tinf, params=None):
        p = params or Params()
        return self.L.po_find_genes_single(self.h, tinf.ptr, ctypes.addressof(p))

    def find_genes_meta(self, bins, params=None):
        p = params or Params()
        arr = (ctypes.c_void_p * len(bins))(*[b.ptr for b in bins])
        return self.L.po_find_genes_meta(self.h, arr, len(bins), ctypes.addressof(p))

    def train(self, params=None, force_nonsd=False, start_weight=4.35, tt=11, upto=0):
        """Single-genome training; ``upto`` = 1 / 2 / 3 stops after the GC bias / hexamer statistics / SD start training."""
        p = params or Params()
        t = Training()
        self.L.po_train_upto(self.h, t.ptr, ctypes.addressof(p), int(force_nonsd), start_weight, tt, upto)
        return t


# ----- gene description helpers (restating Gene properties, ref: lib.pyx:2644-2760, 209-228) -----

RBS_MOTIF = [
    None, "GGA/GAG/AGG", "3Base/5BMM", "4Base/6BMM", "AGxAG", "AGxAG", "GGA/GAG/AGG", "GGxGG", "GGxGG",
    "AGxAG", "AGGAG(G)/GGAGG", "AGGA/GGAG/GAGG", "AGGA/GGAG/GAGG", "GGA/GAG/AGG", "GGxGG", "AGGA",
    "GGAG/GAGG", "AGxAGG/AGGxGG", "AGxAGG/AGGxGG", "AGxAGG/AGGxGG", "AGGAG/GGAGG", "AGGAG", "AGGAG",
    "GGAGG", "GGAGG", "AGGAGG", "AGGAGG", "AGGAGG",
]
RBS_SPACER = [
    None, "3-4bp", "13-15bp", "13-15bp", "11-12bp", "3-4bp", "11-12bp", "11-12bp", "3-4bp", "5-10bp",
    "13-15bp", "3-4bp", "11-12bp", "5-10bp", "5-10bp", "5-10bp", "5-10bp", "11-12bp", "3-4bp", "5-10bp",
    "11-12bp", "3-4bp", "5-10bp", "3-4bp", "5-10bp", "11-12bp", "3-4bp", "5-10bp",
]
NODE_TYPE = ["ATG", "GTG", "TTG", "Edge"]


def mer_text(length, ndx):
    """Prodigal sequence.c mer_text: 2-bit groups, LSB first, letters AGCT."""
    if length == 0:
        return "None"
    return "".join("AGCT"[(ndx >> (2 * i)) & 3] for i in range(length))


def gene_records(genes, nodes, tinf):
    """(begin, end, strand, partial, start_type, rbs_motif, rbs_spacer, gc_cont %.3f) per gene."""
    out = []
    rbs_wt, st_wt = tinf.rbs_wt, tinf.st_wt
    for g in genes:
        s, e = nodes[g["start_ndx"]], nodes[g["stop_ndx"]]
        strand = int(s["strand"])
        pb, pe = (s["edge"], e["edge"]) if strand == 1 else (e["edge"], s["edge"])
        r1, r2 = rbs_wt[s["rbs"][0]] * st_wt, rbs_wt[s["rbs"][1]] * st_wt
        if tinf.uses_sd:
            k = s["rbs"][0] if r1 > r2 else s["rbs"][1]
            motif, spacer = RBS_MOTIF[k], RBS_SPACER[k]
        elif tinf.no_mot > -0.5 and r1 > r2 and r1 > s["mot_score"] * st_wt:
            motif, spacer = RBS_MOTIF[s["rbs"][0]], RBS_SPACER[s["rbs"][0]]
        elif tinf.no_mot > -0.5 and r2 >= r1 and r2 > s["mot_score"] * st_wt:
            motif, spacer = RBS_MOTIF[s["rbs"][1]], RBS_SPACER[s["rbs"][1]]
        elif s["mot_len"] == 0:
            motif, spacer = None, None
        else:
            motif, spacer = mer_text(int(s["mot_len"]), int(s["mot_ndx"])), "%dbp" % s["mot_spacer"]
        out.append((int(g["begin"]), int(g["end"]), strand, "%d%d" % (pb, pe),
                    NODE_TYPE[3 if s["edge"] else int(s["type"])], str(motif), str(spacer),
                    "%.3f" % s["gc_cont"]))
    return out


def find_genes_meta_pool(seqs, bins, threads, params=None):
    """Meta-mode gene finding of many sequences on a pool of C threads that share the models (pyrodigal's thread-pool model, ref:
    cli.py:289-302, without the interpreter lock): returns the number of genes found.  bench.py's all-core CPU baseline."""
    L = lib()
    p = params or Params()
    n = len(seqs)
    ptrs = (ctypes.c_char_p * max(n, 1))(*seqs)
    lens = (ctypes.c_int64 * max(n, 1))(*[len(s) for s in seqs])
    arr = (ctypes.c_void_p * len(bins))(*[b.ptr for b in bins])
    got = L.po_find_genes_meta_pool(ptrs, lens, n, arr, len(bins), ctypes.addressof(p), int(threads))
    if got < 0:
        raise RuntimeError("po_find_genes_meta_pool failed")
    return int(got)


def find_genes_meta_pool_pinned(seqs, bins, threads, total=None, cpus=None, params=None):
    """`find_genes_meta_pool` making `total` calls (the list gone over as often as that takes), thread t pinned to logical CPU
    cpus[t % len(cpus)] when a list is given: returns (genes found, CPU seconds the threads used)."""
    L = lib()
    p = params or Params()
    n = len(seqs)
    ptrs = (ctypes.c_char_p * max(n, 1))(*seqs)
    lens = (ctypes.c_int64 * max(n, 1))(*[len(s) for s in seqs])
    arr = (ctypes.c_void_p * len(bins))(*[b.ptr for b in bins])
    cl = (ctypes.c_int * max(len(cpus or []), 1))(*(cpus or [0]))
    cpu_s = ctypes.c_double(0.0)
    got = L.po_find_genes_meta_pool_pinned(ptrs, lens, n, arr, len(bins), ctypes.addressof(p), int(threads), int(total or n),
                                           ctypes.addressof(cl) if cpus else None, len(cpus or []), ctypes.addressof(cpu_s))
    if got < 0:
        raise RuntimeError("po_find_genes_meta_pool_pinned failed")
    return int(got), float(cpu_s.value)
