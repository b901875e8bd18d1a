"""ctypes binding of ``libpyrodigal_amd.so`` (the C-ABI declared in ``include/pyrodigal_amd.h``).

There is no CPU implementation behind this module: if the shared library is missing, or no
gfx950 device is visible, calls raise instead of silently computing elsewhere.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpyrodigal_amd.so")

PGA_OK, PGA_EINVAL, PGA_ENOMEM, PGA_EDEVICE, PGA_ENODEVICE = 0, -1, -2, -3, -4
TRAINING_SIZE = 558392

EXPORTS = [
    "pga_create", "pga_destroy", "pga_last_error", "pga_device_info", "pga_set_models",
    "pga_score_connections", "pga_score_connections_training", "pga_find_genes_batch", "pga_result_free",
    "pga_batch_create", "pga_batch_free", "pga_find_genes", "pga_nodes_stage",
    "pga_fasta_open", "pga_fasta_next", "pga_fasta_error", "pga_fasta_close", "pga_train", "pga_dp_stats", "pga_dp_timings", "pga_extract_stats", "pga_dp_plan_summary", "pga_dp_start_order", "pga_cs_task_summary",
    "pga_fasta_next_packed", "pga_batch_create_packed", "pga_translate_genes", "pga_fasta_open_callback", "pga_fasta_release_spare", "pga_dp_xcd_order", "pga_release_cached",
]
STAGE_EXTRACT, STAGE_SCORE, STAGE_OVERLAP, STAGE_SEQUENCE = 1, 2, 3, 4


class Params(ctypes.Structure):
    _fields_ = [("closed", ctypes.c_int32), ("min_gene", ctypes.c_int32), ("min_edge_gene", ctypes.c_int32),
                ("max_overlap", ctypes.c_int32), ("meta", ctypes.c_int32), ("want_nodes", ctypes.c_int32),
                ("mask", ctypes.c_int32), ("min_mask", ctypes.c_int32)]


class Gene(ctypes.Structure):
    _fields_ = [("contig", ctypes.c_int32), ("begin", ctypes.c_int32), ("end", ctypes.c_int32),
                ("start_ndx", ctypes.c_int32), ("stop_ndx", ctypes.c_int32), ("strand", ctypes.c_int8),
                ("partial_begin", ctypes.c_uint8), ("partial_end", ctypes.c_uint8), ("start_type", ctypes.c_uint8),
                ("rbs", ctypes.c_uint8 * 2), ("mot_len", ctypes.c_uint8), ("mot_spacer", ctypes.c_uint8),
                ("mot_ndx", ctypes.c_int32), ("gc_cont", ctypes.c_float),
                ("cscore", ctypes.c_double), ("sscore", ctypes.c_double), ("rscore", ctypes.c_double),
                ("uscore", ctypes.c_double), ("tscore", ctypes.c_double), ("mot_score", ctypes.c_double)]


_P = ctypes.POINTER


class Nodes(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32),
                ("ndx", _P(ctypes.c_int32)), ("stop_val", _P(ctypes.c_int32)), ("traceb", _P(ctypes.c_int32)),
                ("tracef", _P(ctypes.c_int32)), ("star_ptr", _P(ctypes.c_int32)),
                ("type", _P(ctypes.c_uint8)), ("edge", _P(ctypes.c_uint8)), ("elim", _P(ctypes.c_uint8)),
                ("rbs", _P(ctypes.c_uint8)), ("strand", _P(ctypes.c_int8)), ("ov_mark", _P(ctypes.c_int8)),
                ("gc_cont", _P(ctypes.c_float)),
                ("cscore", _P(ctypes.c_double)), ("sscore", _P(ctypes.c_double)), ("rscore", _P(ctypes.c_double)),
                ("uscore", _P(ctypes.c_double)), ("tscore", _P(ctypes.c_double)), ("score", _P(ctypes.c_double)),
                ("mot_score", _P(ctypes.c_double)), ("mot_ndx", _P(ctypes.c_int32)),
                ("mot_len", _P(ctypes.c_uint8)), ("mot_spacer", _P(ctypes.c_uint8)), ("mot_spacendx", _P(ctypes.c_uint8))]


class ContigResult(ctypes.Structure):
    _fields_ = [("model", ctypes.c_int32), ("n_nodes", ctypes.c_int32), ("gene_begin", ctypes.c_int64),
                ("n_genes", ctypes.c_int32), ("n_unknown", ctypes.c_int32), ("gc", ctypes.c_double),
                ("score", ctypes.c_double)]


class Result(ctypes.Structure):
    _fields_ = [("n_contigs", ctypes.c_int32), ("n_genes", ctypes.c_int64), ("contigs", _P(ContigResult)),
                ("genes", _P(Gene)), ("nodes", _P(Nodes)), ("t_total_ms", ctypes.c_double),
                ("t_dp_ms", ctypes.c_double), ("node_passes", ctypes.c_int64), ("n_chains", ctypes.c_int32),
                ("_pad", ctypes.c_int32), ("mask_off", _P(ctypes.c_int32)), ("masks", _P(ctypes.c_int32))]


GENE_DTYPE = np.dtype(Gene)
CONTIG_DTYPE = np.dtype(ContigResult)

_lib = None
FASTA_READ_FN = ctypes.CFUNCTYPE(ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64)


def load():
    """Load the C-ABI library; raises ``RuntimeError`` when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); pyrodigal_amd has no CPU fallback")
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, f64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double
    L.pga_create.restype = ctypes.c_int; L.pga_create.argtypes = [ctypes.c_int, _P(vp)]
    L.pga_destroy.restype = None; L.pga_destroy.argtypes = [vp]
    L.pga_last_error.restype = ctypes.c_char_p; L.pga_last_error.argtypes = [vp]
    L.pga_dp_stats.restype = ctypes.c_int; L.pga_dp_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_int32)]
    L.pga_dp_timings.restype = ctypes.c_int; L.pga_dp_timings.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
    L.pga_extract_stats.restype = ctypes.c_int; L.pga_extract_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_int32)]
    L.pga_dp_plan_summary.restype = ctypes.c_int
    L.pga_dp_plan_summary.argtypes = [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)]
    L.pga_device_info.restype = ctypes.c_int
    L.pga_device_info.argtypes = [vp, ctypes.c_char_p, ctypes.c_int, _P(ctypes.c_int), _P(i64)]
    L.pga_set_models.restype = ctypes.c_int; L.pga_set_models.argtypes = [vp, _P(vp), ctypes.c_int]
    L.pga_score_connections.restype = ctypes.c_int
    L.pga_score_connections.argtypes = [vp, i32] + [vp] * 9 + [f64, ctypes.c_int, vp, vp, vp, _P(i32), _P(f64)]
    L.pga_score_connections_training.restype = ctypes.c_int
    L.pga_score_connections_training.argtypes = [vp, i32] + [vp] * 7 + [f64, vp, vp, vp, _P(i32), _P(f64)]
    L.pga_find_genes_batch.restype = ctypes.c_int
    L.pga_find_genes_batch.argtypes = [vp, i32, _P(ctypes.c_char_p), _P(i64), _P(Params), _P(_P(Result))]
    L.pga_result_free.restype = None; L.pga_result_free.argtypes = [_P(Result)]
    L.pga_batch_create.restype = ctypes.c_int
    L.pga_batch_create.argtypes = [vp, i32, _P(ctypes.c_char_p), _P(i64), _P(vp)]
    L.pga_batch_free.restype = None; L.pga_batch_free.argtypes = [vp]
    L.pga_find_genes.restype = ctypes.c_int; L.pga_find_genes.argtypes = [vp, vp, _P(Params), _P(_P(Result))]
    L.pga_nodes_stage.restype = ctypes.c_int
    L.pga_nodes_stage.argtypes = [vp, vp, _P(Params), ctypes.c_int, ctypes.c_int, _P(_P(Result))]
    L.pga_train.restype = ctypes.c_int
    L.pga_train.argtypes = [vp, vp, _P(Params), ctypes.c_int, f64, ctypes.c_int, ctypes.c_int, vp]
    L.pga_fasta_open.restype = ctypes.c_int; L.pga_fasta_open.argtypes = [ctypes.c_char_p, _P(vp)]
    L.pga_fasta_open_callback.restype = ctypes.c_int; L.pga_fasta_open_callback.argtypes = [FASTA_READ_FN, vp, _P(vp)]
    L.pga_fasta_next.restype = ctypes.c_int
    L.pga_fasta_next.argtypes = [vp, i64, i32, _P(i32), _P(_P(ctypes.c_char_p)), _P(_P(vp)), _P(_P(i64))]
    L.pga_fasta_next_packed.restype = ctypes.c_int
    L.pga_fasta_next_packed.argtypes = [vp, i64, i32, i32, _P(i32), _P(_P(ctypes.c_char_p)), _P(vp), _P(_P(i64)), _P(_P(i64))]
    L.pga_batch_create_packed.restype = ctypes.c_int
    L.pga_batch_create_packed.argtypes = [vp, i32, vp, _P(i64), _P(i64), _P(vp)]
    L.pga_translate_genes.restype = ctypes.c_int
    L.pga_translate_genes.argtypes = [vp, vp, i64, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]
    L.pga_fasta_error.restype = ctypes.c_char_p; L.pga_fasta_error.argtypes = [vp]
    L.pga_fasta_close.restype = None; L.pga_fasta_close.argtypes = [vp]
    L.pga_fasta_release_spare.restype = None; L.pga_fasta_release_spare.argtypes = []
    L.pga_release_cached.restype = None; L.pga_release_cached.argtypes = []
    _lib = L
    return L


def dp_start_order(nodes_per_chain):
    """The order in which the wave-batch connection scorer starts the chains of a launch (host arithmetic, no device needed)."""
    L = load()
    n = len(nodes_per_chain)
    arr = (ctypes.c_int32 * max(n, 1))(*[int(x) for x in nodes_per_chain])
    out = (ctypes.c_int32 * max(n, 1))()
    L.pga_dp_start_order.restype = ctypes.c_int
    rc = L.pga_dp_start_order(ctypes.c_int32(n), arr, out)
    if rc != PGA_OK:
        raise ValueError("pga_dp_start_order failed (code %d)" % rc)
    return list(out[:n])


def dp_xcd_order(nodes_per_chain, key_of_chain, order=None):
    """The start order dealt to the eight XCDs (host arithmetic, no device needed): entry 8 k + x is the k-th chain of XCD x, -1 a filler."""
    L = load()
    n = len(nodes_per_chain)
    order = dp_start_order(nodes_per_chain) if order is None else list(order)
    nk = (max(key_of_chain) + 1) if n else 0
    i32 = ctypes.c_int32
    out = (i32 * max(8 * n, 1))()
    L.pga_dp_xcd_order.restype = ctypes.c_int64
    L.pga_dp_xcd_order.argtypes = [i32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, i32, ctypes.c_void_p, ctypes.c_int64]
    got = L.pga_dp_xcd_order(n, (i32 * max(n, 1))(*order), (i32 * max(n, 1))(*[int(x) for x in nodes_per_chain]),
                             (i32 * max(n, 1))(*[int(x) for x in key_of_chain]), nk, out, 8 * n)
    if got < 0:
        raise ValueError("pga_dp_xcd_order failed (code %d)" % -got)
    return list(out[:got])


def cs_task_summary(nodes_per_contig, first_column, models_per_contig, task_nodes=4096):
    """How the coding-score walks of one translation-table group are cut into tasks (host arithmetic, no device needed)."""
    L = load()
    n = len(nodes_per_contig)
    mk = lambda xs: (ctypes.c_int32 * max(n, 1))(*[int(x) for x in xs])
    out = (ctypes.c_int64 * 5)()
    L.pga_cs_task_summary.restype = ctypes.c_int
    rc = L.pga_cs_task_summary(ctypes.c_int32(n), mk(nodes_per_contig), mk(first_column), mk(models_per_contig), ctypes.c_int32(task_nodes), out)
    if rc != PGA_OK:
        raise ValueError("pga_cs_task_summary: the LDS form does not apply (code %d)" % rc)
    return {"tasks": out[0], "entries": out[1], "largest_task": out[2], "nodes": out[3], "high_columns_first": bool(out[4])}


def dp_plan_summary(nodes_per_chain):
    """How a connection-scoring launch over chains of these node counts would be cut (host arithmetic, no device needed)."""
    L = load()
    n = len(nodes_per_chain)
    arr = (ctypes.c_int32 * max(n, 1))(*[int(x) for x in nodes_per_chain])
    out = (ctypes.c_int64 * 4)()
    rc = L.pga_dp_plan_summary(n, arr, out)
    if rc != PGA_OK:
        raise ValueError("pga_dp_plan_summary failed (code %d)" % rc)
    return {"chains": out[0], "segments": out[1], "max_sub_chain": out[2], "scratch": out[3]}


class PgaError(RuntimeError):
    pass


def _raise(L, ctx, code, what):
    msg = L.pga_last_error(ctx).decode("utf-8", "replace") if ctx else ""
    if code == PGA_EINVAL:
        raise ValueError(f"{what}: {msg}")
    if code == PGA_ENOMEM:
        raise MemoryError(f"{what}: {msg}")
    if code == PGA_ENODEVICE:
        raise PgaError(f"{what}: no gfx950 (MI355X) device visible; pyrodigal_amd has no CPU fallback")
    raise PgaError(f"{what}: {msg} (code {code})")


class Context:
    """Owns a ``pga_ctx`` bound to one GPU."""

    def __init__(self, device=0):
        self.L = load()
        h = ctypes.c_void_p()
        rc = self.L.pga_create(int(device), ctypes.byref(h))
        if rc != PGA_OK:
            _raise(self.L, None, rc, "pga_create")
        self.h = h
        self._models = []

    def close(self):
        if getattr(self, "h", None):
            self.L.pga_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def device_info(self):
        name = ctypes.create_string_buffer(256)
        cus, mem = ctypes.c_int(), ctypes.c_int64()
        rc = self.L.pga_device_info(self.h, name, 256, ctypes.byref(cus), ctypes.byref(mem))
        if rc != PGA_OK:
            _raise(self.L, self.h, rc, "pga_device_info")
        return {"name": name.value.decode(), "cus": cus.value, "hbm_bytes": mem.value}

    def dp_stats(self):
        """How the last connection scoring ran: segmented chains, segments, nodes rejected by each verification round,
        chains walked serially in the end."""
        out = (ctypes.c_int32 * 8)()
        rc = self.L.pga_dp_stats(self.h, out)
        if rc != PGA_OK:
            _raise(self.L, self.h, rc, "pga_dp_stats")
        return {"chains": out[0], "segments": out[1], "rejected": [out[2], out[3], out[4]], "serial": out[5],
                "sched": out[6], "sched_missed": out[7]}

    def dp_timings(self):
        """Device milliseconds of the last find_genes call's connection scoring by part: the scoring launch(es), the topology kernels,
        the step-schedule kernels (HIP events on the context's stream)."""
        out = (ctypes.c_double * 4)()
        rc = self.L.pga_dp_timings(self.h, out)
        if rc != PGA_OK:
            _raise(self.L, self.h, rc, "pga_dp_timings")
        return {"dp_ms": out[0], "topo_ms": out[1], "sched_ms": out[2]}

    def extract_stats(self):
        """How the last node extraction ran: passes (2: a tile overflowed the half-density staging and the batch was extracted again)."""
        out = (ctypes.c_int32 * 2)()
        rc = self.L.pga_extract_stats(self.h, out)
        if rc != PGA_OK:
            _raise(self.L, self.h, rc, "pga_extract_stats")
        return {"passes": out[0]}

    @staticmethod
    def dp_kernel_name():
        """The connection-scoring kernel a launch with many (>= 2048) chains runs (dp.hip, pga_launch_dp)."""
        return {"tree1": "k_dp_tree", "scan": "k_dp_chain"}.get(os.environ.get("PGA_DP_KERNEL", ""), "k_dp_wave")

    def set_models(self, blobs):
        """``blobs``: iterable of 558 392-byte ``struct _training`` buffers (bytes / uint8 arrays)."""
        arrs = []
        for b in blobs:
            a = np.frombuffer(bytes(b), dtype=np.uint8).copy() if not isinstance(b, np.ndarray) else np.ascontiguousarray(b, np.uint8)
            if a.size != TRAINING_SIZE:
                raise ValueError(f"training info must be {TRAINING_SIZE} bytes, got {a.size}")
            arrs.append(a)
        ptrs = (ctypes.c_void_p * max(1, len(arrs)))(*[a.ctypes.data for a in arrs])
        rc = self.L.pga_set_models(self.h, ptrs, len(arrs))
        if rc != PGA_OK:
            _raise(self.L, self.h, rc, "pga_set_models")
        self._models = arrs

    def score_connections(self, ndx, stop_val, type_, strand, cscore, sscore, rscore, uscore, star_ptr, st_wt, final=True):
        """Whole-array ``ConnectionScorer.index`` + ``score_connections``. Returns (score, traceb, ov_mark, max_index, kernel_ms)."""
        n = len(ndx)
        c = lambda a, t: np.ascontiguousarray(a, dtype=t)
        ndx, stop_val = c(ndx, np.int32), c(stop_val, np.int32)
        type_, strand = c(type_, np.uint8), c(strand, np.int8)
        cscore, sscore, rscore, uscore = (c(x, np.float64) for x in (cscore, sscore, rscore, uscore))
        star_ptr = c(star_ptr, np.int32).reshape(-1)
        score = np.zeros(n, np.float64); traceb = np.zeros(n, np.int32); ov = np.zeros(n, np.int8)
        mi, ms = ctypes.c_int32(-1), ctypes.c_double(0)
        p = lambda a: a.ctypes.data
        rc = self.L.pga_score_connections(self.h, n, p(ndx), p(stop_val), p(type_), p(strand), p(cscore), p(sscore),
                                          p(rscore), p(uscore), p(star_ptr), float(st_wt), int(final),
                                          p(score), p(traceb), p(ov), ctypes.byref(mi), ctypes.byref(ms))
        if rc != PGA_OK:
            _raise(self.L, self.h, rc, "pga_score_connections")
        return score, traceb, ov, mi.value, ms.value

    def score_connections_training(self, ndx, stop_val, type_, strand, gc_score, bias, star_ptr, st_wt):
        """The training pass (``final=False``) of the same scorer, from the nodes' frame-bias scores."""
        n = len(ndx)
        c = lambda a, t: np.ascontiguousarray(a, dtype=t)
        ndx, stop_val = c(ndx, np.int32), c(stop_val, np.int32)
        type_, strand = c(type_, np.uint8), c(strand, np.int8)
        gc_score, bias = c(gc_score, np.float64).reshape(-1), c(bias, np.float64)
        star_ptr = c(star_ptr, np.int32).reshape(-1)
        score = np.zeros(n, np.float64); traceb = np.zeros(n, np.int32); ov = np.zeros(n, np.int8)
        mi, ms = ctypes.c_int32(-1), ctypes.c_double(0)
        p = lambda a: a.ctypes.data
        rc = self.L.pga_score_connections_training(self.h, n, p(ndx), p(stop_val), p(type_), p(strand), p(gc_score), p(bias),
                                                   p(star_ptr), float(st_wt), p(score), p(traceb), p(ov), ctypes.byref(mi), ctypes.byref(ms))
        if rc != PGA_OK:
            _raise(self.L, self.h, rc, "pga_score_connections_training")
        return score, traceb, ov, mi.value, ms.value


_NODE_FIELDS = [
    ("ndx", np.int32, 1), ("stop_val", np.int32, 1), ("traceb", np.int32, 1), ("tracef", np.int32, 1),
    ("star_ptr", np.int32, 3), ("type", np.uint8, 1), ("edge", np.uint8, 1), ("elim", np.uint8, 1),
    ("rbs", np.uint8, 2), ("strand", np.int8, 1), ("ov_mark", np.int8, 1), ("gc_cont", np.float32, 1),
    ("cscore", np.float64, 1), ("sscore", np.float64, 1), ("rscore", np.float64, 1), ("uscore", np.float64, 1),
    ("tscore", np.float64, 1), ("score", np.float64, 1), ("mot_score", np.float64, 1), ("mot_ndx", np.int32, 1),
    ("mot_len", np.uint8, 1), ("mot_spacer", np.uint8, 1), ("mot_spacendx", np.uint8, 1),
]


class BatchResult:
    """Host copy of a ``pga_result``: ``contigs`` / ``genes`` structured arrays (+ per-contig node dicts)."""

    def __init__(self, contigs, genes, nodes, t_total_ms, t_dp_ms, node_passes, n_chains=0, masks=None):
        self.contigs, self.genes, self.nodes = contigs, genes, nodes
        self.masks = masks          # per contig an (k, 2) array of [begin, end) intervals, or None when masking is off
        self.t_total_ms, self.t_dp_ms, self.node_passes, self.n_chains = t_total_ms, t_dp_ms, node_passes, n_chains

    def genes_of(self, i):
        c = self.contigs[i]
        return self.genes[c["gene_begin"]:c["gene_begin"] + c["n_genes"]]


_SEQ_POINTERS = [False]


def _seq_pointers_fn():
    """``pyrodigal_amd.lib._seq_pointers`` (the Cython host layer's C loop over a list of bytes), or None when that module is not
    built: argument marshalling only -- the ctypes conversions below do the same, slower."""
    if _SEQ_POINTERS[0] is False:
        try:
            from . import lib as _lib
            _SEQ_POINTERS[0] = _lib._seq_pointers
        except Exception:
            _SEQ_POINTERS[0] = None
    return _SEQ_POINTERS[0]


class Batch:
    """Contigs packed and resident in HBM (``pga_batch``)."""

    def __init__(self, ctx, seqs):
        self.ctx = ctx
        self.n = len(seqs)
        fast = _seq_pointers_fn() if type(seqs) is list else None
        arrays = None
        if fast is not None:
            try:
                arrays = fast(seqs)             # every contig a bytes object: the argument arrays by one C loop
            except TypeError:
                arrays = None
        if arrays is not None:
            p_arr, l_arr, self.total = arrays
            ptrs = p_arr.ctypes.data_as(ctypes.POINTER(ctypes.c_char_p))
            lens = l_arr.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
        else:
            seqs = [s.encode("ascii") if isinstance(s, str) else bytes(s) for s in seqs]
            self.total = sum(len(s) for s in seqs)
            ptrs = (ctypes.c_char_p * max(1, self.n))(*seqs)
            lens = (ctypes.c_int64 * max(1, self.n))(*[len(s) for s in seqs])
        h = ctypes.c_void_p()
        rc = ctx.L.pga_batch_create(ctx.h, self.n, ptrs, lens, ctypes.byref(h))
        del arrays
        if rc != PGA_OK:
            _raise(ctx.L, ctx.h, rc, "pga_batch_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.pga_batch_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


class _ResultOwner:
    """Keeps a ``pga_result`` alive for as long as an array built over its memory is."""

    def __init__(self, L, res):
        self.L, self.res = L, res

    def __del__(self):
        if self.res is not None:
            self.L.pga_result_free(self.res)
            self.res = None


def _view(owner, ptr, count, ctype, dtype):
    """A numpy array over ``count`` C structs at ``ptr`` (no copy); the array keeps the owning result alive."""
    if not count or not ptr:
        return np.zeros(0, dtype)
    buf = (ctype * count).from_address(ctypes.addressof(ptr.contents))
    buf._owner = owner                       # numpy array -> base: this ctypes array -> the result
    return np.frombuffer(buf, dtype=dtype)


def _unpack_result(L, res, want_nodes):
    owner = _ResultOwner(L, res)
    r = res.contents
    # contigs and genes stay where the C library put them: two views, no copies (32 MB of gene records per 250 Mbp batch)
    contigs = _view(owner, r.contigs, r.n_contigs, ContigResult, CONTIG_DTYPE)
    genes = _view(owner, r.genes, r.n_genes, Gene, GENE_DTYPE)
    nodes = None
    if want_nodes and r.nodes:
        nodes = []
        for i in range(r.n_contigs):
            nd = r.nodes[i]
            d = {"n": nd.n}
            for name, dt, mult in _NODE_FIELDS:
                ptr = getattr(nd, name)
                if nd.n == 0 or not ptr:
                    a = np.zeros((0, mult) if mult > 1 else 0, dt)
                else:
                    a = np.ctypeslib.as_array(ptr, (nd.n * mult,)).copy()
                    if mult > 1:
                        a = a.reshape(nd.n, mult)
                d[name] = a
            nodes.append(d)
    masks = None
    if r.mask_off:
        off = np.ctypeslib.as_array(r.mask_off, (r.n_contigs + 1,)).copy()
        iv = np.ctypeslib.as_array(r.masks, (2 * int(off[-1]),)).copy().reshape(-1, 2) if off[-1] else np.zeros((0, 2), np.int32)
        masks = [iv[off[i]:off[i + 1]] for i in range(r.n_contigs)]
    return BatchResult(contigs, genes, nodes, r.t_total_ms, r.t_dp_ms, r.node_passes, r.n_chains, masks)


def _upload(self, seqs):
    return Batch(self, seqs)


class PackedRecords:
    """One batch of a :class:`FastaReader` in packed form: the letters of all records back to back in a pinned staging arena of
    the reader (``ptr``), record i at ``offs[i]``, ``lens[i]`` long.  ``ids`` / ``descriptions`` are Python strings.  The arena
    is handed back to the reader by ``release()`` (called by ``Context.upload_packed`` once the letters are on the device)."""

    def __init__(self, reader, n, ids, descriptions, ptr, offs, lens, arena):
        self.reader, self.n, self.ids, self.descriptions = reader, n, ids, descriptions
        self.ptr, self.offs, self.lens, self.arena = ptr, offs, lens, arena
        self.total = int(offs[n]) if n else 0

    def sequence(self, i):
        """A copy of record i's letters (only while the arena has not been released)."""
        return ctypes.string_at(self.ptr + int(self.offs[i]), int(self.lens[i]))

    def release(self):
        if self.arena is not None:
            self.reader._free[self.arena].set()
            self.arena = None


def _upload_packed(self, pb):
    """A resident :class:`Batch` straight from a reader's pinned staging arena: no host-side packing, one DMA."""
    b = Batch.__new__(Batch)
    b.ctx, b.n, b.total = self, pb.n, pb.total
    offs = (ctypes.c_int64 * max(1, pb.n + 1))(*[int(x) for x in pb.offs[:pb.n + 1]])
    lens = (ctypes.c_int64 * max(1, pb.n))(*[int(x) for x in pb.lens[:pb.n]])
    h = ctypes.c_void_p()
    rc = self.L.pga_batch_create_packed(self.h, pb.n, ctypes.c_void_p(pb.ptr), offs, lens, ctypes.byref(h))
    pb.release()
    if rc != PGA_OK:
        _raise(self.L, self.h, rc, "pga_batch_create_packed")
    b.h = h
    return b


def _find_genes(self, batch, meta=True, closed=False, min_gene=90, min_edge_gene=60, max_overlap=60, want_nodes=False,
                mask=False, min_mask=50):
    """``GeneFinder.find_genes`` over every contig of a resident :class:`Batch`."""
    p = Params(int(closed), min_gene, min_edge_gene, max_overlap, int(meta), int(want_nodes), int(mask), min_mask)
    res = _P(Result)()
    rc = self.L.pga_find_genes(self.h, batch.h, ctypes.byref(p), ctypes.byref(res))
    if rc != PGA_OK:
        _raise(self.L, self.h, rc, "pga_find_genes")
    return _unpack_result(self.L, res, want_nodes)


def _find_genes_batch(self, seqs, **kw):
    """Upload + find + free: ``seqs`` is a list of ASCII ``bytes``/``str`` contigs."""
    b = Batch(self, seqs)
    try:
        return _find_genes(self, b, **kw)
    finally:
        b.close()


def _nodes_stage(self, seqs, stage, translation_table=11, closed=False, min_gene=90, min_edge_gene=60, max_overlap=60,
                 is_meta=False, mask=False, min_mask=50):
    """Node arrays after ``Nodes.extract`` (stage 1), ``Nodes.score`` (2) or overlapping starts (3), one dict per contig.

    Stages 2 and 3 score with model 0 of the context (``set_models`` first)."""
    b = seqs if isinstance(seqs, Batch) else Batch(self, seqs)
    try:
        p = Params(int(closed), min_gene, min_edge_gene, max_overlap, int(is_meta), 1, int(mask), min_mask)
        res = _P(Result)()
        rc = self.L.pga_nodes_stage(self.h, b.h, ctypes.byref(p), int(stage), int(translation_table), ctypes.byref(res))
        if rc != PGA_OK:
            _raise(self.L, self.h, rc, "pga_nodes_stage")
        out = _unpack_result(self.L, res, True)
        return out if stage == STAGE_SEQUENCE else out.nodes
    finally:
        if b is not seqs:
            b.close()


def _train(self, seq, translation_table=11, start_weight=4.35, force_nonsd=False, closed=False, min_gene=90, min_edge_gene=60,
           max_overlap=60, mask=False, min_mask=50, upto=0):
    """``GeneFinder.train`` on one sequence: returns the 558 392-byte ``struct _training`` as ``bytes``."""
    b = Batch(self, [seq])
    try:
        p = Params(int(closed), min_gene, min_edge_gene, max_overlap, 0, 0, int(mask), min_mask)
        out = ctypes.create_string_buffer(TRAINING_SIZE)
        rc = self.L.pga_train(self.h, b.h, ctypes.byref(p), int(translation_table), float(start_weight), int(force_nonsd), int(upto), out)
        if rc != PGA_OK:
            _raise(self.L, self.h, rc, "pga_train")
        return out.raw
    finally:
        b.close()


def _translate_genes(self, batch, result, tables=None, unknown_residue="X", include_stop=True, strict=True):
    """Proteins of ``result.genes`` (a result of ``find_genes`` on the resident ``batch``), translated on the device.

    ``tables``: translation table per contig (default: the table of the model that won the contig).  Returns
    ``(letters, offsets)``: gene g is ``letters[offsets[g]:offsets[g + 1]]`` (a uint8 array of ASCII codes)."""
    genes = np.ascontiguousarray(result.genes)
    n = len(genes)
    if tables is None:
        tts = [int(np.frombuffer(m[8:12].tobytes(), np.int32)[0]) for m in self._models]
        tables = [tts[c["model"]] if c["model"] >= 0 else 11 for c in result.contigs]
    tables = np.ascontiguousarray(tables, np.int32)
    stop_edge = np.where(genes["strand"] == 1, genes["partial_end"], genes["partial_begin"]).astype(bool)
    lens = (genes["end"].astype(np.int64) - genes["begin"] + 1) // 3
    if not include_stop:
        lens = np.maximum(lens - (~stop_edge), 0)
    off = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    out = np.zeros(max(int(off[-1]), 1), np.uint8)
    unk = unknown_residue.encode("ascii") if isinstance(unknown_residue, str) else bytes(unknown_residue)
    if len(unk) != 1:
        raise ValueError("`unknown_residue` must be a single character")
    rc = self.L.pga_translate_genes(self.h, batch.h, n, genes.ctypes.data, tables.ctypes.data, unk[0], int(include_stop), int(strict),
                                    off.ctypes.data, out.ctypes.data)
    if rc != PGA_OK:
        _raise(self.L, self.h, rc, "pga_translate_genes")
    return out[:int(off[-1])], off


Context.translate_genes = _translate_genes
Context.train = _train
Context.upload = _upload
Context.upload_packed = _upload_packed
Context.nodes_stage = _nodes_stage
Context.find_genes = _find_genes
Context.find_genes_batch = _find_genes_batch


class FastaReader:
    """Multi-record FASTA reader of the C library (ref: tests/fasta.py:59-86 `parse`, 16-57 `zopen`): plain files are mapped and
    parsed by several threads, gzip is inflated by zlib, bz2 / xz (and lz4 / zstd when their modules are installed) by the Python
    module of the format feeding the same C parser.

    ``batches()`` yields lists of ``(id, description, sequence_bytes)`` bounded by a base / record budget, the
    shape ``Context.find_genes_batch`` takes; ``records()`` yields them one by one."""

    _MAGIC = ((b"BZh", "bz2"), (b"\xfd7zXZ", "lzma"), (b"\x04\x22\x4d\x18", "lz4.frame"), (b"\x28\xb5\x2f\xfd", "zstandard"))

    def __init__(self, path):
        self.L = load()
        self.h = ctypes.c_void_p()
        self._stream = self._cb = self._cb_error = None
        with open(path, "rb") as f:
            head = f.read(8)
        module = next((m for magic, m in self._MAGIC if head.startswith(magic)), None)
        if module is None:
            # plain (mapped, parsed by several threads) or gzip (zlib)
            rc = self.L.pga_fasta_open(os.fsencode(path), ctypes.byref(self.h))
        else:
            # the formats the reference's reader sniffs (tests/fasta.py:16-57): decompressed by the Python module, parsed by the C reader
            import importlib
            try:
                mod = importlib.import_module(module)
            except ImportError as err:
                raise RuntimeError("File compression is %s but %s is not installed" % (module.split(".")[0].upper(), module.split(".")[0])) from err
            self._stream = mod.ZstdDecompressor().stream_reader(open(path, "rb")) if module == "zstandard" else mod.open(path, "rb")
            stream = self._stream

            def read(_user, buf, cap):
                # ctypes swallows whatever a callback raises (KeyboardInterrupt included) and hands 0 -- "end of stream" -- to the C
                # reader: a truncated record set without an error.  So everything is caught here, kept, and reported as a failure;
                # batches() / packed_batches() raise it again, chained.
                try:
                    data = stream.read(int(cap))
                except BaseException as err:
                    self._cb_error = err
                    return -1
                ctypes.memmove(buf, data, len(data))
                return len(data)
            self._cb = FASTA_READ_FN(read)
            rc = self.L.pga_fasta_open_callback(self._cb, None, ctypes.byref(self.h))
        if rc != PGA_OK:
            raise (MemoryError if rc == PGA_ENOMEM else OSError)("cannot open %r" % (path,))

    def close(self):
        if getattr(self, "h", None):
            self.L.pga_fasta_close(self.h)
            self.h = None
        if getattr(self, "_stream", None) is not None:
            self._stream.close()
            self._stream = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def batches(self, max_bases=64 << 20, max_records=0):
        n = ctypes.c_int32()
        hdr = _P(ctypes.c_char_p)(); seq = _P(ctypes.c_void_p)(); lens = _P(ctypes.c_int64)()
        while True:
            rc = self.L.pga_fasta_next(self.h, max_bases, max_records, ctypes.byref(n), ctypes.byref(hdr), ctypes.byref(seq), ctypes.byref(lens))
            if rc != PGA_OK:
                self._raise(rc)
            if n.value == 0:
                return
            out = []
            for i in range(n.value):
                fields = hdr[i].decode("utf-8", "replace").split(maxsplit=1)
                out.append((fields[0] if fields else "", fields[1] if len(fields) > 1 else "", ctypes.string_at(seq[i], lens[i])))
            yield out

    def _raise(self, rc):
        """The reader's error; what the decompressor raised inside the read callback comes with it (an interrupt comes back as itself)."""
        cause, self._cb_error = self._cb_error, None
        if cause is not None and not isinstance(cause, Exception):
            raise cause
        err = (MemoryError if rc == PGA_ENOMEM else ValueError)(self.L.pga_fasta_error(self.h).decode("utf-8", "replace"))
        if cause is not None:
            raise err from cause
        raise err

    def records(self):
        for batch in self.batches():
            yield from batch

    def packed_batches(self, max_bases=64 << 20, max_records=0, n_arenas=3):
        """Yield :class:`PackedRecords`: the reader copies every batch into one of ``n_arenas`` pinned staging arenas, filled
        in turn.  A batch's arena is reused ``n_arenas`` batches later, and only after ``release()`` was called on it --
        parsing batch k + 1 overlaps the upload and the device work of batch k."""
        import threading
        n_arenas = max(2, min(8, int(n_arenas)))
        self._free = [threading.Event() for _ in range(n_arenas)]
        for e in self._free:
            e.set()
        n = ctypes.c_int32()
        hdr = _P(ctypes.c_char_p)(); packed = ctypes.c_void_p(); offs = _P(ctypes.c_int64)(); lens = _P(ctypes.c_int64)()
        k = 0
        while True:
            arena = k % n_arenas
            self._free[arena].wait()
            self._free[arena].clear()
            rc = self.L.pga_fasta_next_packed(self.h, max_bases, max_records, n_arenas, ctypes.byref(n), ctypes.byref(hdr),
                                              ctypes.byref(packed), ctypes.byref(offs), ctypes.byref(lens))
            if rc != PGA_OK:
                self._free[arena].set()
                self._raise(rc)
            if n.value == 0:
                self._free[arena].set()
                return
            ids, descs = [], []
            for i in range(n.value):
                fields = hdr[i].decode("utf-8", "replace").split(maxsplit=1)
                ids.append(fields[0] if fields else ""); descs.append(fields[1] if len(fields) > 1 else "")
            o = np.ctypeslib.as_array(offs, (n.value + 1,)).copy()
            ln = np.ctypeslib.as_array(lens, (n.value,)).copy()
            yield PackedRecords(self, n.value, ids, descs, packed.value, o, ln, arena)
            k += 1
