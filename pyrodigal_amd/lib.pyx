# cython: language_level=3, boundscheck=False, wraparound=False, cdivision=True
"""Cython host layer: pyrodigal's `GeneFinder.find_genes()` / `Genes` API over the HIP C-ABI.

Same names, arguments, defaults, validation and error types as the reference
(/root/reference/src/pyrodigal/lib.pyx; citations below are into that file), but every
per-base / per-node computation happens in `libpyrodigal_amd.so` on an MI355X.  There is no
CPU path: without the library or a gfx950 device, calls raise `RuntimeError`.

`Nodes.extract / sort / reset_scores / score` and `ConnectionScorer` are the stage-level calls of the
reference (lib.pyx:2501-2595, 1315-1357) over `pga_nodes_stage` / `pga_score_connections`; the scorer
works on whole node arrays, not node by node (SURVEY 8b: per-node granularity is useless for a GPU).

`Gene.translate` and the `Genes.write_*` writers are host-side formatting, as in the reference.
`GeneFinder.train` runs on the device too (`pga_train`).
"""
import gzip
import threading

import numpy as np

from cpython.bytes cimport PyBytes_FromStringAndSize, PyBytes_AS_STRING
from libc.stdint cimport int8_t, uint8_t, int32_t, int64_t
from libc.stdlib cimport malloc, free
from libc.string cimport memcpy, memset
from libc.math cimport exp, fmax

cdef extern from "pyrodigal_amd.h" nogil:
    ctypedef struct pga_ctx
    ctypedef struct pga_batch
    ctypedef struct pga_training
    ctypedef struct pga_params:
        int32_t closed
        int32_t min_gene
        int32_t min_edge_gene
        int32_t max_overlap
        int32_t meta
        int32_t want_nodes
        int32_t mask
        int32_t min_mask
    ctypedef struct pga_gene:
        int32_t contig
        int32_t begin
        int32_t end
        int32_t start_ndx
        int32_t stop_ndx
        int8_t strand
        uint8_t partial_begin
        uint8_t partial_end
        uint8_t start_type
        uint8_t rbs[2]
        uint8_t mot_len
        uint8_t mot_spacer
        int32_t mot_ndx
        float gc_cont
        double cscore
        double sscore
        double rscore
        double uscore
        double tscore
        double mot_score
    ctypedef struct pga_nodes:
        int32_t n
        int32_t* ndx
        int32_t* stop_val
        int32_t* traceb
        int32_t* tracef
        int32_t* star_ptr
        uint8_t* type
        uint8_t* edge
        uint8_t* elim
        uint8_t* rbs
        int8_t* strand
        int8_t* ov_mark
        float* gc_cont
        double* cscore
        double* sscore
        double* rscore
        double* uscore
        double* tscore
        double* score
        double* mot_score
        int32_t* mot_ndx
        uint8_t* mot_len
        uint8_t* mot_spacer
        uint8_t* mot_spacendx
    ctypedef struct pga_contig_result:
        int32_t model
        int32_t n_nodes
        int64_t gene_begin
        int32_t n_genes
        int32_t n_unknown
        double gc
        double score
    ctypedef struct pga_result:
        int32_t n_contigs
        int64_t n_genes
        pga_contig_result* contigs
        pga_gene* genes
        pga_nodes* nodes
        double t_total_ms
        double t_dp_ms
        int64_t node_passes
        int32_t n_chains
        int32_t* mask_off
        int32_t* masks
    int PGA_OK, PGA_EINVAL, PGA_ENOMEM, PGA_EDEVICE, PGA_ENODEVICE
    int pga_create(int device, pga_ctx** out)
    void pga_destroy(pga_ctx*)
    const char* pga_last_error(const pga_ctx*)
    int pga_set_models(pga_ctx*, const pga_training* const* models, int n_models)
    int pga_find_genes_batch(pga_ctx*, int32_t n, const char* const* seqs, const int64_t* lens,
                             const pga_params*, pga_result** out)
    void pga_result_free(pga_result*)
    int pga_find_genes(pga_ctx*, const pga_batch*, const pga_params*, pga_result** out)
    int pga_translate_genes(pga_ctx*, const pga_batch*, int64_t n_genes, const pga_gene* genes, const int32_t* table_of_contig,
                            int unknown_residue, int include_stop, int strict, const int64_t* offsets, char* out)
    int pga_batch_create(pga_ctx*, int32_t n, const char* const* seqs, const int64_t* lens, pga_batch** out)
    void pga_batch_free(pga_batch*)
    int PGA_STAGE_EXTRACT, PGA_STAGE_SCORE, PGA_STAGE_OVERLAP, PGA_STAGE_SEQUENCE
    int pga_nodes_stage(pga_ctx*, const pga_batch*, const pga_params*, int stage, int translation_table, pga_result** out)
    int pga_train(pga_ctx*, const pga_batch*, const pga_params*, int translation_table, double start_weight, int force_nonsd,
                  int upto, pga_training* out)
    int pga_score_connections(pga_ctx*, int32_t n, const int32_t* ndx, const int32_t* stop_val, const uint8_t* type,
                              const int8_t* strand, const double* cscore, const double* sscore, const double* rscore,
                              const double* uscore, const int32_t* star_ptr, double st_wt, int final,
                              double* score, int32_t* traceb, int8_t* ov_mark, int32_t* max_index, double* kernel_ms)
    int pga_score_connections_training(pga_ctx*, int32_t n, const int32_t* ndx, const int32_t* stop_val, const uint8_t* type,
                                       const int8_t* strand, const double* gc_score, const double* bias, const int32_t* star_ptr,
                                       double st_wt, double* score, int32_t* traceb, int8_t* ov_mark, int32_t* max_index,
                                       double* kernel_ms)

# --- constants (ref: lib.pyx:166-228) ------------------------------------------------------
MIN_SINGLE_GENOME = 20000
IDEAL_SINGLE_GENOME = 100000
TRANSLATION_TABLES = frozenset(set(range(1, 7)) | set(range(9, 17)) | set(range(21, 27)) | {29, 30, 32, 33})
PRODIGAL_VERSION = "v2.6.3+c1e2d36"
_VERSION = "0.1.0"
TRAINING_INFO_SIZE = 558392

_RBS_MOTIF = [
    None, "GGA/GAG/AGG", "3Base/5BMM", "4Base/6BMM", "AGxAG", "AGxAG", "GGA/GAG/AGG", "GGxGG", "GGxGG",
    "AGxAG", "AGGAG(G)/GGAGG", "AGGA/GGAG/GAGG", "AGGA/GGAG/GAGG", "GGA/GAG/AGG", "GGxGG", "AGGA",
    "GGAG/GAGG", "AGxAGG/AGGxGG", "AGxAGG/AGGxGG", "AGxAGG/AGGxGG", "AGGAG/GGAGG", "AGGAG", "AGGAG",
    "GGAGG", "GGAGG", "AGGAGG", "AGGAGG", "AGGAGG",
]
_RBS_SPACER = [
    None, "3-4bp", "13-15bp", "13-15bp", "11-12bp", "3-4bp", "11-12bp", "11-12bp", "3-4bp", "5-10bp",
    "13-15bp", "3-4bp", "11-12bp", "5-10bp", "5-10bp", "5-10bp", "5-10bp", "11-12bp", "3-4bp", "5-10bp",
    "11-12bp", "3-4bp", "5-10bp", "3-4bp", "5-10bp", "11-12bp", "3-4bp", "5-10bp",
]
_NODE_TYPE = ["ATG", "GTG", "TTG", "Edge"]

# NCBI genetic codes: amino acids of the 64 codons in TCAG order (first base slowest).  Table numbers as in
# the reference (_translation.h; TRANSLATION_TABLES above).
_NCBI_CODES = {
    1: "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
    2: "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSS**VVVVAAAADDEEGGGG",
    3: "FFLLSSSSYY**CCWWTTTTPPPPHHQQRRRRIIMMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
    4: "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
    5: "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSSSVVVVAAAADDEEGGGG",
    6: "FFLLSSSSYYQQCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
    9: "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG",
    10: "FFLLSSSSYY**CCCWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
    11: "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
    12: "FFLLSSSSYY**CC*WLLLSPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
    13: "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSGGVVVVAAAADDEEGGGG",
    14: "FFLLSSSSYYY*CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG",
    15: "FFLLSSSSYY*QCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
    16: "FFLLSSSSYY*LCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
    21: "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNNKSSSSVVVVAAAADDEEGGGG",
    22: "FFLLSS*SYY*LCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
    23: "FF*LSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
    24: "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSSKVVVVAAAADDEEGGGG",
    25: "FFLLSSSSYY**CCGWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
    26: "FFLLSSSSYY**CC*WLLLAPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
    29: "FFLLSSSSYYYYCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
    30: "FFLLSSSSYYEECC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
    32: "FFLLSSSSYY*WCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
    33: "FFLLSSSSYYY*CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSSKVVVVAAAADDEEGGGG",
}


def _digit_order_table(str code):
    """Re-index a TCAG-ordered code by the digit alphabet used on the device (A0 G1 C2 T3)."""
    ncbi = (2, 3, 1, 0)     # digit -> position in TCAG
    return "".join(code[ncbi[a] * 16 + ncbi[b] * 4 + ncbi[c]] for a in range(4) for b in range(4) for c in range(4)).encode("ascii")

_CODE_BY_DIGITS = {tt: _digit_order_table(code) for tt, code in _NCBI_CODES.items()}
_DIGIT_OF = bytes(0 if c in b"Aa" else 1 if c in b"Gg" else 2 if c in b"Cc" else 3 if c in b"Tt" else 6 for c in range(256))


cdef bint _codon_is_stop(int x0, int x1, int x2, int tt) noexcept nogil:       # ref: _sequence.h:19-43
    if x0 == 0 and tt == 2:
        return x1 == 1 and (x2 == 0 or x2 == 1)                                  # AGA / AGG
    if x0 != 3:
        return False
    if x1 == 0 and x2 == 1:                                                      # TAG
        return tt in (1, 2, 3, 4, 5, 9, 10, 11, 12, 13, 14, 21, 23, 24, 25, 26, 33)
    if x1 == 1 and x2 == 0:                                                      # TGA
        return tt in (1, 6, 11, 12, 15, 16, 22, 23, 26, 29, 30, 32)
    if x1 == 0 and x2 == 0:                                                      # TAA
        return tt in (1, 2, 3, 4, 5, 9, 10, 11, 12, 13, 15, 16, 21, 22, 23, 24, 25, 26, 32)
    if tt == 22:
        return x1 == 2 and x2 == 0                                               # TCA
    if tt == 23:
        return x1 == 3 and x2 == 0                                               # TTA
    return False


cdef bint _codon_is_start(int x0, int x1, int x2, int tt) noexcept nogil:      # ref: _sequence.h:45-73
    if x1 != 3 or x2 != 1:
        return False
    if x0 == 0:
        return True
    if tt in (6, 10, 14, 15, 16, 2):
        return False
    if x0 == 1:
        return not (tt == 1 or tt == 3 or tt == 12 or tt == 2)
    if x0 == 3:
        return not (tt < 4 or tt == 9 or (21 <= tt < 25))
    return False


def _stop_codon_set(int tt):
    return frozenset((a, b, c) for a in range(4) for b in range(4) for c in range(4) if _codon_is_stop(a, b, c, tt))

_STOP_CODONS = {tt: _stop_codon_set(tt) for tt in _NCBI_CODES}
_COMPLEMENT = bytes.maketrans(b"ACGTN", b"TGCAN")
_COMPLEMENT_ANY = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


cdef object _raise_for(pga_ctx* ctx, int rc, str what):
    cdef bytes msg = pga_last_error(ctx) if ctx != NULL else b""
    text = "%s: %s" % (what, msg.decode("utf-8", "replace"))
    if rc == PGA_EINVAL:
        raise ValueError(text)
    if rc == PGA_ENOMEM:
        raise MemoryError(text)
    if rc == PGA_ENODEVICE:
        raise RuntimeError(what + ": no gfx950 (MI355X) device visible; pyrodigal_amd has no CPU fallback")
    raise RuntimeError(text)


# --- TrainingInfo / MetagenomicBins (ref: lib.pyx:3898-4282, 4888-5069) -------------------
cdef class TrainingInfo:
    """The parameters of one gene model: the reference's 558 392-byte ``struct _training``."""
    cdef object _raw              # numpy uint8[558392]; `raw` hands out a read-only view
    cdef readonly unsigned long long _version   # bumped by every setter: device copies of the model are reloaded when it moves

    def __init__(self, double gc=0.5, *, int translation_table=11, double start_weight=4.35, object raw=None):
        if raw is not None:
            arr = np.frombuffer(bytes(raw), dtype=np.uint8).copy()
            if arr.size != TRAINING_INFO_SIZE:
                raise ValueError("a raw training info must be %d bytes (got %d)" % (TRAINING_INFO_SIZE, arr.size))
            self._raw = arr
        else:
            if translation_table not in TRANSLATION_TABLES:
                raise ValueError("%d is not a valid translation table index" % translation_table)
            self._raw = np.zeros(TRAINING_INFO_SIZE, dtype=np.uint8)
            self._f64(0)[0] = gc
            self._i32(8)[0] = translation_table
            self._f64(16)[0] = start_weight
            self._i32(72)[0] = 1

    @classmethod
    def load(cls, fp):
        """Load a training info from a file object or path (raw dump, optionally gzipped) -- ref: lib.pyx:3910-3953."""
        if hasattr(fp, "read"):
            data = fp.read()
        else:
            opener = gzip.open if str(fp).endswith(".gz") else open
            with opener(fp, "rb") as f:
                data = f.read()
        if len(data) != TRAINING_INFO_SIZE:
            raise EOFError("Expected %d bytes, only read %d" % (TRAINING_INFO_SIZE, len(data)))
        return cls(raw=data)

    def dump(self, fp):
        """Write the raw structure to a file object -- ref: lib.pyx:4865-4885."""
        fp.write(self._raw.tobytes())

    @property
    def raw(self):
        """The 558 392 bytes of the structure, read-only: the setters are what tells a device copy of the model that it is stale."""
        v = self._raw.view()
        v.setflags(write=False)
        return v

    def _f64(self, int off, int n=1):
        return self._raw[off:off + 8 * n].view(np.float64)

    def _i32(self, int off):
        return self._raw[off:off + 4].view(np.int32)

    def __repr__(self):
        return "<pyrodigal_amd.lib.TrainingInfo gc=%r start_weight=%r translation_table=%r uses_sd=%r>" % (
            self.gc, self.start_weight, self.translation_table, self.uses_sd)

    @property
    def gc(self):
        return float(self._f64(0)[0])

    @gc.setter
    def gc(self, double v):
        self._version += 1
        if v < 0 or v > 1:
            raise ValueError("Invalid GC percent: %r" % v)
        self._f64(0)[0] = v

    @property
    def translation_table(self):
        return int(self._i32(8)[0])

    @translation_table.setter
    def translation_table(self, int v):
        self._version += 1
        if v not in TRANSLATION_TABLES:
            raise ValueError("%d is not a valid translation table index" % v)
        self._i32(8)[0] = v

    @property
    def start_weight(self):
        return float(self._f64(16)[0])

    @property
    def bias(self):
        return tuple(self._f64(24, 3))

    @property
    def type_weights(self):
        return tuple(self._f64(48, 3))

    @property
    def uses_sd(self):
        return bool(self._i32(72)[0])

    @property
    def rbs_weights(self):
        return self._f64(80, 28).copy()

    @property
    def missing_motif_weight(self):
        return float(self._f64(525616)[0])

    @property
    def coding_statistics(self):
        return self._f64(525624, 4096).copy()

    # the remaining fields and the setters of the reference (ref: lib.pyx:4067-4213)
    @start_weight.setter
    def start_weight(self, double v):
        self._version += 1
        self._f64(16)[0] = v

    @bias.setter
    def bias(self, object v):
        self._version += 1
        self._f64(24, 3)[:] = np.asarray(v, np.float64)

    @type_weights.setter
    def type_weights(self, object v):
        self._version += 1
        self._f64(48, 3)[:] = np.asarray(v, np.float64)

    @uses_sd.setter
    def uses_sd(self, bint v):
        self._version += 1
        self._i32(72)[0] = 1 if v else 0

    @rbs_weights.setter
    def rbs_weights(self, object v):
        self._version += 1
        self._f64(80, 28)[:] = np.asarray(v, np.float64)

    @property
    def upstream_compositions(self):
        return self._f64(304, 128).reshape(32, 4).copy()

    @upstream_compositions.setter
    def upstream_compositions(self, object v):
        self._version += 1
        self._f64(304, 128)[:] = np.asarray(v, np.float64).reshape(-1)

    @property
    def motif_weights(self):
        return self._f64(1328, 65536).reshape(4, 4, 4096).copy()

    @motif_weights.setter
    def motif_weights(self, object v):
        self._version += 1
        self._f64(1328, 65536)[:] = np.asarray(v, np.float64).reshape(-1)

    @missing_motif_weight.setter
    def missing_motif_weight(self, double v):
        self._version += 1
        self._f64(525616)[0] = v

    @coding_statistics.setter
    def coding_statistics(self, object v):
        self._version += 1
        self._f64(525624, 4096)[:] = np.asarray(v, np.float64).reshape(-1)

    def __eq__(self, other):
        return isinstance(other, TrainingInfo) and np.array_equal(self._raw, (<TrainingInfo> other)._raw)

    def __reduce__(self):           # pickling (ref: lib.pyx:4024-4035)
        return _training_info_from_bytes, (self._raw.tobytes(),)


def _training_info_from_bytes(bytes raw):
    return TrainingInfo(raw=raw)


cdef class MetagenomicBin:
    """A pre-trained model with a description (ref: lib.pyx:4888-4946)."""
    cdef readonly TrainingInfo training_info
    cdef readonly str description

    def __init__(self, TrainingInfo training_info not None, str description=""):
        self.training_info = training_info
        self.description = description

    def __repr__(self):
        return "<pyrodigal_amd.lib.MetagenomicBin description=%r>" % self.description

    def __reduce__(self):
        return MetagenomicBin, (self.training_info, self.description)


cdef class MetagenomicBins:
    """An immutable collection of `MetagenomicBin` (ref: lib.pyx:4950-5066)."""
    cdef readonly tuple _bins

    def __init__(self, object iterable=()):
        bins = tuple(iterable)
        for b in bins:
            if not isinstance(b, MetagenomicBin):
                raise TypeError("expected MetagenomicBin, got %s" % type(b).__name__)
        self._bins = bins

    def __len__(self):
        return len(self._bins)

    def __getitem__(self, index):
        if isinstance(index, slice):
            return MetagenomicBins(self._bins[index])
        return self._bins[index]

    def __iter__(self):
        return iter(self._bins)

    def __reduce__(self):
        return MetagenomicBins, (list(self._bins),)


# Prodigal's 50 built-in models are not part of the reference checkout (un-vendored submodule), so the
# default collection is empty: meta mode needs `metagenomic_bins=`.
METAGENOMIC_BINS = MetagenomicBins()


# --- Sequence / Nodes / Gene / Genes ----------------------------------------------------------

cdef class _StageContext:
    """One lazily created device context for the stage-level calls (`Nodes.*`, `ConnectionScorer`, `Sequence`)."""
    cdef pga_ctx* ctx
    cdef object loaded        # the TrainingInfo blob currently loaded as model 0
    cdef unsigned long long loaded_version

    def __cinit__(self):
        self.ctx = NULL
        self.loaded = None

    def __dealloc__(self):
        if self.ctx != NULL:
            pga_destroy(self.ctx)
            self.ctx = NULL

    cdef int ensure(self) except -1:
        cdef int rc
        if self.ctx == NULL:
            rc = pga_create(0, &self.ctx)
            if rc != PGA_OK:
                self.ctx = NULL
                _raise_for(NULL, rc, "pga_create")
        return 0

    cdef int load(self, TrainingInfo tinf) except -1:
        cdef const pga_training* ptr
        cdef int rc
        if self.loaded is tinf._raw and self.loaded_version == tinf._version:
            return 0
        ptr = <const pga_training*> <size_t> tinf._raw.ctypes.data
        rc = pga_set_models(self.ctx, &ptr, 1)
        if rc != PGA_OK:
            _raise_for(self.ctx, rc, "pga_set_models")
        self.loaded = tinf._raw
        self.loaded_version = tinf._version
        return 0


cdef class _StagePool:
    """The stage-level calls are re-entrant like the reference's (each builds private state, ref: lib.pyx:2528-2595): a caller takes
    a context of its own for the duration of its device call -- up to `limit` contexts (one HIP stream and one set of scratch
    buffers each), created on demand -- so that concurrent callers overlap instead of queueing behind one lock."""
    cdef list idle
    cdef int created, limit
    cdef object cv

    def __cinit__(self):
        self.idle = []
        self.created = 0
        self.limit = 3
        self.cv = threading.Condition(threading.Lock())

    cdef _StageContext take(self, TrainingInfo want=None):
        cdef _StageContext s
        cdef ssize_t k
        with self.cv:
            while True:
                if self.idle:
                    # prefer a context that already holds the model the caller is about to use
                    k = len(self.idle) - 1
                    if want is not None:
                        for j in range(len(self.idle)):
                            s = self.idle[j]
                            if s.loaded is want._raw and s.loaded_version == want._version:
                                k = j
                                break
                    return self.idle.pop(k)
                if self.created < self.limit:
                    self.created += 1
                    return _StageContext()
                self.cv.wait()

    cdef void give(self, _StageContext s):
        with self.cv:
            self.idle.append(s)
            self.cv.notify()

cdef _StagePool _STAGES = _StagePool()

cdef class _StageLease:
    """`with _StageLease(tinf) as S:` -- a stage context of the caller's own for one device call."""
    cdef _StageContext s
    cdef TrainingInfo want

    def __cinit__(self, TrainingInfo want=None):
        self.want = want
        self.s = None

    def __enter__(self):
        self.s = _STAGES.take(self.want)
        return self.s

    def __exit__(self, *exc):
        _STAGES.give(self.s)
        self.s = None
        return False



cdef class Mask:
    """A masked region `[begin, end)` of a sequence (ref: lib.pyx:283-340)."""
    cdef readonly int begin
    cdef readonly int end

    def __init__(self, int begin, int end):
        self.begin = begin
        self.end = end

    def __repr__(self):
        return "<pyrodigal_amd.lib.Mask begin=%d end=%d>" % (self.begin, self.end)

    def __eq__(self, other):
        return isinstance(other, Mask) and (self.begin, self.end) == ((<Mask> other).begin, (<Mask> other).end)


cdef class Sequence:
    """The input as ASCII bytes.  Digitising, GC content, the unknown-base count and the masked regions are
    computed on the device (ref: lib.pyx:664-713), on first use."""
    cdef readonly bytes data
    cdef readonly bint mask
    cdef readonly size_t mask_size
    cdef double _gc
    cdef ssize_t _unknown
    cdef list _masks

    def __init__(self, object sequence, bint mask=False, size_t mask_size=50):
        if isinstance(sequence, Sequence):
            self.data = (<Sequence> sequence).data
        elif type(sequence) is bytes:
            self.data = sequence                      # immutable: no copy
        elif isinstance(sequence, str):
            self.data = sequence.encode("ascii", "replace")
        else:
            self.data = bytes(memoryview(sequence))
        self.mask = mask
        self.mask_size = mask_size
        self._gc = -1.0
        self._unknown = -1
        self._masks = None

    def __len__(self):
        return len(self.data)

    def __str__(self):
        up = self.data.upper()
        return "".join(chr(c) if c in b"ACGT" else "N" for c in up)

    cdef int _build(self) except -1:
        cdef pga_params p
        cdef pga_batch* batch = NULL
        cdef pga_result* res = NULL
        cdef const char* ptr = PyBytes_AS_STRING(self.data)
        cdef int64_t length = len(self.data)
        cdef int rc, k
        cdef _StageContext S
        if self._unknown >= 0:
            return 0
        p.closed = 0; p.min_gene = 90; p.min_edge_gene = 60; p.max_overlap = 60; p.meta = 0; p.want_nodes = 0
        p.mask = self.mask; p.min_mask = <int32_t> self.mask_size
        with _StageLease() as S:
            S.ensure()
            rc = pga_batch_create(S.ctx, 1, &ptr, &length, &batch)
            if rc != PGA_OK:
                _raise_for(S.ctx, rc, "pga_batch_create")
            try:
                with nogil:
                    rc = pga_nodes_stage(S.ctx, batch, &p, PGA_STAGE_SEQUENCE, 11, &res)
                if rc != PGA_OK:
                    _raise_for(S.ctx, rc, "pga_nodes_stage")
                try:
                    self._gc = res.contigs[0].gc
                    self._unknown = res.contigs[0].n_unknown
                    self._masks = []
                    if res.mask_off != NULL:
                        for k in range(res.mask_off[0], res.mask_off[1]):
                            self._masks.append(Mask(res.masks[2 * k], res.masks[2 * k + 1]))
                finally:
                    pga_result_free(res)
            finally:
                pga_batch_free(batch)
        return 0

    @property
    def gc(self):
        """GC fraction over all bases (ref: lib.pyx:596-600)."""
        self._build()
        return self._gc

    @property
    def unknown(self):
        """Number of bases that are not A, C, G or T (ref: lib.pyx:602-606)."""
        self._build()
        return self._unknown

    @property
    def gc_known(self):
        """GC fraction over the known bases (ref: lib.pyx:608-614)."""
        self._build()
        cdef ssize_t n = len(self.data)
        cdef double gc_count = round(self._gc * <double> n)      # the device returns count / length; the count is recovered exactly
        return gc_count / (<double> n - <double> self._unknown) if n > self._unknown else 0.0

    @property
    def masks(self):
        """The masked regions, empty unless `mask=True` (ref: lib.pyx:616-620)."""
        self._build()
        return list(self._masks)

    def __reduce__(self):
        return Sequence, (self.data, self.mask, self.mask_size)


cdef class Node:
    """A read-only view of one node (ref: lib.pyx:1437-1552)."""
    cdef readonly Nodes owner
    cdef readonly ssize_t i

    def __getattr__(self, name):
        arr = self.owner._f.get(name)
        if arr is None:
            raise AttributeError(name)
        v = arr[self.i]
        return v.item() if hasattr(v, "item") and getattr(v, "ndim", 0) == 0 else v

    @property
    def index(self):
        return int(self.owner._f["ndx"][self.i])

    @property
    def type(self):
        return _NODE_TYPE[3 if self.owner._f["edge"][self.i] else int(self.owner._f["type"][self.i])] if self.owner._f["type"][self.i] != 3 else "Stop"


cdef class Nodes:
    """The nodes of one sequence as struct-of-arrays (ref: lib.pyx:1652-1795, 2501-2595)."""
    cdef readonly dict _f
    cdef dict _extract_kw      # the arguments of the last extract(), which score() has to repeat

    def __init__(self):
        self._f = {}
        self._extract_kw = None

    def __len__(self):
        return len(self._f["ndx"]) if "ndx" in self._f else 0

    def __getitem__(self, ssize_t i):
        cdef ssize_t n = len(self)
        if i < 0:
            i += n
        if i < 0 or i >= n:
            raise IndexError("node index out of range")
        cdef Node nd = Node.__new__(Node)
        nd.owner = self
        nd.i = i
        return nd

    def array(self, str name):
        """The numpy array of one field (ndx, stop_val, type, strand, edge, cscore, sscore, ...)."""
        return self._f[name]

    def copy(self):
        """A deep copy (ref: lib.pyx:2514-2526)."""
        cdef Nodes out = Nodes()
        out._f = {k: v.copy() for k, v in self._f.items()}
        out._extract_kw = None if self._extract_kw is None else dict(self._extract_kw)
        return out

    def clear(self):
        """Remove all nodes (ref: lib.pyx:2501-2512)."""
        self._f = {}
        self._extract_kw = None

    cdef object _run_stage(self, Sequence seq, int stage, TrainingInfo tinf, bint is_meta):
        cdef pga_params p
        cdef pga_batch* batch = NULL
        cdef pga_result* res = NULL
        cdef const char* ptr = PyBytes_AS_STRING(seq.data)
        cdef int64_t length = len(seq.data)
        cdef int rc, tt
        cdef _StageContext S
        kw = self._extract_kw
        p.closed = kw["closed"]; p.min_gene = kw["min_gene"]; p.min_edge_gene = kw["min_edge_gene"]
        p.max_overlap = 60; p.meta = is_meta; p.want_nodes = 1
        p.mask = seq.mask; p.min_mask = <int32_t> seq.mask_size
        tt = kw["translation_table"]
        with _StageLease(tinf) as S:
            S.ensure()
            if tinf is not None:
                S.load(tinf)
            rc = pga_batch_create(S.ctx, 1, &ptr, &length, &batch)
            if rc != PGA_OK:
                _raise_for(S.ctx, rc, "pga_batch_create")
            try:
                with nogil:
                    rc = pga_nodes_stage(S.ctx, batch, &p, stage, tt, &res)
                if rc != PGA_OK:
                    _raise_for(S.ctx, rc, "pga_nodes_stage")
                try:
                    return _copy_nodes(&res.nodes[0])
                finally:
                    pga_result_free(res)
            finally:
                pga_batch_free(batch)

    def extract(self, Sequence sequence not None, *, bint closed=False, int min_gene=90, int min_edge_gene=60,
                int translation_table=11):
        """Extract the nodes of `sequence`, in sorted order (ref: lib.pyx:2528-2560); returns how many were added."""
        if translation_table not in TRANSLATION_TABLES:
            raise ValueError("%d is not a valid translation table index" % translation_table)
        self._extract_kw = dict(closed=closed, min_gene=min_gene, min_edge_gene=min_edge_gene, translation_table=translation_table)
        cdef Nodes got = self._run_stage(sequence, PGA_STAGE_EXTRACT, None, False)
        self._f = got._f
        return len(self)

    def sort(self):
        """Sort by position then strand (ref: lib.pyx:2575-2580): the device extraction already emits that order."""
        return None

    def reset_scores(self):
        """Reset every score and DP field (ref: lib.pyx:2562-2573)."""
        cdef ssize_t n = len(self)
        for k in ("cscore", "sscore", "rscore", "uscore", "tscore", "score", "mot_score"):
            self._f[k] = np.zeros(n, np.float64)
        self._f["star_ptr"] = np.zeros((n, 3), np.int32)
        self._f["rbs"] = np.zeros((n, 2), np.uint8)
        for k in ("traceb", "tracef"):
            self._f[k] = np.full(n, -1, np.int32)
        self._f["ov_mark"] = np.full(n, -1, np.int8)
        self._f["elim"] = np.zeros(n, np.uint8)
        self._f["mot_ndx"] = np.zeros(n, np.int32)
        for k in ("mot_len", "mot_spacer", "mot_spacendx"):
            self._f[k] = np.zeros(n, np.uint8)

    def score(self, Sequence sequence not None, TrainingInfo training_info not None, *, bint closed=False, bint is_meta=False):
        """Score the start nodes with `training_info` (ref: lib.pyx:2582-2595).

        The device scores the nodes it extracts itself, so `sequence` and the extraction options must be
        the ones `extract` was called with (the reference has the same precondition)."""
        if self._extract_kw is None:
            raise RuntimeError("Nodes.score needs nodes from Nodes.extract")
        if self._extract_kw["translation_table"] != training_info.translation_table:
            raise ValueError("nodes were extracted with translation table %d, the training info uses %d"
                             % (self._extract_kw["translation_table"], training_info.translation_table))
        if self._extract_kw["closed"] != closed:
            raise ValueError("`closed` differs from the value used by Nodes.extract")
        cdef Nodes got = self._run_stage(sequence, PGA_STAGE_SCORE, training_info, is_meta)
        if len(got) != len(self) or not np.array_equal(got._f["ndx"], self._f["ndx"]):
            raise ValueError("sequence does not match the nodes held by this object")
        keep = {k: self._f[k] for k in ("traceb", "tracef", "ov_mark", "score", "elim", "star_ptr") if k in self._f}
        self._f = got._f
        self._f.update(keep)


cdef class ConnectionScorer:
    """Connection scoring of a whole sorted node list on the device (ref: lib.pyx:1297-1435).

    The reference scores one node at a time (`compute_skippable(min, i)` + `score_connections(nodes, min,
    i, tinf, final)`); a GPU needs the whole array, so `score_connections(nodes, tinf, final=True)` runs the
    complete dynamic-programming pass with the reference's window rule and writes `score`, `traceb` and
    `ov_mark` of every node.  `index` and `compute_skippable` are kept so that call sites read the same."""
    cdef readonly str backend
    cdef Nodes _indexed

    def __init__(self, str backend="detect"):
        if backend not in ("detect", "hip"):
            raise ValueError("unsupported backend %r: this build only has the HIP (gfx950) backend" % backend)
        self.backend = "hip"
        self._indexed = None

    def index(self, Nodes nodes not None):
        self._indexed = nodes

    def compute_skippable(self, int min, int i):
        return None     # the skip conditions (impl/generic.h:29-36) are folded into the device scorer

    def score_connections(self, Nodes nodes not None, TrainingInfo training_info not None, bint final=False):
        cdef ssize_t n = len(nodes)
        cdef int rc
        cdef int32_t mi = -1
        f = nodes._f
        if "cscore" not in f:
            nodes.reset_scores()
        if not final:
            return self._score_connections_training(nodes, training_info)
        cdef object ndx = np.ascontiguousarray(f["ndx"], np.int32), stop_val = np.ascontiguousarray(f["stop_val"], np.int32)
        cdef object typ = np.ascontiguousarray(f["type"], np.uint8), strand = np.ascontiguousarray(f["strand"], np.int8)
        cdef object cs = np.ascontiguousarray(f["cscore"], np.float64), ss = np.ascontiguousarray(f["sscore"], np.float64)
        cdef object rs = np.ascontiguousarray(f["rscore"], np.float64), us = np.ascontiguousarray(f["uscore"], np.float64)
        cdef object sp = np.ascontiguousarray(f["star_ptr"], np.int32)
        cdef object score = np.zeros(n, np.float64), traceb = np.full(n, -1, np.int32), ov = np.full(n, -1, np.int8)
        cdef size_t p_ndx = ndx.ctypes.data, p_stop = stop_val.ctypes.data, p_typ = typ.ctypes.data, p_strand = strand.ctypes.data
        cdef size_t p_cs = cs.ctypes.data, p_ss = ss.ctypes.data, p_rs = rs.ctypes.data, p_us = us.ctypes.data, p_sp = sp.ctypes.data
        cdef size_t p_score = score.ctypes.data, p_tb = traceb.ctypes.data, p_ov = ov.ctypes.data
        cdef double st_wt = training_info.start_weight
        cdef _StageContext S
        with _StageLease() as S:
            S.ensure()
            with nogil:
                rc = pga_score_connections(S.ctx, <int32_t> n, <const int32_t*> p_ndx, <const int32_t*> p_stop,
                                           <const uint8_t*> p_typ, <const int8_t*> p_strand, <const double*> p_cs,
                                           <const double*> p_ss, <const double*> p_rs, <const double*> p_us,
                                           <const int32_t*> p_sp, st_wt, 1, <double*> p_score, <int32_t*> p_tb,
                                           <int8_t*> p_ov, &mi, NULL)
            if rc != PGA_OK:
                _raise_for(S.ctx, rc, "pga_score_connections")
        f["score"] = score; f["traceb"] = traceb; f["ov_mark"] = ov
        return int(mi)

    cdef object _score_connections_training(self, Nodes nodes, TrainingInfo training_info):
        # final = False (the reference's default, ref: lib.pyx:1336-1357): a connection is worth its length times the
        # frame-bias factor bias . gc_score of one of its nodes (ref: _connection.h, `final == false` branches).  Nodes that
        # never went through the training carry gc_score = 0, as in the reference after reset_scores().
        cdef ssize_t n = len(nodes)
        cdef int rc
        cdef int32_t mi = -1
        f = nodes._f
        cdef object ndx = np.ascontiguousarray(f["ndx"], np.int32), stop_val = np.ascontiguousarray(f["stop_val"], np.int32)
        cdef object typ = np.ascontiguousarray(f["type"], np.uint8), strand = np.ascontiguousarray(f["strand"], np.int8)
        cdef object gcs = np.ascontiguousarray(f["gc_score"] if "gc_score" in f else np.zeros((n, 3)), np.float64).reshape(-1)
        cdef object bias = np.ascontiguousarray(training_info.bias, np.float64)
        cdef object sp = np.ascontiguousarray(f["star_ptr"], np.int32)
        cdef object score = np.zeros(n, np.float64), traceb = np.full(n, -1, np.int32), ov = np.full(n, -1, np.int8)
        if gcs.size != 3 * n:
            raise ValueError("`gc_score` must hold three frame scores per node")
        cdef size_t p_ndx = ndx.ctypes.data, p_stop = stop_val.ctypes.data, p_typ = typ.ctypes.data, p_strand = strand.ctypes.data
        cdef size_t p_gcs = gcs.ctypes.data, p_bias = bias.ctypes.data, p_sp = sp.ctypes.data
        cdef size_t p_score = score.ctypes.data, p_tb = traceb.ctypes.data, p_ov = ov.ctypes.data
        cdef double st_wt = training_info.start_weight
        cdef _StageContext S
        with _StageLease() as S:
            S.ensure()
            with nogil:
                rc = pga_score_connections_training(S.ctx, <int32_t> n, <const int32_t*> p_ndx, <const int32_t*> p_stop,
                                                    <const uint8_t*> p_typ, <const int8_t*> p_strand, <const double*> p_gcs,
                                                    <const double*> p_bias, <const int32_t*> p_sp, st_wt, <double*> p_score,
                                                    <int32_t*> p_tb, <int8_t*> p_ov, &mi, NULL)
            if rc != PGA_OK:
                _raise_for(S.ctx, rc, "pga_score_connections_training")
        f["score"] = score; f["traceb"] = traceb; f["ov_mark"] = ov
        return int(mi)


cdef double _confidence(double score, double st_wt) noexcept nogil:   # Prodigal gene.c calculate_confidence
    cdef double r = score / st_wt, conf
    if r < 41:
        conf = exp(r)
        conf = (conf / (conf + 1)) * 100.0
    else:
        conf = 99.99
    return fmax(conf, 50.0)


cdef class Gene:
    """A single predicted gene (ref: lib.pyx:2610-3047)."""
    cdef readonly Genes owner
    cdef pga_gene g
    cdef ssize_t _index        # position in the owner's list (-1: unknown), for the owner's device-side translations

    def __cinit__(self):
        self._index = -1

    @property
    def begin(self):
        return self.g.begin

    @property
    def end(self):
        return self.g.end

    @property
    def strand(self):
        return self.g.strand

    @property
    def partial_begin(self):
        return bool(self.g.partial_begin)

    @property
    def partial_end(self):
        return bool(self.g.partial_end)

    @property
    def start_type(self):
        return _NODE_TYPE[self.g.start_type]

    cdef tuple _rbs(self):
        cdef TrainingInfo t = self.owner.training_info
        w = t._f64(80, 28)
        cdef double st = t.start_weight
        cdef double r1 = w[self.g.rbs[0]] * st, r2 = w[self.g.rbs[1]] * st
        cdef double ms = self.g.mot_score * st
        if t.uses_sd:
            k = self.g.rbs[0] if r1 > r2 else self.g.rbs[1]
            return _RBS_MOTIF[k], _RBS_SPACER[k]
        elif t.missing_motif_weight > -0.5 and r1 > r2 and r1 > ms:
            return _RBS_MOTIF[self.g.rbs[0]], _RBS_SPACER[self.g.rbs[0]]
        elif t.missing_motif_weight > -0.5 and r2 >= r1 and r2 > ms:
            return _RBS_MOTIF[self.g.rbs[1]], _RBS_SPACER[self.g.rbs[1]]
        elif self.g.mot_len == 0:
            return None, None
        else:
            motif = "".join("AGCT"[(self.g.mot_ndx >> (2 * i)) & 3] for i in range(self.g.mot_len))
            return motif, "%dbp" % self.g.mot_spacer

    @property
    def rbs_motif(self):
        return self._rbs()[0]

    @property
    def rbs_spacer(self):
        return self._rbs()[1]

    @property
    def gc_cont(self):
        return self.g.gc_cont

    @property
    def translation_table(self):
        return self.owner.training_info.translation_table

    @property
    def cscore(self):
        return self.g.cscore

    @property
    def rscore(self):
        return self.g.rscore

    @property
    def sscore(self):
        return self.g.sscore

    @property
    def tscore(self):
        return self.g.tscore

    @property
    def uscore(self):
        return self.g.uscore

    @property
    def score(self):
        return self.g.cscore + self.g.sscore

    @property
    def start_node(self):
        return self.owner.nodes[self.g.start_ndx] if self.owner.nodes is not None else None

    @property
    def stop_node(self):
        return self.owner.nodes[self.g.stop_ndx] if self.owner.nodes is not None else None

    cpdef double confidence(self):
        return _confidence(self.g.cscore + self.g.sscore, self.owner.training_info.start_weight)

    def sequence(self):
        """The nucleotide sequence of the gene, reverse-complemented on the reverse strand; unknown bases read N
        (ref: lib.pyx:2874-2930)."""
        cdef bytes s = self.owner.sequence.data[self.g.begin - 1:self.g.end].upper()
        s = bytes(c if c in b"ACGT" else 78 for c in s)
        if self.g.strand != 1:
            s = s.translate(_COMPLEMENT)[::-1]
        return s.decode("ascii")

    def translate(self, object translation_table=None, object unknown_residue="X", bint include_stop=True, bint strict=True):
        """Translate the gene into a protein sequence (ref: lib.pyx:2932-3047, `Sequence._amino` 770-789).

        The first codon of a gene that does not start at an edge reads M when it is a start codon of the table;
        a stop codon of the table reads `*`; a codon with an unknown base reads `unknown_residue`, unless
        `strict=False` and every completion of the codon gives the same residue."""
        cdef int tt, owner_tt = self.owner.training_info.translation_table
        if translation_table is None:
            tt = owner_tt
        elif translation_table not in _CODE_BY_DIGITS:
            raise ValueError("%r is not a valid translation table index" % (translation_table,))
        else:
            tt = translation_table
            if _STOP_CODONS[tt] != _STOP_CODONS[owner_tt]:
                import warnings
                warnings.warn("requested translation table (%r) has different STOP codons than the one these genes "
                              "were called with (%r), consider calling genes with the proper translation table instead."
                              % (translation_table, owner_tt), stacklevel=2)
        cdef bytes unk = unknown_residue.encode("ascii") if isinstance(unknown_residue, str) else bytes(unknown_residue)
        if len(unk) != 1:
            raise ValueError("`unknown_residue` must be a single character")
        if (self.owner._prot is not None and self._index >= 0 and tt == self.owner._prot_tt and unk == b"X" and include_stop and strict):
            # translated on the device together with the gene calls (GeneFinder.find_genes_batch(..., translate=True))
            return self.owner._prot[self.owner._prot_off[self._index]:self.owner._prot_off[self._index + 1]].decode("ascii")
        cdef bytes nuc = self.owner.sequence.data[self.g.begin - 1:self.g.end]
        if self.g.strand != 1:
            nuc = nuc.translate(_COMPLEMENT_ANY)[::-1]
        cdef bytes dig = nuc.translate(_DIGIT_OF)
        cdef const unsigned char* d = <const unsigned char*> PyBytes_AS_STRING(dig)
        cdef const char* table = PyBytes_AS_STRING(_CODE_BY_DIGITS[tt])
        cdef ssize_t n = len(dig) // 3, i, k
        # partial flags are in sequence orientation; the gene's own first / last codon follow its strand
        cdef bint start_edge = self.g.partial_begin if self.g.strand == 1 else self.g.partial_end
        cdef bint stop_edge = self.g.partial_end if self.g.strand == 1 else self.g.partial_begin
        if not stop_edge and not include_stop:
            n -= 1
        cdef bytearray out = bytearray(max(n, 0))
        cdef int x0, x1, x2, y
        cdef char aa, c2
        for i in range(n):
            x0 = d[3 * i]; x1 = d[3 * i + 1]; x2 = d[3 * i + 2]
            if x0 <= 3 and x1 <= 3 and x2 <= 3:
                if _codon_is_stop(x0, x1, x2, tt):
                    aa = 42                                                      # '*'
                elif i == 0 and not start_edge and _codon_is_start(x0, x1, x2, tt):
                    aa = 77                                                      # 'M'
                else:
                    aa = table[(x0 << 4) + (x1 << 2) + x2]
            else:
                aa = 88                                                          # 'X'
                if not strict and x0 <= 3 and (x1 <= 3) != (x2 <= 3):
                    # one unknown base in second or third position: unambiguous when all four completions agree
                    aa = table[(x0 << 4) + ((x1 if x1 <= 3 else 0) << 2) + (x2 if x2 <= 3 else 0)]
                    for y in range(1, 4):
                        c2 = table[(x0 << 4) + ((x1 if x1 <= 3 else y) << 2) + (x2 if x2 <= 3 else y)]
                        if c2 != aa:
                            aa = 88
                            break
            out[i] = unk[0] if aa == 88 else aa
        return out.decode("ascii")

    cpdef str _gene_data(self, object sequence_id, ssize_t index):
        motif, spacer = self._rbs()
        return "ID={}_{};partial={}{};start_type={};rbs_motif={};rbs_spacer={};gc_cont={:.3f}".format(
            sequence_id, index + 1, int(self.g.partial_begin), int(self.g.partial_end),
            _NODE_TYPE[self.g.start_type], motif, spacer, self.g.gc_cont)

    cpdef str _score_data(self):
        return "conf={:.2f};score={:.2f};cscore={:.2f};sscore={:.2f};rscore={:.2f};uscore={:.2f};tscore={:.2f};".format(
            self.confidence(), self.score, self.cscore, self.sscore, self.rscore, self.uscore, self.tscore)


cdef class Genes:
    """The genes of one sequence (ref: lib.pyx:3049-3186)."""
    cdef readonly Sequence sequence
    cdef readonly object training_info
    cdef readonly object metagenomic_bin
    cdef readonly bint meta
    cdef readonly double score
    cdef readonly ssize_t _num_seq
    cdef list _genes           # the Gene objects, built from _recs when first asked for
    cdef bytes _recs           # the packed gene records of this sequence as the device call returned them
    cdef ssize_t _n
    cdef bytes _node_blob      # the node arrays of the winning model, field after field (None: keep_nodes=False)
    cdef ssize_t _node_n
    cdef object _nodes
    cdef object _prot          # proteins of all genes back to back, translated on the device with the default arguments, or None
    cdef object _prot_off      # int64[len + 1] offsets into _prot
    cdef int _prot_tt          # the translation table the device translated with

    cdef list _list(self):
        cdef ssize_t j
        cdef Gene gene
        cdef const pga_gene* g
        if self._genes is None:
            out = []
            if self._n > 0:
                g = <const pga_gene*> PyBytes_AS_STRING(self._recs)
                for j in range(self._n):
                    gene = Gene.__new__(Gene)
                    gene.owner = self
                    gene.g = g[j]
                    gene._index = j
                    out.append(gene)
            self._genes = out
        return self._genes

    @property
    def nodes(self):
        """The scored nodes of the sequence (`Nodes`), or None when the finder was created with keep_nodes=False."""
        if self._nodes is None and self._node_blob is not None:
            self._nodes = _nodes_from_blob(self._node_blob, self._node_n)
        return self._nodes

    def __len__(self):
        return self._n

    def __getitem__(self, index):
        return self._list()[index]

    def __iter__(self):
        return iter(self._list())

    def __bool__(self):
        return self._n > 0

    # --- writers (ref: lib.pyx:3405-3894): host-side formatting of the results, byte-compatible with the
    #     reference except for the tool name and version strings -------------------------------------------

    cdef tuple _model(self):
        """(TrainingInfo, description) the header lines report (ref: lib.pyx:3575-3583)."""
        if self.meta:
            if self.metagenomic_bin is None:
                raise RuntimeError("no metagenomic model was selected for this sequence")
            return self.training_info, self.metagenomic_bin.description
        return self.training_info, "Ab initio"

    def write_gff(self, object file, str sequence_id, bint header=True, bint include_translation_table=False,
                  bint full_id=True, str version_separator="_v"):
        """Write the genes to `file` in General Feature Format (ref: lib.pyx:3534-3644)."""
        cdef ssize_t n = 0, i
        cdef Gene gene
        tinf, desc = self._model()
        run = "Metagenomic" if self.meta else "Single"
        if header:
            n += file.write("##gff-version  3\n")
        n += file.write('# Sequence Data: seqnum=%d;seqlen=%d;seqhdr="%s"\n' % (self._num_seq, len(self.sequence), sequence_id))
        n += file.write('# Model Data: version=pyrodigal_amd.v%s;run_type=%s;model="%s";gc_cont=%.2f;transl_table=%d;uses_sd=%d\n'
                        % (_VERSION, run, desc, tinf.gc * 100, tinf.translation_table, int(tinf.uses_sd)))
        for i, gene in enumerate(self._list()):
            ident = gene._gene_data(sequence_id if full_id else self._num_seq, i)
            n += file.write("%s\tpyrodigal_amd%s%s\tCDS\t%d\t%d\t%.1f\t%s\t0\t%s;" % (
                sequence_id, version_separator, _VERSION, gene.g.begin, gene.g.end, gene.g.sscore + gene.g.cscore,
                "+" if gene.g.strand > 0 else "-", ident))
            if include_translation_table:
                n += file.write("transl_table=%d;" % tinf.translation_table)
            n += file.write(gene._score_data())
            n += file.write("\n")
        return n

    def write_genes(self, object file, str sequence_id, object width=70, bint full_id=False):
        """Write the nucleotide sequences of the genes to `file` in FASTA format (ref: lib.pyx:3646-3706)."""
        cdef ssize_t n = 0, i, k
        cdef Gene gene
        for i, gene in enumerate(self._list()):
            n += file.write(">%s_%d # %d # %d # %d # %s\n" % (sequence_id, i + 1, gene.g.begin, gene.g.end, gene.g.strand,
                                                              gene._gene_data(sequence_id if full_id else self._num_seq, i)))
            seq = gene.sequence()
            for k in range(0, len(seq), width):
                n += file.write(seq[k:k + width])
                n += file.write("\n")
        return n

    def write_translations(self, object file, str sequence_id, object width=60, object translation_table=None,
                           bint include_stop=True, bint strict_translation=True, bint full_id=False):
        """Write the protein translations of the genes to `file` in FASTA format (ref: lib.pyx:3708-3792)."""
        cdef ssize_t n = 0, i, k
        cdef Gene gene
        if translation_table is not None and translation_table not in _CODE_BY_DIGITS:
            raise ValueError("%r is not a valid translation table index" % (translation_table,))
        for i, gene in enumerate(self._list()):
            n += file.write(">%s_%d # %d # %d # %d # %s\n" % (sequence_id, i + 1, gene.g.begin, gene.g.end, gene.g.strand,
                                                              gene._gene_data(sequence_id if full_id else self._num_seq, i)))
            prot = gene.translate(translation_table, include_stop=include_stop, strict=strict_translation)
            for k in range(0, len(prot), width):
                n += file.write(prot[k:k + width])
                n += file.write("\n")
        return n

    def write_genbank(self, object file, str sequence_id, str division="BCT", object date=None, object translation_table=None,
                      bint strict_translation=True):
        """Write the genes and the sequence to `file` as a complete GenBank record (ref: lib.pyx:3405-3532)."""
        import datetime
        import textwrap
        cdef ssize_t n = 0, i, j
        cdef Gene gene
        if translation_table is None:
            if self.training_info is not None:
                translation_table = self.training_info.translation_table
        elif translation_table not in _CODE_BY_DIGITS:
            raise ValueError("%r is not a valid translation table index" % (translation_table,))
        if date is None:
            date = datetime.date.today()
        elif not isinstance(date, datetime.date):
            raise TypeError("Expected datetime.date, found %s" % type(date).__name__)
        slen = len(self.sequence)
        n += file.write("LOCUS       {:<23} {} bp    DNA     linear   {} {}\n".format(sequence_id, slen, division, date.strftime("%d-%b-%y").upper()))
        n += file.write("REFERENCE   1  (bases 1 to %d)\n" % slen)
        n += file.write("  AUTHORS   Hyatt,D., Chen,G-L., LoCascio,P.F., Land,M.L., Larimer,F.W.\n")
        n += file.write("            Hauser,L.J.\n")
        n += file.write("  TITLE     Prodigal: prokaryotic gene recognition and translation initiation\n")
        n += file.write("            site identification\n")
        n += file.write("  JOURNAL   BMC Bioinformatics. 2010;11:119.\n")
        n += file.write("   PUBMED   20211023\n")
        n += file.write("FEATURES             Location/Qualifiers\n")
        for i, gene in enumerate(self._list()):
            start_edge = gene.g.partial_begin if gene.g.strand == 1 else gene.g.partial_end
            stop_edge = gene.g.partial_end if gene.g.strand == 1 else gene.g.partial_begin
            begin = "<%d" % gene.g.begin if start_edge else "%d" % gene.g.begin
            end = ">%d" % gene.g.end if stop_edge else "%d" % gene.g.end
            loc = "%s..%s" % (begin, end)
            n += file.write("     CDS             %s\n" % (loc if gene.g.strand == 1 else "complement(%s)" % loc))
            pad = " " * 21
            n += file.write('%s/codon_start=1\n' % pad)
            n += file.write('%s/inference="ab initio prediction:pyrodigal_amd:%s"\n' % (pad, _VERSION))
            n += file.write('%s/locus_tag="%s_%d"\n' % (pad, sequence_id, i + 1))
            n += file.write('%s/transl_table=%s\n' % (pad, translation_table))
            tr = '/translation="%s"' % gene.translate(translation_table=translation_table, include_stop=False, strict=strict_translation)
            for block in textwrap.wrap(tr, 59):
                n += file.write(pad + block + "\n")
        seq = str(self.sequence).lower()
        n += file.write("ORIGIN\n")
        for i in range(0, len(seq), 60):
            n += file.write("{:>9}".format(i + 1))
            for j in range(i, min(i + 60, len(seq)), 10):
                n += file.write(" " + seq[j:j + 10])
            n += file.write("\n")
        n += file.write("//\n")
        return n

    def write_scores(self, object file, str sequence_id, bint header=True):
        """Write the scores of every start node, grouped by stop codon, to `file` (ref: lib.pyx:3794-3894)."""
        cdef ssize_t n = 0
        if self.nodes is None:
            raise RuntimeError("write_scores needs the nodes: create the GeneFinder with keep_nodes=True")
        tinf, _ = self._model()
        f = self.nodes._f
        rbs_wt = tinf._f64(80, 28)
        cdef double st_wt = tinf.start_weight, no_mot = tinf.missing_motif_weight, rbs1, rbs2
        cdef bint uses_sd = tinf.uses_sd
        # Prodigal's stopcmp_nodes: stop_val ascending, then strand descending, then ndx ascending
        order = np.lexsort((f["ndx"], -f["strand"].astype(np.int32), f["stop_val"]))
        if header:
            n += file.write('# Sequence Data: seqnum=%d;seqlen=%d;seqhdr="%s"\n' % (self._num_seq, len(self.sequence), sequence_id))
            n += file.write("# Run Data: version=pyrodigal_amd.v%s;gc_cont=%.2f;transl_table=%d;uses_sd=%d\n"
                            % (_VERSION, tinf.gc * 100, tinf.translation_table, int(uses_sd)))
            n += file.write("Beg\tEnd\tStd\tTotal\tCodPot\tStrtSc\tCodon\tRBSMot\tSpacer\tRBSScr\tUpsScr\tTypeScr\tGCCont\n")
        prev_stop, prev_strand = -1, 0
        for i in order:
            if f["type"][i] == 3:
                continue
            ndx = int(f["ndx"][i]); stop_val = int(f["stop_val"][i]); strand = int(f["strand"][i])
            if stop_val != prev_stop or strand != prev_strand:
                prev_stop, prev_strand = stop_val, strand
                n += file.write("\n")
            if strand == 1:
                n += file.write("%d\t%d\t+\t" % (ndx + 1, stop_val + 3))
            else:
                n += file.write("%d\t%d\t-\t" % (stop_val - 1, ndx + 1))
            cs = float(f["cscore"][i]); ss = float(f["sscore"][i]); rs = float(f["rscore"][i])
            n += file.write("%.2f\t%.2f\t%.2f\t%s\t" % (cs + ss, cs, ss, ["ATG", "GTG", "TTG", "Edge"][3 if f["edge"][i] else int(f["type"][i])]))
            r0 = int(f["rbs"][i][0]); r1 = int(f["rbs"][i][1])
            rbs1 = rbs_wt[r0] * st_wt; rbs2 = rbs_wt[r1] * st_wt
            mot = float(f["mot_score"][i]) * st_wt
            if uses_sd:
                k = r0 if rbs1 > rbs2 else r1
                n += file.write("%s\t%s\t%.2f\t" % (_RBS_MOTIF[k], _RBS_SPACER[k], rs))
            elif no_mot > -0.5 and rbs1 > rbs2 and rbs1 > mot:
                n += file.write("%s\t%s\t%.2f\t" % (_RBS_MOTIF[r0], _RBS_SPACER[r0], rs))
            elif no_mot > -0.5 and rbs2 >= rbs1 and rbs2 > mot:
                n += file.write("%s\t%s\t%.2f\t" % (_RBS_MOTIF[r1], _RBS_SPACER[r1], rs))
            elif f["mot_len"][i] == 0:
                n += file.write("None\tNone\t%.2f\t" % rs)
            else:
                motif = "".join("AGCT"[(int(f["mot_ndx"][i]) >> (2 * q)) & 3] for q in range(int(f["mot_len"][i])))
                n += file.write("%s\t%dbp\t%.2f\t" % (motif, int(f["mot_spacer"][i]), rs))
            n += file.write("%.2f\t%.2f\t%.3f\n" % (float(f["uscore"][i]), float(f["tscore"][i]), float(f["gc_cont"][i])))
        n += file.write("\n")
        return n


# --- GeneFinder (ref: lib.pyx:5073-5575) ------------------------------------------------------
cdef class _FinderSlot:
    """One device context of a finder: its own HIP stream, scratch buffers and copy of the models."""
    cdef pga_ctx* ctx
    cdef bint busy
    cdef bint models_loaded
    cdef object models_sig      # (id(raw), version) of every model as loaded: a TrainingInfo changed in place is reloaded

    def __cinit__(self):
        self.ctx = NULL
        self.busy = False
        self.models_loaded = False
        self.models_sig = None

    def __dealloc__(self):
        if self.ctx != NULL:
            pga_destroy(self.ctx)
            self.ctx = NULL


cdef class _FindRequest:
    """The sequences of one `find_genes` / `find_genes_batch` call, waiting for a device call to ride."""
    cdef list seqs              # Sequence objects
    cdef bint translate
    cdef ssize_t first_id
    cdef int64_t bases
    cdef list out               # one Genes per sequence, filled in by the thread that ran the device call
    cdef object error
    cdef bint done
    cdef object sem             # a lock used as a binary semaphore (C-level, no Python-side condition variable): released when the result
                                # is there, or when `lead` holds a context for this caller to run the next device call
    cdef bint signaled          # the semaphore was released and the owner has not looked yet (never released twice)
    cdef object lead


cdef object _new_lock = threading.Lock


cdef inline void _signal(_FindRequest r):
    """(finder lock held) Wake the owner of a request: at most one release per look of the owner."""
    if not r.signaled:
        r.signaled = True
        r.sem.release()


cdef class GeneFinder:
    """A configurable gene finder for genomes and metagenomes, running on one MI355X.

    Re-entrant like the reference's (lib.pyx:5424-5446, README "thread-safety"): `find_genes` may be called from any number
    of threads (the reference's own CLI maps it over a thread pool, cli.py:289-302).  Concurrent calls do not queue behind a
    lock: the finder owns up to `contexts` device contexts (one HIP stream each; two by default: with one contig per call more
    contexts only cut the same callers into smaller device calls), and the calls that are waiting when a context
    is free are packed into ONE device call (`pga_find_genes_batch` over all their sequences, at most `coalesce_bases` bases)
    by whichever caller finds the context -- so a pool of threads rides the batch path, and a lone caller pays no wait."""
    cdef readonly bint meta
    cdef readonly bint closed
    cdef readonly bint mask
    cdef readonly int min_mask
    cdef readonly int min_gene
    cdef readonly int min_edge_gene
    cdef readonly int max_overlap
    cdef readonly str backend
    cdef readonly object training_info
    cdef readonly MetagenomicBins metagenomic_bins
    cdef readonly int device
    cdef readonly bint keep_nodes
    cdef readonly int contexts
    cdef readonly int64_t coalesce_bases
    cdef readonly dict stats    # device calls, sequences and the largest number of calls packed into one (diagnostics)
    cdef object lock            # kept for callers that serialise around a finder themselves (ref: lib.pyx:5196)
    cdef object _cv
    cdef object _lock
    cdef list _slots
    cdef list _pending
    cdef ssize_t _num_seq

    def __cinit__(self):
        self._num_seq = 1

    def __init__(self, TrainingInfo training_info=None, *, bint meta=False, MetagenomicBins metagenomic_bins=None,
                 bint closed=False, bint mask=False, int min_mask=50, int min_gene=90, int min_edge_gene=60,
                 int max_overlap=60, str backend="detect", int device=0, bint keep_nodes=True, int contexts=2,
                 int64_t coalesce_bases=64 << 20):
        # argument validation as in the reference (lib.pyx:5169-5181)
        if meta and training_info is not None:
            raise ValueError("cannot use a training info in meta mode.")
        if min_gene <= 0:
            raise ValueError("`min_gene` must be strictly positive")
        if min_edge_gene <= 0:
            raise ValueError("`min_edge_gene` must be strictly positive")
        if min_edge_gene < 4 and not closed:
            raise ValueError("`min_edge_gene` below 4 is not supported with open ends (one-codon edge genes)")
        if min_mask < 0:
            raise ValueError("`min_mask` must be positive")
        if max_overlap < 0:
            raise ValueError("`max_overlap` must be positive")
        elif max_overlap > min_gene:
            raise ValueError("`max_overlap` must be lower than `min_gene`")
        if backend not in ("detect", "hip"):
            raise ValueError("unsupported backend %r: this build only has the HIP (gfx950) backend" % backend)
        if contexts < 1 or contexts > 16:
            raise ValueError("`contexts` must be between 1 and 16")
        self.meta = meta
        self.closed = closed
        self.mask = mask
        self.min_mask = min_mask
        self.min_gene = min_gene
        self.min_edge_gene = min_edge_gene
        self.max_overlap = max_overlap
        self.backend = backend
        self.training_info = training_info
        self.metagenomic_bins = METAGENOMIC_BINS if metagenomic_bins is None else metagenomic_bins
        self.device = device
        self.keep_nodes = keep_nodes
        self.contexts = contexts
        self.coalesce_bases = max(coalesce_bases, 1)
        self.lock = threading.Lock()
        self._lock = threading.Lock()
        self._cv = threading.Condition(self._lock)
        self._slots = [_FinderSlot() for _ in range(contexts)]
        self._pending = []
        self.stats = {"device_calls": 0, "sequences": 0, "max_calls_per_device_call": 0}

    def __reduce__(self):           # ref: lib.pyx:5219-5234
        return _gene_finder_from_state, (self.training_info, dict(
            meta=self.meta, metagenomic_bins=self.metagenomic_bins if self.meta else None, closed=self.closed, mask=self.mask,
            min_mask=self.min_mask, min_gene=self.min_gene, min_edge_gene=self.min_edge_gene, max_overlap=self.max_overlap,
            backend=self.backend, device=self.device, keep_nodes=self.keep_nodes, contexts=self.contexts,
            coalesce_bases=self.coalesce_bases))

    def __repr__(self):
        parts = []
        if self.training_info is not None:
            parts.append("training_info=%r" % self.training_info)
        if self.meta:
            parts.append("meta=True")
        if self.closed:
            parts.append("closed=True")
        return "pyrodigal_amd.lib.GeneFinder(%s)" % ", ".join(parts)

    cdef int _ensure_models(self, _FinderSlot slot) except -1:
        cdef int rc, n, i
        cdef const pga_training** ptrs
        cdef list blobs
        if slot.ctx == NULL:
            rc = pga_create(self.device, &slot.ctx)
            if rc != PGA_OK:
                slot.ctx = NULL
                _raise_for(NULL, rc, "pga_create")
        cdef list tinfs
        if self.meta:
            tinfs = [(<MetagenomicBin> b).training_info for b in self.metagenomic_bins]
        else:
            tinfs = [self.training_info]
        # the reference shares the struct by pointer, so a setter takes effect at the next call: reload when one moved
        cdef tuple sig = tuple([(id((<TrainingInfo> t)._raw), (<TrainingInfo> t)._version) for t in tinfs])
        if slot.models_loaded and sig == slot.models_sig:
            return 0
        blobs = [(<TrainingInfo> t)._raw for t in tinfs]
        n = len(blobs)
        ptrs = <const pga_training**> malloc(sizeof(void*) * max(n, 1))
        if ptrs == NULL:
            raise MemoryError()
        try:
            for i in range(n):
                ptrs[i] = <const pga_training*> <size_t> blobs[i].ctypes.data
            rc = pga_set_models(slot.ctx, ptrs, n)
        finally:
            free(ptrs)
        if rc != PGA_OK:
            _raise_for(slot.ctx, rc, "pga_set_models")
        slot.models_loaded = True
        slot.models_sig = sig
        return 0

    def find_genes(self, object sequence):
        """Find all the genes in the input DNA sequence (ref: lib.pyx:5400-5469)."""
        return self.find_genes_batch([sequence])[0]

    cdef _FinderSlot _free_slot(self):
        # a context that already exists first: a lone caller never makes a second one
        cdef _FinderSlot s, spare = None
        for s in self._slots:
            if not s.busy:
                if s.ctx != NULL:
                    return s
                if spare is None:
                    spare = s
        return spare

    cdef list _take_pending(self):
        """The waiting requests that ride the next device call: in arrival order, same options, up to the base budget."""
        cdef _FindRequest r, head = self._pending[0]
        cdef int64_t bases = 0
        cdef list take = []
        cdef ssize_t k = 0
        while k < len(self._pending):
            r = self._pending[k]
            if r.translate != head.translate or (take and bases + r.bases > self.coalesce_bases):
                break
            take.append(r)
            bases += r.bases
            k += 1
        del self._pending[:k]
        return take

    def find_genes_batch(self, object sequences, *, bint translate=False):
        """`find_genes` for many sequences in one device pass; returns one `Genes` per input, in order.

        `translate=True` also translates every gene on the device while the batch is resident (one thread per codon, the
        translation table of the model that called the gene): `Gene.translate()` with its default arguments and
        `Genes.write_translations` then read those proteins instead of translating codon by codon on the host."""
        if not self.meta and self.training_info is None:
            raise RuntimeError("cannot find genes without having trained in single mode")
        # the reference always re-wraps with the finder's masking rule (ref: lib.pyx:5433-5438); a Sequence that already
        # follows it is used as it is
        cdef list seqs = []
        cdef int64_t bases = 0
        for s in sequences:
            if isinstance(s, Sequence):
                if (<Sequence> s).mask != self.mask or (self.mask and <int> (<Sequence> s).mask_size != self.min_mask):
                    s = Sequence((<Sequence> s).data, mask=self.mask, mask_size=self.min_mask)
            else:
                s = Sequence(s, mask=self.mask, mask_size=self.min_mask)
            seqs.append(s)
            bases += len((<Sequence> s).data)
        if not seqs:
            return []
        cdef _FindRequest req = _FindRequest.__new__(_FindRequest)
        cdef _FindRequest r
        cdef _FinderSlot slot = None
        cdef list take
        req.seqs = seqs; req.translate = translate; req.bases = bases; req.out = None; req.error = None; req.done = False
        req.lead = None
        req.signaled = False
        req.sem = _new_lock()
        req.sem.acquire()
        lock = self._lock
        # Either a context is free: this caller runs a device call right away, for itself and for everyone who is waiting.  Or it
        # waits on a semaphore of its own -- woken when its result is there, or when a context came free and it is this caller's
        # turn to run the next device call over whatever is waiting by then (the baton goes to the oldest waiting request: one
        # wake-up per device call, not one per waiting thread).
        with lock:
            req.first_id = self._num_seq
            self._num_seq += len(seqs)
            self._pending.append(req)
            slot = self._free_slot()
            if slot is not None:
                slot.busy = True
        try:
            while True:
                if slot is None:
                    req.sem.acquire()                        # blocks without the GIL until somebody signals this request
                    with lock:
                        req.signaled = False
                        slot = <_FinderSlot> req.lead        # the baton: a context reserved for this caller (or None: the result is there)
                        req.lead = None
                    if slot is None:
                        if req.done:
                            break
                        continue
                with lock:
                    take = self._take_pending() if self._pending else []
                try:
                    if take:
                        self._run(slot, take)
                finally:
                    # whatever happened to this caller, the requests it took get their wake-up and the context moves on
                    with lock:
                        for r in take:
                            if not r.done and r.out is None and r.error is None:
                                r.error = RuntimeError("the device call this request rode was interrupted in another thread")
                            r.done = True
                            if r is not req:
                                _signal(r)
                        self._release_slot(slot)
                    slot = None
                if req.done:
                    break
        except BaseException:
            # a caller interrupted while it waits (KeyboardInterrupt in `acquire`) leaves nothing behind: its request leaves the queue,
            # a context that was handed to it in the meantime goes to the next waiting request
            with lock:
                if req in self._pending:
                    self._pending.remove(req)
                if slot is None and req.lead is not None:
                    slot = <_FinderSlot> req.lead
                    req.lead = None
                if slot is not None:
                    self._release_slot(slot)
            raise
        if req.error is not None:
            raise req.error
        return req.out

    cdef int _release_slot(self, _FinderSlot slot) except -1:
        """(lock held) The context goes to the oldest waiting request that has no context yet, or back to the pool."""
        cdef _FindRequest r
        for r in self._pending:
            if r.lead is None:
                r.lead = slot
                _signal(r)
                return 0
        slot.busy = False
        self._cv.notify_all()
        return 0

    cdef int _run(self, _FinderSlot slot, list take) except -1:
        """One device call over the sequences of every request in `take`; each request gets its `Genes` (or the error)."""
        cdef _FindRequest r
        cdef list seqs = []
        for r in take:
            seqs.extend(r.seqs)
        cdef bint translate = (<_FindRequest> take[0]).translate
        st = self.stats
        try:
            out = self._device_call(slot, seqs, translate, take)
        except Exception as e:
            if len(take) == 1:
                (<_FindRequest> take[0]).error = e
                return 0
            # A device call that carries the sequences of several callers failed.  The reference's calls have private state: one
            # caller's bad input never fails another's.  So every request rides a device call of its own now, and gets its own
            # result or its own error.
            st["device_calls_retried_per_request"] = st.get("device_calls_retried_per_request", 0) + 1
            for r in take:
                try:
                    r.out = self._device_call(slot, r.seqs, translate, [r])
                    st["device_calls"] += 1
                    st["sequences"] += len(r.seqs)
                except Exception as e1:
                    r.error = e1
            return 0
        # (anything else -- KeyboardInterrupt, SystemExit -- is this thread being stopped: it goes up, and the caller's clean-up tells
        #  the passengers that their device call was interrupted)
        cdef ssize_t k = 0
        for r in take:
            r.out = out[k:k + len(r.seqs)]
            k += len(r.seqs)
        st["device_calls"] += 1
        st["sequences"] += len(seqs)
        if len(take) > st["max_calls_per_device_call"]:
            st["max_calls_per_device_call"] = len(take)
        return 0

    cdef list _device_call(self, _FinderSlot slot, list seqs, bint translate, list take):
        cdef int n = len(seqs), i, j, rc
        cdef const char** ptrs = <const char**> malloc(sizeof(char*) * max(n, 1))
        cdef int64_t* lens = <int64_t*> malloc(sizeof(int64_t) * max(n, 1))
        cdef pga_params p
        cdef pga_result* res = NULL
        cdef pga_batch* batch = NULL
        cdef list out = []
        cdef list ids = []
        cdef Genes genes
        cdef _FindRequest r
        cdef pga_contig_result* cr
        cdef object prot = None, prot_off = None, tables = None
        cdef size_t p_tab, p_off, p_out
        cdef int64_t ng
        cdef pga_ctx* ctx
        if ptrs == NULL or lens == NULL:
            free(ptrs); free(lens)
            raise MemoryError()
        p.closed = self.closed; p.min_gene = self.min_gene; p.min_edge_gene = self.min_edge_gene
        p.max_overlap = self.max_overlap; p.meta = self.meta; p.want_nodes = self.keep_nodes
        p.mask = self.mask; p.min_mask = self.min_mask
        for r in take:
            for j in range(len(r.seqs)):
                ids.append(r.first_id + j)
        try:
            for i in range(n):
                ptrs[i] = PyBytes_AS_STRING((<Sequence> seqs[i]).data)
                lens[i] = len((<Sequence> seqs[i]).data)
            self._ensure_models(slot)
            ctx = slot.ctx
            if not translate:
                with nogil:
                    rc = pga_find_genes_batch(ctx, n, ptrs, lens, &p, &res)
                if rc != PGA_OK:
                    _raise_for(ctx, rc, "pga_find_genes_batch")
            else:
                rc = pga_batch_create(ctx, n, ptrs, lens, &batch)
                if rc != PGA_OK:
                    _raise_for(ctx, rc, "pga_batch_create")
                try:
                    with nogil:
                        rc = pga_find_genes(ctx, batch, &p, &res)
                    if rc != PGA_OK:
                        _raise_for(ctx, rc, "pga_find_genes")
                    prot_off = np.zeros(res.n_genes + 1, np.int64)
                    tables = np.full(max(n, 1), 11, np.int32)
                    for i in range(n):
                        cr = &res.contigs[i]
                        if self.meta:
                            if cr.model >= 0:
                                tables[i] = (<MetagenomicBin> self.metagenomic_bins[cr.model]).training_info.translation_table
                        else:
                            tables[i] = (<TrainingInfo> self.training_info).translation_table
                    for j in range(res.n_genes):
                        prot_off[j + 1] = prot_off[j] + (res.genes[j].end - res.genes[j].begin + 1) // 3
                    prot = np.zeros(max(int(prot_off[res.n_genes]), 1), np.uint8)
                    p_tab = tables.ctypes.data; p_off = prot_off.ctypes.data; p_out = prot.ctypes.data
                    ng = res.n_genes
                    with nogil:
                        rc = pga_translate_genes(ctx, batch, ng, res.genes, <const int32_t*> p_tab, 88, 1, 1,
                                                 <const int64_t*> p_off, <char*> p_out)
                    if rc != PGA_OK:
                        _raise_for(ctx, rc, "pga_translate_genes")
                finally:
                    pga_batch_free(batch)
            for i in range(n):
                cr = &res.contigs[i]
                genes = Genes.__new__(Genes)
                genes.sequence = seqs[i]
                (<Sequence> seqs[i])._gc = cr.gc
                (<Sequence> seqs[i])._unknown = cr.n_unknown
                if (<Sequence> seqs[i])._masks is None:
                    (<Sequence> seqs[i])._masks = []
                    if res.mask_off != NULL:
                        for j in range(res.mask_off[i], res.mask_off[i + 1]):
                            (<Sequence> seqs[i])._masks.append(Mask(res.masks[2 * j], res.masks[2 * j + 1]))
                genes.meta = self.meta
                genes._num_seq = ids[i]
                genes.score = cr.score
                if self.meta:
                    if cr.model >= 0:
                        genes.metagenomic_bin = self.metagenomic_bins[cr.model]
                        genes.training_info = genes.metagenomic_bin.training_info
                    else:
                        genes.metagenomic_bin = genes.training_info = None
                else:
                    genes.metagenomic_bin = None
                    genes.training_info = self.training_info
                genes._nodes = None
                genes._node_blob = None
                genes._node_n = 0
                if self.keep_nodes and res.nodes != NULL:
                    genes._node_blob = _pack_nodes(&res.nodes[i])
                    genes._node_n = res.nodes[i].n
                genes._genes = None
                genes._n = cr.n_genes
                genes._recs = PyBytes_FromStringAndSize(<const char*> &res.genes[cr.gene_begin], cr.n_genes * sizeof(pga_gene)) if cr.n_genes > 0 else b""
                genes._prot = None; genes._prot_off = None; genes._prot_tt = 0
                if prot is not None:
                    genes._prot = prot[prot_off[cr.gene_begin]:prot_off[cr.gene_begin + cr.n_genes]].tobytes()
                    genes._prot_off = (prot_off[cr.gene_begin:cr.gene_begin + cr.n_genes + 1] - prot_off[cr.gene_begin]).copy()
                    genes._prot_tt = int(tables[i])
                out.append(genes)
        finally:
            free(ptrs); free(lens)
            if res != NULL:
                pga_result_free(res)
        return out

    def train(self, object sequence, *sequences, bint force_nonsd=False, double start_weight=4.35, int translation_table=11):
        """Train on the given genome, on the device, and use the result for the next `find_genes` (ref: lib.pyx:5471-5575).

        Several sequences (the contigs of one genome) are joined with `TTAATTAATTAA` linkers like in Prodigal."""
        import warnings
        cdef Sequence seq
        cdef pga_params p
        cdef pga_batch* batch = NULL
        cdef const char* ptr
        cdef int64_t length
        cdef int rc
        cdef object raw
        cdef _FinderSlot slot
        cdef pga_ctx* ctx
        if self.meta:
            raise RuntimeError("cannot use training sequence in metagenomic mode")
        if translation_table not in TRANSLATION_TABLES:
            raise ValueError("%d is not a valid translation table index" % translation_table)
        if isinstance(sequence, Sequence):
            if sequences:
                raise NotImplementedError("cannot use more than one `Sequence` object in `GeneFinder.train`")
            seq = Sequence(sequence, mask=self.mask, mask_size=self.min_mask)
        elif isinstance(sequence, str):
            if sequences:
                sequence = "TTAATTAATTAA".join(list((sequence,) + sequences) + [""])
            seq = Sequence(sequence, mask=self.mask, mask_size=self.min_mask)
        else:
            if sequences:
                sequence = b"TTAATTAATTAA".join([bytes(memoryview(x)) for x in (sequence,) + sequences] + [b""])
            seq = Sequence(sequence, mask=self.mask, mask_size=self.min_mask)
        if len(seq) < MIN_SINGLE_GENOME:
            raise ValueError("sequence must be at least %d characters (%d found)" % (MIN_SINGLE_GENOME, len(seq)))
        elif len(seq) < IDEAL_SINGLE_GENOME:
            warnings.warn("sequence should be at least %d characters (%d found)" % (IDEAL_SINGLE_GENOME, len(seq)))
        p.closed = self.closed; p.min_gene = self.min_gene; p.min_edge_gene = self.min_edge_gene
        p.max_overlap = self.max_overlap; p.meta = 0; p.want_nodes = 0
        p.mask = self.mask; p.min_mask = self.min_mask
        raw = np.zeros(TRAINING_INFO_SIZE, np.uint8)
        cdef size_t out_ptr = raw.ctypes.data
        ptr = PyBytes_AS_STRING(seq.data)
        length = len(seq.data)
        # the training takes a context for itself, like a device call of find_genes
        with self._cv:
            while True:
                slot = self._free_slot()
                if slot is not None:
                    break
                self._cv.wait()
            slot.busy = True
        try:
            if slot.ctx == NULL:
                rc = pga_create(self.device, &slot.ctx)
                if rc != PGA_OK:
                    slot.ctx = NULL
                    _raise_for(NULL, rc, "pga_create")
            ctx = slot.ctx
            slot.models_loaded = False            # the training loads its own partial models into the context
            rc = pga_batch_create(ctx, 1, &ptr, &length, &batch)
            if rc != PGA_OK:
                _raise_for(ctx, rc, "pga_batch_create")
            try:
                with nogil:
                    rc = pga_train(ctx, batch, &p, translation_table, start_weight, force_nonsd, 0, <pga_training*> out_ptr)
                if rc != PGA_OK:
                    _raise_for(ctx, rc, "pga_train")
            finally:
                pga_batch_free(batch)
            tinf = TrainingInfo(raw=raw)
            self.training_info = tinf
        finally:
            with self._lock:
                self._release_slot(slot)
        return tinf


def _gene_finder_from_state(training_info, dict kw):
    return GeneFinder(training_info, **kw)


cdef object _arr(const void* ptr, ssize_t nbytes, object dtype):
    if ptr == NULL or nbytes == 0:
        return np.zeros(0, dtype=dtype)
    return np.frombuffer(PyBytes_FromStringAndSize(<const char*> ptr, nbytes), dtype=dtype)


cdef Nodes _copy_nodes(const pga_nodes* nd):
    cdef Nodes out = Nodes()
    cdef ssize_t n = nd.n
    f = out._f
    f["ndx"] = _arr(nd.ndx, 4 * n, np.int32); f["stop_val"] = _arr(nd.stop_val, 4 * n, np.int32)
    f["traceb"] = _arr(nd.traceb, 4 * n, np.int32); f["tracef"] = _arr(nd.tracef, 4 * n, np.int32)
    f["star_ptr"] = _arr(nd.star_ptr, 12 * n, np.int32).reshape(-1, 3)
    f["type"] = _arr(nd.type, n, np.uint8); f["edge"] = _arr(nd.edge, n, np.uint8); f["elim"] = _arr(nd.elim, n, np.uint8)
    f["rbs"] = _arr(nd.rbs, 2 * n, np.uint8).reshape(-1, 2)
    f["strand"] = _arr(nd.strand, n, np.int8); f["ov_mark"] = _arr(nd.ov_mark, n, np.int8)
    f["gc_cont"] = _arr(nd.gc_cont, 4 * n, np.float32)
    f["cscore"] = _arr(nd.cscore, 8 * n, np.float64); f["sscore"] = _arr(nd.sscore, 8 * n, np.float64)
    f["rscore"] = _arr(nd.rscore, 8 * n, np.float64); f["uscore"] = _arr(nd.uscore, 8 * n, np.float64)
    f["tscore"] = _arr(nd.tscore, 8 * n, np.float64); f["score"] = _arr(nd.score, 8 * n, np.float64)
    f["mot_score"] = _arr(nd.mot_score, 8 * n, np.float64); f["mot_ndx"] = _arr(nd.mot_ndx, 4 * n, np.int32)
    f["mot_len"] = _arr(nd.mot_len, n, np.uint8); f["mot_spacer"] = _arr(nd.mot_spacer, n, np.uint8)
    f["mot_spacendx"] = _arr(nd.mot_spacendx, n, np.uint8)
    return out


# The node arrays of one contig, field after field in one bytes object: what a Genes keeps until `.nodes` is asked for.
_NODE_BLOB_FIELDS = [
    ("ndx", np.int32, 1), ("stop_val", np.int32, 1), ("traceb", np.int32, 1), ("tracef", np.int32, 1), ("star_ptr", np.int32, 3),
    ("gc_cont", np.float32, 1), ("mot_ndx", np.int32, 1),
    ("cscore", np.float64, 1), ("sscore", np.float64, 1), ("rscore", np.float64, 1), ("uscore", np.float64, 1),
    ("tscore", np.float64, 1), ("score", np.float64, 1), ("mot_score", np.float64, 1),
    ("type", np.uint8, 1), ("edge", np.uint8, 1), ("elim", np.uint8, 1), ("rbs", np.uint8, 2), ("strand", np.int8, 1),
    ("ov_mark", np.int8, 1), ("mot_len", np.uint8, 1), ("mot_spacer", np.uint8, 1), ("mot_spacendx", np.uint8, 1),
]


cdef bytes _pack_nodes(const pga_nodes* nd):
    cdef ssize_t n = nd.n
    cdef const void* src[23]
    cdef ssize_t width[23]
    cdef ssize_t k, total = 0, at = 0
    src[0] = nd.ndx; src[1] = nd.stop_val; src[2] = nd.traceb; src[3] = nd.tracef; src[4] = nd.star_ptr
    src[5] = nd.gc_cont; src[6] = nd.mot_ndx
    src[7] = nd.cscore; src[8] = nd.sscore; src[9] = nd.rscore; src[10] = nd.uscore; src[11] = nd.tscore; src[12] = nd.score
    src[13] = nd.mot_score
    src[14] = nd.type; src[15] = nd.edge; src[16] = nd.elim; src[17] = nd.rbs; src[18] = nd.strand; src[19] = nd.ov_mark
    src[20] = nd.mot_len; src[21] = nd.mot_spacer; src[22] = nd.mot_spacendx
    width[0] = 4; width[1] = 4; width[2] = 4; width[3] = 4; width[4] = 12; width[5] = 4; width[6] = 4
    for k in range(7, 14):
        width[k] = 8
    for k in range(14, 23):
        width[k] = 1
    width[17] = 2
    for k in range(23):
        total += width[k] * n
    cdef bytes blob = PyBytes_FromStringAndSize(NULL, total)
    cdef char* dst = PyBytes_AS_STRING(blob)
    for k in range(23):
        if n > 0 and src[k] != NULL:
            memcpy(dst + at, src[k], width[k] * n)
        elif n > 0:
            memset(dst + at, 0, width[k] * n)
        at += width[k] * n
    return blob


cdef Nodes _nodes_from_blob(bytes blob, ssize_t n):
    cdef Nodes out = Nodes()
    cdef ssize_t at = 0
    f = out._f
    for name, dt, mult in _NODE_BLOB_FIELDS:
        a = np.frombuffer(blob, dtype=dt, count=n * mult, offset=at)
        f[name] = a.reshape(-1, mult) if mult > 1 else a
        at += a.nbytes
    return out


def _seq_pointers(list seqs):
    """The C-ABI's argument arrays for a list of ``bytes`` contigs: (addresses as uint64, lengths as int64, total bases).  What
    `_cabi.Batch` hands to ``pga_batch_create`` -- one C loop instead of 6 000 ctypes conversions under the GIL per device call
    (4 - 7 ms of a 6 250-contig call's host time).  The caller keeps ``seqs`` alive while the addresses are in use."""
    cdef ssize_t n = len(seqs), i
    cdef int64_t total = 0
    ptrs = np.empty(max(n, 1), dtype=np.uint64)
    lens = np.empty(max(n, 1), dtype=np.int64)
    cdef unsigned long long[::1] p = ptrs
    cdef int64_t[::1] l = lens
    cdef object o
    for i in range(n):
        o = seqs[i]
        if not isinstance(o, bytes):
            raise TypeError("contigs must be bytes")
        p[i] = <unsigned long long> <size_t> PyBytes_AS_STRING(o)
        l[i] = len(<bytes> o)
        total += l[i]
    return ptrs, lens, total
