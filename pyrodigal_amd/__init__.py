"""MI355X-native Prodigal gene-finding core behind pyrodigal's GeneFinder / Genes API.

``pyrodigal_amd.lib`` (Cython, built by ``__graft_entry__.build()``) mirrors the reference's
``pyrodigal.lib``; ``pyrodigal_amd._cabi`` is the raw ctypes view of the C-ABI.
"""
from . import _cabi  # noqa: F401

__version__ = "0.1.0"

try:
    from .lib import (  # noqa: F401
        GeneFinder, Genes, Gene, Nodes, Node, Sequence, TrainingInfo, MetagenomicBin, MetagenomicBins,
        METAGENOMIC_BINS, TRANSLATION_TABLES, PRODIGAL_VERSION, MIN_SINGLE_GENOME, IDEAL_SINGLE_GENOME,
    )
except ImportError as _e:      # extension not built yet: `python -c "import __graft_entry__ as g; g.build()"`
    _lib_import_error = _e
