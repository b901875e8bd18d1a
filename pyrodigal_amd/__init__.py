"""MI355X-native Prodigal gene-finding core behind pyrodigal's GeneFinder / Genes API.

``pyrodigal_amd.lib`` (Cython, built by ``__graft_entry__.build()``) mirrors the reference's
``pyrodigal.lib``; ``pyrodigal_amd._cabi`` is the raw ctypes view of the C-ABI.

The names of ``pyrodigal_amd.lib`` are re-exported lazily: importing the package (or its pure-Python helpers such as
``pyrodigal_amd.benchdata``) does not load the HIP runtime; the first use of ``GeneFinder`` & co. does.
"""
__version__ = "0.1.0"

_LIB_NAMES = ("GeneFinder", "Genes", "Gene", "Nodes", "Node", "Sequence", "TrainingInfo", "MetagenomicBin", "MetagenomicBins",
              "ConnectionScorer", "Mask", "METAGENOMIC_BINS", "TRANSLATION_TABLES", "PRODIGAL_VERSION", "MIN_SINGLE_GENOME",
              "IDEAL_SINGLE_GENOME")
__all__ = list(_LIB_NAMES)


def __getattr__(name):
    if name in _LIB_NAMES or name == "lib":
        import importlib
        try:
            lib = importlib.import_module(".lib", __name__)
        except ImportError as e:      # extension not built yet: `python -c "import __graft_entry__ as g; g.build()"`
            raise AttributeError("pyrodigal_amd.%s needs the Cython extension (%s)" % (name, e)) from e
        return lib if name == "lib" else getattr(lib, name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
