"""MI355X-native Prodigal gene-finding core behind pyrodigal's GeneFinder / Genes API."""
from . import _cabi  # noqa: F401

__version__ = "0.1.0"
