"""meilisearch_amd — MI355X-native query-time scoring path for Meilisearch's milli.

The product is libmsi.so (hand-written HIP for gfx950 behind the C ABI of
include/msi.h).  This package is the thin Python binding used by the tests and
bench.py; it never falls back to a CPU implementation: importing the kernels on
a machine without the built library or without a gfx950 device raises.
"""
from ._lib import MsiError, abi_version, lib, lib_path  # noqa: F401
from .device import Context, DeviceBuffer  # noqa: F401
from .vector_store import GpuBqStore, GpuStore, VectorStore, dense_filter  # noqa: F401
from .typo import GpuDictionary, number_of_typos_allowed, pack_queries  # noqa: F401
from .bits import BitsPool, DocKeys, DocValues, FacetKeys, GeoPoints, facet_number_key  # noqa: F401
from . import scoring  # noqa: F401
from . import ranking  # noqa: F401
