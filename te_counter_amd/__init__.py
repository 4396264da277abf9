"""te_counter_amd: an MI355X-native (gfx950, HIP) drop-in for te_count's read-vs-annotation
counting path.  The compute lives in libtecount.so (csrc/, include/tecount.h); this package is
the host-side mirror of the reference's `measureTE` interface."""

from . import _lib
from .index import FlatIndex
from .engine import Engine
from .measure import measureTE

__all__ = ['FlatIndex', 'Engine', 'measureTE', '_lib']
