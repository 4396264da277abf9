"""te_counter_amd: an MI355X-native (gfx950, HIP) drop-in for te_count's read-vs-annotation
counting path.  The compute lives in libtecount.so (csrc/, include/tecount.h); this package is
the host-side mirror of the reference's `measureTE` interface."""

import os as _os

# The on-GPU BAM decoder keeps two inflate launches in flight on two streams; with ROCm's default of four hardware
# queues per process two of the library's streams can land on one queue and run one after the other.  Read by the HIP
# runtime when it initialises, so it has to be in the environment before the first GPU call of the process.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

from . import _lib  # noqa: E402
from .index import FlatIndex  # noqa: E402
from .engine import Engine  # noqa: E402
from .measure import measureTE  # noqa: E402

__all__ = ['FlatIndex', 'Engine', 'measureTE', '_lib']
