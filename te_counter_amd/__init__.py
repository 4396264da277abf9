"""te_counter_amd: an MI355X-native (gfx950, HIP) drop-in for te_count's read-vs-annotation
counting path.  The compute lives in libtecount.so (csrc/, include/tecount.h); this package is
the host-side mirror of the reference's `measureTE` interface."""

import os as _os

# The on-GPU BAM decoder keeps the inflate launches of several chunks in flight, each on a stream of its own; with four
# hardware queues per process (ROCm's default) the library's streams share queues and those launches run one after the
# other.  Raised to sixteen unless the caller asked for more.  Read by the HIP runtime when it initialises, so it has to be
# in the environment before the first GPU call of the process.
if int(_os.environ.get('GPU_MAX_HW_QUEUES') or 0) < 16:
    _os.environ['GPU_MAX_HW_QUEUES'] = '16'

from . import _lib  # noqa: E402
from .index import FlatIndex  # noqa: E402
from .engine import Engine  # noqa: E402
from .measure import measureTE  # noqa: E402

__all__ = ['FlatIndex', 'Engine', 'measureTE', '_lib']
