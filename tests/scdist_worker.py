"""Worker of tests/test_scdist_cpu.py: one rank of a world_size-N gloo job running a golden single-cell
case through te_counter_amd.scdist.run_parts with the numpy stand-in engine."""

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from conftest import load_golden  # noqa: E402
from te_counter_amd.dist import World  # noqa: E402
import test_scdist_cpu as T  # noqa: E402


def main():
    g = load_golden(sys.argv[1])
    w = World.from_env(backend='gloo')
    n = len(g['records'])
    # uneven cuts: the first rank gets 37 % of the records, the rest is split evenly
    first = int(n * 0.37)
    cuts = [0] + [first + (n - first) * r // (w.size - 1) for r in range(w.size)]
    out, result, barcodes = T.run_rank(g, w, cuts)
    if w.rank == 0:
        T.check(g, out, result, barcodes)
        print('OK %d' % w.size, out['stats']['bundles'], len(barcodes))
    w.close()


if __name__ == '__main__':
    main()
