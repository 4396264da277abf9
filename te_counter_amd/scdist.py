"""Single-cell counting over one or several GPUs: the steps between "every rank has filtered its own
records" and "the cell x feature counts are on rank 0".

The reference (te_count/te_count.py:298-707) is one sequential pass whose result depends on file order
in three places: where the (CB,UB) dict is flushed to a bundle (every 1e7 keys, :377-390), which
barcodes make the cut (best maxcells + 1000 by count, ties in first-seen order, :502) and which lines
the bundle reader drops (:509-543).  Here cells are dealt to the GPUs by barcode id, so every key and
every cell row is local to one GPU, and exactly those three things are agreed on globally:

    record ordinals   base = records in the ranks before this one            (one gather)
    survivors         all-to-all by barcode id % world, HBM to HBM               (RCCL point-to-point)
    bundle boundaries the ordinal whose global rank among "records that open a new key since the last
                      flush" is bundle_keys, by bisection over per-GPU rank queries (all-reduce of one
                      integer per probe)
    selection         gather of (barcode, count, first ordinal), same sort on every rank
    drop rule         OR of the per-GPU (bundle, barcode) presence tables
    results           disjoint cells: gathered, not summed

`eng` is te_counter_amd.engine.Engine (or, in the CPU tests, an engine with the same methods working on
numpy arrays); with one rank every collective is the identity and the same code path runs.
"""

import numpy as np

from . import dist as _dist

NONE64 = 2 ** 64 - 1


def run_parts(eng, world, n_records_local, bundle_keys, maxcells, pad):
    """-> dict(total_records, raw (ids, counts, first) of all cells, selected, kept_lines, cells, cell_hits,
    coo_cell, coo_name, coo_count, stats) -- identical on every rank."""
    counts = world.gather(int(n_records_local))
    base, total = sum(counts[:world.rank]), sum(counts)
    eng.sc_set_record_base(base, total)
    if world.size > 1:  # deal the survivors to the GPUs by barcode id
        if hasattr(eng, 'sc_deal'):  # in HBM: grouped by a kernel, moved GPU to GPU, ordered on arrival
            _dist.exchange_survivors(world, eng)
        else:  # (the numpy stand-in engine of the CPU tests)
            rows = eng.sc_survivors()
            dest = rows['bcid'] % np.uint32(world.size)
            parts = [{k: v[dest == r] for k, v in rows.items()} for r in range(world.size)]
            eng.sc_survivors_replace(_dist.exchange_rows(world, parts))
    n_keys = eng.sc_part1_sort()
    # the reference dies at the first record (file order) that lacks a barcode / UMI tag (te_count.py:409, 426)
    missing = world.gather(eng.sc_missing_tags())
    no_cb, no_ub = min(m[0] for m in missing), min(m[1] for m in missing)
    if no_cb != NONE64 or no_ub != NONE64:
        raise AssertionError('CB or CR tag not found!' if no_cb < no_ub else 'UB or UR tag not found!')
    if min(world.gather(eng.sc_long_umi())) != NONE64:  # (a limit of this library, not of the reference)
        raise ValueError('UMI longer than 24 bytes is not supported')
    # ---- bundle boundaries
    bounds, lower = [], -1
    if _dist.world_sum(world, n_keys) >= bundle_keys:
        while True:
            if _dist.world_sum(world, eng.sc_epoch_marks(lower)) < bundle_keys:
                break
            if world.size == 1:
                b = eng.sc_epoch_kth(bundle_keys)
            else:  # smallest ordinal v with at least bundle_keys marks <= v over all GPUs
                lo, hi = lower + 1, total - 1
                while lo < hi:
                    mid = (lo + hi) // 2
                    if _dist.world_sum(world, eng.sc_epoch_rank(mid)) >= bundle_keys:
                        hi = mid
                    else:
                        lo = mid + 1
                b = lo
            bounds.append(b)
            lower = b
            if len(bounds) > total // max(1, bundle_keys) + 2:
                raise RuntimeError('bundle boundary search does not terminate')
    ids, cnt, first = eng.sc_part1_finish(bounds)
    # ---- selection (te_count.py:502): best maxcells + pad raw counts, ties first-seen first
    ids = np.concatenate(_dist.all_gather_array(world, ids))
    cnt = np.concatenate(_dist.all_gather_array(world, cnt))
    first = np.concatenate(_dist.all_gather_array(world, first))
    order = np.lexsort((first, -cnt.astype(np.int64)))
    selected = np.sort(ids[order[0:maxcells + pad]]).astype(np.uint32)
    # ---- parts 2 and 3
    if world.size > 1:
        n_bundles = len(bounds) + 1
        present = _dist.world_max_u8(world, eng.sc_presence(n_bundles))
        out = eng.sc_part23(selected, n_bundles, present)
    else:
        out = eng.sc_part23(selected)
    st = eng.sc_stats()
    key_bits = eng.sc_sort_key_bits() if hasattr(eng, 'sc_sort_key_bits') else None
    eng.sc_end()
    # ---- merge (cells are disjoint over the ranks)
    res = {'total_records': total, 'raw_ids': ids, 'raw_counts': cnt, 'selected': selected,
           'kept_lines': _dist.world_sum(world, out['kept_lines'])}
    if world.size == 1:  # one GPU's triplets come sorted by (cell, name) already
        res['cells'], res['cell_hits'] = out['cells'], out['cell_hits']
        res['coo_cell'], res['coo_name'], res['coo_count'] = out['coo_cell'], out['coo_name'], out['coo_count']
    else:
        cells = np.concatenate(_dist.all_gather_array(world, out['cells']))
        hits = np.concatenate(_dist.all_gather_array(world, out['cell_hits']))
        o = np.argsort(cells, kind='stable')
        res['cells'], res['cell_hits'] = cells[o], hits[o]
        cc = np.concatenate(_dist.all_gather_array(world, out['coo_cell']))
        cn = np.concatenate(_dist.all_gather_array(world, out['coo_name']))
        cv = np.concatenate(_dist.all_gather_array(world, out['coo_count']))
        o = np.lexsort((cn, cc))
        res['coo_cell'], res['coo_name'], res['coo_count'] = cc[o], cn[o], cv[o]
    keys = [k for k in st if k not in ('keys', 'bundles')]
    summed = {k: _dist.world_sum(world, st[k]) for k in keys}
    summed['keys'] = _dist.world_sum(world, st['keys'])
    summed['bundles'] = len(bounds) + 1 if summed['survivors'] else 0
    summed['kept_lines'] = res['kept_lines']
    summed['sort_key_bits'] = key_bits  # this rank's: how part 1 ordered its survivors (0 = column by column)
    res['stats'] = summed
    return res
