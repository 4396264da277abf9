import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import numpy as np
import bench_sc
from te_counter_amd import Engine, synth
ann = synth.make_annotation('hg38', scale=0.05)
eng = Engine(0); eng.upload_index(ann.flat)
arrays, wl, tables = synth.make_sc_arrays(ann, 3_000_000, n_cells=1500, n_whitelist=40000, seed=5, order='sorted')
ref = None
for rep in range(16):
    out, st, _, (ids, counts) = bench_sc.run(eng, arrays, wl, tables, 1000, False, 250_000, batch=1 << 20)
    cur = {'st': st, 'ids': ids, 'counts': counts, **{k: out[k] for k in ('cells', 'cell_hits', 'coo_cell', 'coo_name', 'coo_count')}, 'kept': out['kept_lines']}
    if ref is None: ref = cur; continue
    diffs = [k for k in cur if (cur[k] != ref[k] if not isinstance(cur[k], np.ndarray) else (cur[k].shape != ref[k].shape or not np.array_equal(cur[k], ref[k])))]
    print(rep, diffs, st['assigned'], ref['st']['assigned'], len(cur['coo_cell']), len(ref['coo_cell']), int(cur['coo_count'].sum()), int(ref['coo_count'].sum()), int(cur['cell_hits'].sum()), int(ref['cell_hits'].sum()))
