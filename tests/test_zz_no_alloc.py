"""tec_bulk_count_noalloc (include/tecount.h): the one-shot bulk count that allocates nothing, against the regular
entry points (tec_bulk_begin / tec_bulk_submit_device / tec_bulk_finish) on the same device-resident batches --
bit-exact, two calls in a row, the second with a smaller input (the scratch array keeps the first call's contents)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _device_units(units):
    import torch
    dev = [torch.from_numpy(u.view(np.int16) if u.dtype == np.uint16 else u).cuda() for u in units]
    torch.cuda.synchronize()
    return dev


def test_noalloc_equals_the_regular_entry_points_twice_in_a_row():
    import torch
    from te_counter_amd import Engine, synth
    from oracle import c_oracle
    ann = synth.make_annotation('hg38', scale=0.05)
    eng = Engine(0)
    eng.upload_index(ann.flat)
    n_names = eng.n_names
    out_counts = torch.zeros(n_names, dtype=torch.int64, device='cuda')
    out_stats = torch.zeros(8, dtype=torch.int64, device='cuda')
    ci = c_oracle.COracleIndex.from_flat(ann.flat)
    for n_units, seed, paired in ((3_000_000, 5, True), (70_001, 6, False)):
        units = synth.make_units(ann, n_units, seed=seed, paired=paired)
        dev = _device_units(units)
        ptrs = [d.data_ptr() for d in dev]
        eng.bulk_begin(20)
        eng.bulk_submit_device(n_units, *ptrs)
        want, wst = eng.bulk_finish()
        out_counts.fill_(-7)  # the call overwrites, it does not add
        out_stats.fill_(-7)
        torch.cuda.synchronize()
        eng.bulk_count_noalloc(n_units, *ptrs, 20, out_counts.data_ptr(), out_stats.data_ptr())
        eng.bulk_sync()  # (the call queued its kernels on the context's compute stream)
        got = out_counts.cpu().numpy().view(np.uint64)
        gst = out_stats.cpu().numpy()
        assert np.array_equal(got[:len(want)], want), 'no-allocation entry point differs from the regular one'
        assert [int(x) for x in gst[:5]] == [wst[k] for k in ('units', 'assigned', 'lowq', 'badchrom', 'qcfail')]
        assert int(gst[5]) == 0
        ref, rst = ci.bulk_count(*units, 20)
        assert np.array_equal(got[:len(ref)], ref) and wst['assigned'] == rst['assigned'] > 0
    eng.close()


def test_noalloc_refuses_what_does_not_fit():
    import torch
    from te_counter_amd import Engine, _lib, synth
    ann = synth.make_custom_annotation(scale=0.02)  # 60 k unique names: more than the fixed scratch array holds
    eng = Engine(0)
    eng.upload_index(ann.flat)
    assert eng.n_names > 49152
    buf = torch.zeros(eng.n_names, dtype=torch.int64, device='cuda')
    z = torch.zeros(16, dtype=torch.int32, device='cuda')
    with pytest.raises(_lib.TecError) as ei:
        eng.bulk_count_noalloc(4, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 20, buf.data_ptr())
    assert ei.value.code == _lib.TEC_EOVERFLOW and '49152' in str(ei.value)
    eng.close()
