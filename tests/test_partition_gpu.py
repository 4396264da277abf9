"""The experimental partitioned pipeline (TEC_PARTITION, csrc/bulk.hip: window histogram, scan,
scatter, count over items) must give the direct kernel's counts and stats bit for bit."""

import numpy as np
import pytest

from oracle import c_oracle
from te_counter_amd import Engine, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('paired,order', [(True, 'shuffled'), (False, 'sorted')])
def test_partitioned_equals_direct_and_oracle(paired, order, monkeypatch):
    ann = synth.make_annotation('hg38', scale=0.1)
    units = synth.make_units(ann, 3_000_000, seed=77, paired=paired, order=order)
    # boundary-heavy tail: positions on and next to 10 kb multiples
    k = 50_000
    units[1][-k:] = (np.arange(k) % 2000) * 10000 + (np.arange(k) % 5) - 2
    units[2][-k:] = units[1][-k:] + np.where(np.arange(k) % 3 == 0, 9999 - (np.arange(k) % 7), 150)
    want, wst = c_oracle.COracleIndex.from_flat(ann.flat).bulk_count(*[u[:1_000_000] for u in units], 20)
    res = {}
    for mode in ('0', '2'):
        monkeypatch.setenv('TEC_PARTITION', mode)
        eng = Engine(0)
        eng.upload_index(ann.flat)
        eng.bulk_begin(20)
        eng.bulk_submit(*[u[:1_000_000] for u in units])
        got, gst = eng.bulk_finish()
        assert np.array_equal(got, want) and all(gst[key] == v for key, v in wst.items()), mode
        eng.bulk_begin(20)
        for a in range(0, 3_000_000, 1_100_000):  # three launches, the last one short
            eng.bulk_submit(*[u[a:a + 1_100_000] for u in units])
        res[mode] = eng.bulk_finish()
        eng.close()
    assert np.array_equal(res['0'][0], res['2'][0])
    assert res['0'][1] == res['2'][1]
