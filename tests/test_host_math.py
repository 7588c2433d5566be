"""CPU tier: the arithmetic header the HIP kernels use (gndt_math.hpp), driven through a host shim and
numpy data movement, must reproduce the oracle: keys, counts, order, labels exact; statistics within
the stated tolerances.  This is the order-free restatement of isSlope the kernels rely on."""
import numpy as np
import pytest

from tests import host_emulation as he
from tests import parity, scenes

CASES = {
    "bridge_ground": (lambda: scenes.bridge_ground(), scenes.BRIDGE_PARAMS),
    "campus_slope": (lambda: scenes.campus_frame(80000), scenes.CAMPUS_PARAMS),
    "uniform_cubic": (lambda: scenes.uniform_box(120000), dict(grid_len=0.5, z_len=0.5, slope_interval=0.08, demand="slope")),
    "terrain_true": (lambda: scenes.terrain_cloud(150000), dict(grid_len=0.2, z_len=0.1, slope_interval=0.08, demand="true")),
}


@pytest.mark.parametrize("name", list(CASES))
def test_emulated_pipeline_matches_oracle(name):
    make, P = CASES[name]
    cloud = make()
    ref = parity.ref_from_cloud(cloud, P)
    emu = he.build(cloud, P["grid_len"], P["z_len"], P["slope_interval"], P["demand"])
    parity.assert_parity(emu, ref)


def test_shard_additivity_and_first_idx_min():
    cloud = scenes.campus_frame(40000)
    P = scenes.CAMPUS_PARAMS
    body, o = cloud[1:], cloud[0]
    whole = he.accumulate(body, o, P["grid_len"], P["z_len"])
    cut = 17001
    a = he.accumulate(body[:cut], o, P["grid_len"], P["z_len"], 0)
    b = he.accumulate(body[cut:], o, P["grid_len"], P["z_len"], cut)
    keys = np.union1d(a[0], b[0])
    assert np.array_equal(keys, whole[0])
    cnt = np.zeros(keys.size, np.int64)
    first = np.full(keys.size, 2**40, np.int64)
    sums = np.zeros((keys.size, 9))
    for part in (a, b):
        pos = np.searchsorted(keys, part[0])
        cnt[pos] += part[1]
        first[pos] = np.minimum(first[pos], part[2])
        sums[pos] += part[3]
    assert np.array_equal(cnt, whole[1]) and np.array_equal(first, whole[2])
    assert np.allclose(sums, whole[3], rtol=1e-12, atol=1e-12)
