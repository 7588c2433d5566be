"""GPU tier: BLOCKED buckets (gndt_blocked.hpp, round 6) — spatial blocks of 512 nodes as buckets and a bucket kernel that addresses
its table by the key, taken for clouds whose map is a dense, evenly filled box (what the previous build on the handle found) and
abandoned, the build re-run with hashed buckets, when a cloud does not fit the box.  Whatever the buckets, the map is the oracle's."""
import numpy as np
import pytest

from grid_ndt_amd import scenes
from tests import parity

pytestmark = pytest.mark.gpu


def _handle(P, **kw):
    import grid_ndt_amd as g
    m = g.TwoDmap(P["grid_len"], P["z_len"], **kw)
    m.setInterval(P["slope_interval"])
    return m


def _dev(cloud):
    import torch
    return torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda()


def _ref(cloud, P):
    return parity.ref_from_cloud(cloud, P, mode=2)


@pytest.mark.parametrize("demand", ["slope", "true"])
def test_dense_box_takes_blocked_buckets_and_gives_the_oracles_map(demand):
    P = dict(grid_len=0.5, z_len=0.5, slope_interval=0.08, demand=demand)
    cloud = scenes.uniform_box(2_500_001, half_xy=50.0)
    ref = _ref(cloud, P)
    m = _handle(P)
    m.setCloudFirst(cloud[0])
    t = _dev(cloud)
    names = []
    for k in range(3):
        m.create2DMap(demand, t)
        out = m.export()
        names.append(m.STRATEGY_NAMES[m.last_strategy()])
        rep = parity.compare(out, ref, demand)
        assert rep["ok"], (k, names, rep["fail"])
    assert names == ["partition", "partition_blocked", "partition_blocked"] and m.retry_count() == 0, names
    # another cloud of the same scene (other noise: a few points beyond the first cloud's extent — the box has a block of margin)
    other = scenes.uniform_box(2_500_001, seed=0x5EED0777, half_xy=50.0)
    other[0] = cloud[0]
    m.create2DMap(demand, _dev(other))
    out = m.export()
    assert m.STRATEGY_NAMES[m.last_strategy()] == "partition_blocked" and m.retry_count() == 0
    rep = parity.compare(out, _ref(other, P), demand)
    assert rep["ok"], rep["fail"]


def test_cloud_that_leaves_the_box_is_rebuilt_with_hashed_buckets():
    P = dict(grid_len=0.5, z_len=0.5, slope_interval=0.08)
    cloud = scenes.uniform_box(2_500_001, half_xy=50.0)
    m = _handle(P)
    m.setCloudFirst(cloud[0])
    t = _dev(cloud)
    for _ in range(2):
        m.create2DMap("slope", t)
        m.sync()
    assert m.last_strategy() == 7
    wide = cloud.copy()
    wide[1:, :2] *= np.float32(1.5)                       # same size, 1.5 x the extent: most points outside the box
    m.create2DMap("slope", _dev(wide))
    out = m.export()
    assert m.last_strategy() != 7 and m.retry_count() >= 1
    rep = parity.compare(out, _ref(wide, P))
    assert rep["ok"], rep["fail"]
    before = m.retry_count()
    m.create2DMap("slope", _dev(wide))                    # the handle has forgotten the box: hashed from the start, no re-run
    out = m.export()
    assert m.last_strategy() == 2 and m.retry_count() == before
    assert parity.compare(out, _ref(wide, P))["ok"]


def test_tall_or_uneven_maps_keep_hashed_buckets():
    """Blocks are for dense, evenly filled boxes: a map more than 32 levels high, a map with one node that holds a block's worth of
    points (the reference's (0,0,0) padding) and a sparse map stay with hashed buckets — and are the oracle's maps."""
    cases = []
    tall = scenes.uniform_box(2_000_001, half_xy=50.0)
    tall[1:, 2] *= np.float32(12.0)                       # z in [-12, 12): 48 levels of 0.5 m
    cases.append(("tall", tall, dict(grid_len=0.5, z_len=0.5, slope_interval=0.08)))
    padded = scenes.uniform_box(2_000_001, half_xy=50.0)
    padded[-300_000:] = 0.0                               # 15 % of the points in one node
    cases.append(("padded", padded, dict(grid_len=0.5, z_len=0.5, slope_interval=0.08)))
    sparse = scenes.uniform_box(1_300_001, half_xy=100.0)  # 8 points per column
    cases.append(("sparse", sparse, dict(grid_len=0.5, z_len=0.5, slope_interval=0.08)))
    for name, cloud, P in cases:
        m = _handle(P)
        m.setCloudFirst(cloud[0])
        t = _dev(cloud)
        for _ in range(3):
            m.create2DMap("slope", t)
            m.sync()
            assert m.last_strategy() != 7, name
        rep = parity.compare(m.export(), _ref(cloud, P), "slope", dense=(name == "padded"), interval=P["slope_interval"])
        assert rep["ok"], (name, rep["fail"])
