"""GPU tier: eager PARTITION builds without a k_part_clear launch (round 6, FoldClear in gndt_partition.hpp / Part::cursors_alt).

From the second eager one-level build (a frame of < 1 M points) of a handle on, the level-1 kernel of a build zeroes what the NEXT build needs zeroed before it starts
(the other set of cursors / partition counters) and what THIS build needs zeroed before its bucket kernel (bitmap, word weights,
Counters).  What can go wrong is state: a set that is not clean when it is taken, an error count that is lost or counted twice,
another strategy or a recorded build in between.  Every build here is compared with the oracle."""
import numpy as np
import pytest

from grid_ndt_amd import scenes
from tests import parity

pytestmark = pytest.mark.gpu

P = dict(grid_len=0.5, z_len=0.5, slope_interval=0.08, demand="slope")


def _handle(strategy=2, **kw):
    import grid_ndt_amd as g
    m = g.TwoDmap(P["grid_len"], P["z_len"], strategy=strategy, **kw)
    m.setInterval(P["slope_interval"])
    return m


def _dev(cloud):
    import torch
    return torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda()


def _check(m, cloud, what):
    out = m.export()
    rep = parity.compare(out, parity.ref_from_cloud(cloud, P, mode=2), "slope")
    assert rep["ok"], (what, rep["fail"])


def test_builds_of_changing_size_on_one_handle():
    """one-level, exact and two-level partitions, growing and shrinking clouds, the same cloud again: both sets of cursors are taken in turn"""
    m = _handle()
    first = scenes.uniform_box(300_001, half_xy=20.0)
    m.setCloudFirst(first[0])
    # (half_xy 20: a few hundred buckets — the one-level partition; 60: more buckets than one level writes — counting partition below
    #  a million points, two levels above)
    cases = [(300_001, 20.0), (300_001, 20.0), (300_001, 60.0), (1_600_001, 60.0), (200_001, 20.0), (200_001, 20.0), (1_600_001, 20.0),
             (250_001, 20.0), (2_400_001, 60.0), (90_001, 20.0), (300_001, 20.0), (300_001, 20.0)]
    used = []
    for k, (n, half) in enumerate(cases):
        cloud = scenes.uniform_box(n, seed=0x1000 + k, half_xy=half)
        cloud[0] = first[0]
        m.create2DMap("slope", _dev(cloud))
        _check(m, cloud, (k, n, half))
        used.append(m.last_strategy())
    assert used.count(6) >= 6 and (2 in used or 7 in used), used


def test_campus_frames_in_a_row():
    """the reference's own workload (a 200 k-point frame per callback) — ten frames, every one the oracle's map"""
    import grid_ndt_amd as g
    CP = scenes.CAMPUS_PARAMS
    m = g.TwoDmap(CP["grid_len"], CP["z_len"])
    m.setInterval(CP["slope_interval"])
    frames = [scenes.campus_frame(200000 - 1000 * k, seed=0x5EED0001 + k) for k in range(5)]
    m.setCloudFirst(frames[0][0])
    for k in range(10):
        c = frames[k % 5].copy()
        c[0] = frames[0][0]
        m.create2DMap("slope", _dev(c))
        out = m.export()
        rep = parity.compare(out, parity.ref_from_cloud(c, CP, mode=2), "slope")
        assert rep["ok"], (k, m.last_strategy(), rep["fail"])
    assert m.last_strategy() == 6


def test_points_outside_the_key_range_are_counted_once_per_build():
    import grid_ndt_amd as g
    m = _handle()
    cloud = scenes.uniform_box(500_001, half_xy=25.0)
    m.setCloudFirst(cloud[0])
    good = _dev(cloud)
    bad = cloud.copy()
    bad[[7, 70_000, 499_999]] = np.float32([0.5 * 70000, 1.0, 1.0])     # |nx| > 65535 (Stopwatch.h:102-110)
    bad_t = _dev(bad)
    for k in range(2):
        m.create2DMap("slope", good)
        _check(m, cloud, ("good", k))
    assert m.last_strategy() == 6
    for k in range(2):            # (twice: the second build's count must not carry the first's)
        with pytest.raises(g.GndtError) as e:
            m.create2DMap("slope", bad_t)
            m.sync()
        assert e.value.code == 4 and " 3 point(s) outside" in str(e.value), str(e.value)
    for k in range(2):
        m.create2DMap("slope", good)
        _check(m, cloud, ("good again", k))


def test_other_strategies_between_partition_builds():
    """ATOMIC builds + updates and TILE builds share the bitmap, the Counters and the partition counters with PARTITION builds"""
    cloud = scenes.uniform_box(350_001, half_xy=22.0)
    frame = scenes.uniform_box(50_001, seed=0x77, half_xy=22.0)[1:]
    import grid_ndt_amd as g
    import torch
    maps = {s: _handle(strategy=s) for s in (2, 1, 5)}
    for m in maps.values():
        m.setCloudFirst(cloud[0])
    t = _dev(cloud)
    whole = np.concatenate([cloud, frame], 0)
    for rnd in range(3):
        for s, m in maps.items():
            m.create2DMap("slope", t)
            _check(m, cloud, (rnd, s))
        a = maps[1]
        a.change2DMap("slope", torch.from_numpy(frame).cuda())
        out = a.export()
        rep = parity.compare(out, parity.ref_from_cloud(whole, P, mode=2), "slope")
        assert rep["ok"], (rnd, "update", rep["fail"])
    # AUTO: a depth-camera frame (TILE) and a sparse cloud (PARTITION) in turn on ONE handle
    depth = scenes.depth_frame()
    DP = scenes.DEPTH_PARAMS
    auto = g.TwoDmap(DP["grid_len"], DP["z_len"])
    auto.setInterval(DP["slope_interval"])
    auto.setCloudFirst(depth[0])
    sparse = scenes.uniform_box(300_001, seed=0x55, half_xy=8.0)
    sparse[0] = depth[0]
    used = set()
    for rnd in range(3):
        for c in (depth, sparse, sparse):
            auto.create2DMap("slope", _dev(c))
            out = auto.export()
            used.add(auto.last_strategy())
            rep = parity.compare(out, parity.ref_from_cloud(c, DP, mode=2), DP.get("demand", "slope"))
            assert rep["ok"], (rnd, auto.last_strategy(), rep["fail"])
    assert 5 in used and used & {2, 3, 6, 7}, used


def test_recorded_build_switches_the_folding_off_for_good():
    """a handle that ever recorded a hipGraph keeps k_part_clear in front of every build: a replay works on the set it was recorded with"""
    import torch
    import grid_ndt_amd as g
    m = _handle(max_nodes_hint=400_000, max_points_hint=700_000)
    cloud = scenes.uniform_box(600_001, half_xy=28.0)
    m.setCloudFirst(cloud[0])
    t = _dev(cloud)
    for _ in range(3):
        m.create2DMap("slope", t)
        m.sync()
    _check(m, cloud, "eager")
    graph = torch.cuda.CUDAGraph()
    other = scenes.uniform_box(600_001, seed=0x99, half_xy=28.0)
    other[0] = cloud[0]
    buf = t.clone()
    with g.graph_capture(graph):
        m.create2DMap("slope", buf)
    for k in range(3):
        buf.copy_(_dev(other) if k % 2 else t)
        graph.replay()
        torch.cuda.synchronize()
        _check(m, other if k % 2 else cloud, ("replay", k))
        m.create2DMap("slope", t)           # eager builds between the replays
        _check(m, cloud, ("eager between", k))
