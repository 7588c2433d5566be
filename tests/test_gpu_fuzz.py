"""GPU tier: seeded random clouds against the oracle, every strategy.  Small clouds with the structure that breaks
binning code: points exactly on cell boundaries and a few ulps off them, duplicates, runs of 64+ identical points
(the reference converters' padding), clusters that put dozens of points in one node, isolated far points, negative
and positive quadrants, awkward cell sizes."""
import numpy as np
import pytest

from tests import parity

pytestmark = pytest.mark.gpu

CELLS = [(0.5, 0.1), (0.2, 0.2), (0.1, 0.05), (1.0 / 3.0, 0.07), (0.25, 0.5)]


def _cloud(seed):
    rng = np.random.default_rng(seed)
    gl, zl = CELLS[seed % len(CELLS)]
    origin = (rng.random(3) * 20 - 10).astype(np.float32)
    n = int(rng.integers(1, 6000))
    parts = []
    k = rng.integers(0, 5, size=6)
    # clusters: many points per node
    for _ in range(int(k[0]) + 1):
        c = origin + (rng.random(3) * 8 - 4) * np.float32([1, 1, 0.2])
        parts.append((c + rng.normal(0, [0.15, 0.15, 0.02], size=(max(n // 6, 1), 3))).astype(np.float32))
    # lattice points on cell boundaries, and neighbours a few ulps away
    ij = rng.integers(-40, 40, size=(max(n // 8, 1), 3)).astype(np.float64)
    lat = (origin.astype(np.float64) + ij * np.array([gl, gl, zl])).astype(np.float32)
    parts += [lat, np.nextafter(lat, np.float32(np.inf)), np.nextafter(lat, np.float32(-np.inf))]
    # duplicates of existing points and a padding run of identical points (>= 64 in a row somewhere)
    base = np.concatenate(parts, 0)
    parts.append(base[rng.integers(0, base.shape[0], size=max(n // 10, 1))])
    if k[1] >= 2:
        parts.append(np.tile(np.float32([[0, 0, 0]]), (int(rng.integers(64, 400)), 1)))
    # isolated far points
    parts.append((origin + (rng.random((max(n // 20, 1), 3)) * 2 - 1) * np.float32([900, 900, 30])).astype(np.float32))
    body = np.concatenate(parts, 0)
    if k[2] >= 2:
        body = body[rng.permutation(body.shape[0])]
    cloud = np.concatenate([origin[None, :], body], 0).astype(np.float32)
    P = dict(grid_len=gl, z_len=zl, slope_interval=0.08, demand="true" if seed % 7 == 0 else "slope")
    return cloud, P


@pytest.mark.parametrize("seed", range(24))
def test_random_structured_clouds_every_strategy(seed):
    cloud, P = _cloud(1000 + seed)
    ref = parity.ref_from_cloud(cloud, P)
    for strategy in (1, 3, 4, 5):
        m, out = parity.gpu_from_cloud(cloud, P, on_device=True, strategy=strategy)
        parity.assert_parity(out, ref, adversarial=True)


def test_single_points_and_tiny_clouds():
    for n in (0, 1, 2, 3, 4, 5, 63, 64, 65):
        rng = np.random.default_rng(n)
        cloud = np.concatenate([np.float32([[0.1, 0.2, 0.3]]), rng.random((n, 3)).astype(np.float32)], 0)
        P = dict(grid_len=0.5, z_len=0.1, slope_interval=0.08, demand="slope")
        ref = parity.ref_from_cloud(cloud, P)
        for strategy in (1, 3, 4):
            _, out = parity.gpu_from_cloud(cloud, P, on_device=True, strategy=strategy)
            parity.assert_parity(out, ref)                    # (random points: the share caps of tests/parity.py stay in force)


@pytest.mark.parametrize("kind", ["pole", "identical", "two_hot_columns"])
def test_degenerate_large_clouds_fall_through_the_partition_paths(kind):
    """Clouds above the two-level threshold whose points sit in one column, one node or two columns: the fixed-capacity
    level-1 regions, the sampled bucket layout and the LDS node tables cannot all hold them, so the build has to walk
    its retries (learnt capacity, exact partition, larger tables, node table in HBM) and still return the oracle's map."""
    rng = np.random.default_rng(11)
    n = 1_300_000
    if kind == "pole":                                       # one column, ~4000 z levels
        xyz = np.stack([0.3 + 0.1 * rng.random(n), -0.7 + 0.1 * rng.random(n), rng.random(n) * 400.0 - 200.0], 1)
    elif kind == "identical":                                # one node (and the converters' padding pattern)
        xyz = np.tile(np.float64([[2.25, -3.5, 0.125]]), (n, 1))
    else:                                                    # two hot columns among ordinary ground
        xyz = np.stack([rng.random(n) * 60 - 30, rng.random(n) * 60 - 30, 0.02 * rng.normal(size=n)], 1)
        hot = rng.random(n) < 0.6
        xyz[hot, 0] = np.where(rng.random(hot.sum()) < 0.5, 1.26, -7.31) + 0.01 * rng.random(hot.sum())
        xyz[hot, 1] = 4.03 + 0.01 * rng.random(hot.sum())
        xyz[hot, 2] = rng.random(hot.sum()) * 3.0
    cloud = np.concatenate([np.float32([[0.0, 0.0, 0.0]]), xyz.astype(np.float32)], 0)
    P = dict(grid_len=0.5, z_len=0.1, slope_interval=0.08, demand="slope")
    ref = parity.ref_from_cloud(cloud, P)
    for strategy in (0, 4):
        m, out = parity.gpu_from_cloud(cloud, P, on_device=True, strategy=strategy)
        # The share caps of tests/parity.py stay in force except for the pole: ~330 points per node at |z| up to 200 m, where the
        # fp32-sequential oracle itself is more than 5e-6 off the exact scatter on 53 of the 4000 nodes (1.3 %; cap 0.1 %) — the
        # widening FACTOR stays capped for it as well.
        parity.assert_parity(out, ref, adversarial=(kind == "pole"))



@pytest.mark.parametrize("min_points", [1, 2, 5])
def test_min_points_other_than_the_reference_constant(min_points):
    """MINPOINTSIZE is 3 in the reference (map2D.h:28); the handle takes it as a parameter and every place that tests it
    (statistics, the `visited` rule of the slope test, the lazily evaluated `up`) must follow."""
    cloud, P = _cloud(1100)
    P = dict(P, min_points=min_points)
    ref = parity.ref_from_cloud(cloud, P)
    assert ((ref["flags"] & 1) != 0).sum() != ((parity.ref_from_cloud(cloud, dict(P, min_points=3))["flags"] & 1) != 0).sum()
    for strategy in (1, 3, 4):
        _, out = parity.gpu_from_cloud(cloud, P, on_device=True, strategy=strategy)
        parity.assert_parity(out, ref, adversarial=True)       # (_cloud: lattice points and their float neighbours)


@pytest.mark.parametrize("pattern", ["abab", "runs", "abcabc_plus_padding"])
def test_interleaved_nodes_inside_a_wave(pattern):
    """Strategy ATOMIC merges the lanes of a wave that hold the same node before it touches the table (a bitonic sort of the
    wave's keys, then a run reduction): clouds whose consecutive points alternate between a few nodes, with lengths that are
    not multiples of 64, dead lanes at the end and a padding run in the middle — as a build and as a stream of uneven updates."""
    rng = np.random.default_rng(7)
    P = dict(grid_len=0.5, z_len=0.1, slope_interval=0.08, demand="slope")
    centres = np.float32([[1.25, 2.25, 0.05], [1.25, 2.25, 0.15], [3.75, -1.25, 0.05], [1.75, 2.25, 0.05], [-2.25, 0.75, 0.25]])
    n = 64 * 37 + 29
    if pattern == "abab":
        which = np.arange(n) % 2
    elif pattern == "runs":
        which = np.repeat(rng.integers(0, 5, size=n // 3 + 1), rng.integers(1, 7, size=n // 3 + 1))[:n]
    else:
        which = np.arange(n) % 3
    body = centres[which] + (rng.random((n, 3)).astype(np.float32) - 0.5) * np.float32([0.4, 0.4, 0.08])
    if pattern == "abcabc_plus_padding":
        body = np.concatenate([body[:1000], np.zeros((200, 3), np.float32), body[1000:]], 0)
    cloud = np.concatenate([np.float32([[0.1, 0.2, 0.3]]), body], 0)
    ref = parity.ref_from_cloud(cloud, P)
    _, out = parity.gpu_from_cloud(cloud, P, on_device=True, strategy=1)
    parity.assert_parity(out, ref)
    # the same points as a stream of uneven frames (gndt_update_device: the table path's incremental finalisation)
    import torch
    import grid_ndt_amd as g
    m = g.TwoDmap(P["grid_len"], P["z_len"], strategy=1)
    m.setInterval(P["slope_interval"])
    m.setCloudFirst(cloud[0])
    pos = 1
    for step in (700, 1, 63, 64, 65, 900, 10 ** 9):
        chunk = cloud[pos:pos + step]
        if chunk.shape[0] == 0:
            break
        m.change2DMap(P["demand"], torch.from_numpy(np.ascontiguousarray(chunk)).cuda())
        pos += chunk.shape[0]
    parity.assert_parity(m.export(), ref)


@pytest.mark.parametrize("seed", [11, 12])
def test_one_handle_builds_a_sequence_of_unrelated_clouds(seed):
    """tools/fuzz_campaign.py in small: handles of random strategy / cells / hint, each building 2-5 unrelated clouds of random
    structure and size (what a handle learnt from the last cloud is wrong for the next), some as update streams; every map
    against the oracle under the dense-cloud gates of tests/parity.py.  (The long run: profiles/r03_fuzz_campaign.json.)"""
    from tools import fuzz_campaign
    stats = fuzz_campaign.run(seconds=120.0, seed=seed, max_points=400_000, max_handles=12)
    assert not stats["failures"], stats["failures"]
    assert stats["builds"] >= 24
    # The dense gate lets a slope / down label differ from the fp32 oracle's where it is the label the reference's rule gives on the
    # EXACT centroids (tests/parity.py).  That exception is reported and capped at 8e-5 of the nodes (the long campaigns measured
    # 1.0e-5 of 104 M nodes and 5.9e-5 of 617 M with denser clouds, profiles/r04_fuzz_campaign*.json; ADVICE r4: the assertion said
    # 1e-4 under a comment that said 1e-5): more would mean the gate, not the reference's fp32 rounding, is doing the work.
    share = stats.get("labels_within_margin", 0) / max(1, stats.get("nodes", 1))
    print("labels that differ from the fp32 oracle and equal the rule on exact centroids:", stats.get("labels_within_margin", 0), "of",
          stats.get("nodes", 0), "nodes; decided within 1e-5 of the interval:", stats.get("labels_on_the_margin", 0))
    assert share <= 8e-5, (stats.get("labels_within_margin"), stats.get("nodes"))
    # (about a third of them are decided within 1e-5 of the interval on the exact centroids — 8 930 of 27 436 in the long campaign —; the
    #  rest are dense nodes whose SEQUENTIAL fp32 mean in the oracle is further than that from the exact mean: reported, not asserted)
