"""GPU tier (-m gpu): libgndt through its C ABI versus the CPU oracle on the same seeded inputs.
Gates: keys / counts / order / labels bit-exact; covariance within 1e-5 relative (tests/parity.py)."""
import numpy as np
import pytest

from grid_ndt_amd import scenes
from tests import parity

pytestmark = pytest.mark.gpu

UNIFORM_CUBIC = dict(grid_len=0.5, z_len=0.5, slope_interval=0.08, demand="slope")
UNIFORM_ZLAUNCH = dict(grid_len=0.5, z_len=0.1, slope_interval=0.08, demand="slope")   # receiver.cpp:35 default z
TERRAIN = dict(grid_len=0.2, z_len=0.2, slope_interval=0.08, demand="slope")
TERRAIN_TRUE = dict(grid_len=0.2, z_len=0.1, slope_interval=0.08, demand="true")


@pytest.fixture(scope="module", autouse=True)
def _require_gpu():
    import torch
    assert torch.cuda.is_available(), "the gpu tier needs a GPU; libgndt has no CPU path"
    import grid_ndt_amd as g
    g.build_native()
    from grid_ndt_amd import _lib
    _lib.lib()


CASES = {
    "bridge_ground": (lambda: scenes.bridge_ground(), scenes.BRIDGE_PARAMS),            # reference's own scene
    "campus_200k": (lambda: scenes.campus_frame(200000), scenes.CAMPUS_PARAMS),          # BASELINE configs[0] stand-in
    "uniform_300k_cubic": (lambda: scenes.uniform_box(300000), UNIFORM_CUBIC),
    "uniform_300k_z01": (lambda: scenes.uniform_box(300000), UNIFORM_ZLAUNCH),
    "terrain_400k": (lambda: scenes.terrain_cloud(400000), TERRAIN),
    "terrain_true": (lambda: scenes.terrain_cloud(200000), TERRAIN_TRUE),
    "site_zero_padded": (lambda: scenes.site_two_storey(300000), dict(grid_len=0.1, z_len=0.1, slope_interval=0.08, demand="slope")),
}


_REF_CACHE = {}


def _case(name):
    if name not in _REF_CACHE:
        make, P = CASES[name]
        cloud = make()
        _REF_CACHE[name] = (cloud, P, parity.ref_from_cloud(cloud, P))
    return _REF_CACHE[name]


# Strategies: 1 ATOMIC (node table in HBM), 3 PARTITION_EXACT (single-level counting partition), 4 the two-level
# partition forced at any size (2 = PARTITION picks it from 2^20 points).  A forced partition build must end on
# the LDS-resident pipeline: a low node estimate is retried with more buckets (the ATOMIC fallback is for clouds
# needing more buckets than one level can address), an overflowing two-level region falls back to the exact one.
@pytest.mark.parametrize("strategy", [1, 3, 4, 5], ids=["atomic", "partition_exact", "partition_two_level", "tile"])
@pytest.mark.parametrize("name", list(CASES))
def test_parity_device_input(name, strategy):
    cloud, P, ref = _case(name)
    m, out = parity.gpu_from_cloud(cloud, P, on_device=True, strategy=strategy)
    rep = parity.assert_parity(out, ref)
    ran = m.last_strategy()
    print(name, "ran", m.STRATEGY_NAMES[ran], {k: v for k, v in rep.items() if k not in ("fail",)})
    # (a two-level build whose fixed-capacity regions overflow ends on the exact partition: allowed, not expected here)
    assert ran == {1: 1, 3: 3, 4: 2, 5: 5}[strategy], (name, ran)


def test_partition_build_captured_after_an_atomic_fallback_is_replayed_again_and_again():
    """A PARTITION build recorded into a hipGraph right after the handle's eager builds ended on the ATOMIC fallback (columns of
    ~2000 nodes: no LDS table holds them) carries the reset of that table.  The SECOND replay used to walk the table's node list with
    the first replay's node count — the counters are shared, the list is not — and cleared "slots" read from beyond the list: wild
    writes, a GPU memory fault in a process whose allocations held garbage there (tools/fuzz_graph.py --seed 6, round 4).  The
    counters now say whose they are (Counters::part_owned).  Several replays, maps with more nodes than the fallback's, each against
    the oracle; then the handle goes back to the fallback eagerly."""
    import torch
    import grid_ndt_amd as g
    n = 140_000
    P = dict(grid_len=0.1, z_len=0.05, slope_interval=0.08, demand="slope")
    rng = np.random.default_rng(5)
    tall = np.empty((n + 1, 3), np.float32)                    # 25 columns of 0.1 m, 100 m tall: ~2000 levels each
    tall[:, 0:2] = 1.0 + rng.random((n + 1, 2)) * 0.5
    tall[:, 2] = rng.random(n + 1) * 100.0
    flat = []
    for k in range(3):                                         # benign clouds of the same size, more nodes than `tall` has columns
        c = np.empty((n + 1, 3), np.float32)
        c[:, 0:2] = 1.0 + rng.random((n + 1, 2)) * (24.0 + k)      # ~52-58 k nodes: more than the fallback's list, within the capture's tables
        c[:, 2] = 0.01 + rng.random(n + 1) * 0.03
        c[0] = tall[0]
        flat.append(c)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        buf = torch.from_numpy(np.ascontiguousarray(tall[1:])).cuda()
        m = g.TwoDmap(P["grid_len"], P["z_len"], strategy=4, max_points_hint=n + 1)
        m.setInterval(P["slope_interval"])
        m.setCloudFirst(tall[0])
        ref_tall = parity.ref_from_cloud(tall, P, mode=2)
        for _ in range(2):
            m.create2DMap("slope", buf, s)
            out = m.export()
            assert m.STRATEGY_NAMES[m.last_strategy()] == "atomic"           # (the partition attempts all overflowed)
            assert parity.compare(out, ref_tall, "slope", dense=True)["ok"]
        graph = torch.cuda.CUDAGraph()
        with g.graph_capture(graph, s):
            m.create2DMap("slope", buf, s)
        for c in flat + flat[::-1]:
            buf.copy_(torch.from_numpy(np.ascontiguousarray(c[1:])))
            graph.replay()
            s.synchronize()
            out = m.export()
            rep = parity.compare(out, parity.ref_from_cloud(c, P, mode=2), "slope", dense=True)
            assert rep["ok"], rep["fail"][:3]
        buf.copy_(torch.from_numpy(np.ascontiguousarray(tall[1:])))
        m.create2DMap("slope", buf, s)
        assert parity.compare(m.export(), ref_tall, "slope", dense=True)["ok"]


@pytest.mark.parametrize("strategy", [0, 1, 2, 3, 4, 5], ids=["auto", "atomic", "partition", "partition_exact", "partition_two_level", "tile"])
def test_build_captured_on_a_reserved_fresh_handle(strategy):
    """gndt_reserve sizes every buffer a build of that size can ask for, so the FIRST build of a handle can be recorded into a
    hipGraph (no eager warm-up) and replayed on other clouds of that size: nothing allocates under capture (which would invalidate
    it), no replay finds a buffer moved.  Without the reservation the same capture is refused cleanly (GNDT_ERR_CAPACITY)."""
    import torch
    import grid_ndt_amd as g
    from grid_ndt_amd._lib import GndtError
    n = 150_000
    P = scenes.CAMPUS_PARAMS
    clouds = [scenes.campus_frame(n + 1, seed=0x5EED0001 + k) for k in range(3)]
    for c in clouds[1:]:
        c[0] = clouds[0][0]                                   # (the origin stays the handle's)
    refs = [parity.ref_from_cloud(c, P) for c in clouds]
    hint = int(max(r["num_nodes"] for r in refs) * 1.25)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        buf = torch.from_numpy(np.ascontiguousarray(clouds[0][1:])).cuda()
        # not reserved: refused before the allocator is touched, and the process can still capture afterwards
        m0 = g.TwoDmap(P["grid_len"], P["z_len"], strategy=strategy, max_nodes_hint=hint, max_points_hint=n + 1)
        m0.setInterval(P["slope_interval"])
        m0.setCloudFirst(clouds[0][0])
        m0._ensure(P["demand"])                                # (the handle itself is created outside the capture)
        gr = torch.cuda.CUDAGraph()
        with pytest.raises(GndtError) as ei:
            with g.graph_capture(gr, s):
                m0.create2DMap(P["demand"], buf, s)
        assert ei.value.code == 5 and "gndt_reserve" in str(ei.value)
        del m0, gr
        m = g.TwoDmap(P["grid_len"], P["z_len"], strategy=strategy, max_nodes_hint=hint, max_points_hint=n + 1)
        m.setInterval(P["slope_interval"])
        m.setCloudFirst(clouds[0][0])
        m.reserve(n + 1, hint, P["demand"])
        graph = torch.cuda.CUDAGraph()
        with g.graph_capture(graph, s):
            m.create2DMap(P["demand"], buf, s)
        for c, ref in zip(clouds, refs):
            buf.copy_(torch.from_numpy(np.ascontiguousarray(c[1:])))
            graph.replay()
            s.synchronize()
            parity.assert_parity(m.export(), ref)
    print("strategy", strategy, "ran", m.STRATEGY_NAMES[m.last_strategy()])


def test_overflowing_tables_take_the_second_pass_not_a_re_run():
    """Columns of 20 z levels make the node count of a bucket jump by a column at a time: some 512-slot tables overflow whatever the
    average load (here ~384 nodes per table with a spread of ~90).  Those buckets are done again by the bucket kernel's second pass
    (1024 slots) — the build is NOT re-run, on a fresh handle without a hint either (whose node count comes from the HyperLogLog
    pass) — and the map is the oracle's."""
    import grid_ndt_amd as g
    cloud = scenes.uniform_box(3_000_001, seed=0x5EED0301, half_xy=50.0, half_z=1.0)
    P = dict(grid_len=0.5, z_len=0.1, slope_interval=0.08, demand="slope")
    ref = parity.ref_from_cloud(cloud, P, mode=2)
    m, out = parity.gpu_from_cloud(cloud, P, on_device=True, strategy=2)

    def check(o):      # (800 k random nodes: one or two labels sit within an fp32 rounding of the interval — the dense gate's rule, capped)
        rep = parity.compare(o, ref, "slope", dense=True, interval=P["slope_interval"])
        assert rep["ok"], rep["fail"]
        assert rep.get("labels_within_margin", 0) <= 4, rep
    check(out)
    assert m.last_strategy() == 2
    assert m.retry_count() == 0, m.retry_count()
    first = m.second_pass_buckets()
    assert first > 0, "the scene was meant to overflow some tables"
    import torch
    t = torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda()
    for _ in range(2):                                   # steady state: still no re-run, the pass stays on while it has work
        m.create2DMap("slope", t)
        m.sync()
    assert m.retry_count() == 0
    check(m.export())
    print("buckets through the second pass: first build", first, "steady", m.second_pass_buckets())


def test_captured_build_overflow_then_fit_then_overflow():
    """a PARTITION build recorded on a reserved handle and replayed on clouds that do NOT fit what was reserved
    (more nodes than staging rows), then on one that does, then again on one that does not — starting with the overflow, which
    until round 5 left every later, fitting replay "no finished build".  Every overflowing replay is reported (GNDT_ERR_CAPACITY),
    every fitting one exports the oracle's map."""
    import torch
    import grid_ndt_amd as g
    from grid_ndt_amd._lib import GndtError
    n = 200_000
    P = dict(grid_len=0.5, z_len=0.5, slope_interval=0.08, demand="slope")
    small = scenes.uniform_box(n + 1, seed=0x5EED0101, half_xy=20.0)            # ~25 k nodes
    big = scenes.uniform_box(n + 1, seed=0x5EED0102, half_xy=400.0)             # nearly every point its own node
    big[0] = small[0]
    ref_small = parity.ref_from_cloud(small, P)
    assert ref_small["num_nodes"] < 40_000
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        buf = torch.from_numpy(np.ascontiguousarray(big[1:])).cuda()
        m = g.TwoDmap(P["grid_len"], P["z_len"], strategy=2, max_nodes_hint=40_000, max_points_hint=n + 1)
        m.setInterval(P["slope_interval"])
        m.setCloudFirst(small[0])
        m.reserve(n + 1, 40_000, P["demand"])
        graph = torch.cuda.CUDAGraph()
        with g.graph_capture(graph, s):
            m.create2DMap(P["demand"], buf, s)
        for k, (cloud, fits) in enumerate([(big, False), (small, True), (small, True), (big, False), (big, False), (small, True)]):
            buf.copy_(torch.from_numpy(np.ascontiguousarray(cloud[1:])))
            graph.replay()
            s.synchronize()
            if fits:
                parity.assert_parity(m.export(), ref_small)
            else:
                with pytest.raises(GndtError) as ei:
                    m.sync()
                assert ei.value.code == 5, (k, ei.value)
                with pytest.raises(GndtError):
                    m.export()


@pytest.mark.parametrize("bits", [0, 2, 6])
def test_fingerprint_clash_takes_the_exact_second_pass(bits):
    """The bucket kernel names a node by a 21-bit fingerprint of its key and confirms it with the key (gndt_bucket3.hpp); a bucket
    where the fingerprint named the wrong node (~1 in 10^4) is accumulated again with every probe confirmed by the key.  With the
    fingerprint narrowed to 0 / 2 / 6 bits that happens in (nearly) every bucket: the map must still be the oracle's."""
    import grid_ndt_amd as g
    try:
        for name in ("campus_200k", "uniform_300k_z01", "site_zero_padded", "bridge_ground"):
            cloud, P, ref = _case(name)
            from grid_ndt_amd import _lib
            _lib.lib().gndt_debug_set_fp_bits(bits)
            for strategy in (3, 4):
                m, out = parity.gpu_from_cloud(cloud, P, on_device=True, strategy=strategy)
                parity.assert_parity(out, ref)
                clashes = m.fp_clashes()
                print(name, "bits", bits, "strategy", strategy, "buckets redone", clashes)
                if bits <= 2:
                    assert clashes > 0, (name, bits, strategy)
    finally:
        from grid_ndt_amd import _lib
        _lib.lib().gndt_debug_set_fp_bits(21)


@pytest.mark.parametrize("cells", [(0.5, 0.1), (0.1, 0.05)], ids=["launch_cells", "fine_cells"])
def test_depth_camera_frame_every_strategy(cells):
    """One organised 640 x 480 depth frame (what BASELINE configs[0]'s .pcd files are): hundreds of points per node and hot
    columns at the launch cells.  Every strategy gives the oracle's map (a forced two-level build may end on the exact
    partition here: one region holds a third of the frame); AUTO measures the locality and takes TILE at the launch cells."""
    cloud = scenes.depth_frame()
    P = dict(scenes.DEPTH_PARAMS, grid_len=cells[0], z_len=cells[1])
    ref = parity.ref_from_cloud(cloud, P)
    for strategy in (0, 1, 2, 3, 4, 5):
        m, out = parity.gpu_from_cloud(cloud, P, on_device=True, strategy=strategy)
        parity.assert_parity(out, ref)
        ran = m.last_strategy()
        print(cells, "strategy", strategy, "ran", m.STRATEGY_NAMES[ran], "nodes", int(ref["num_nodes"]))
        if strategy == 0 and cells[0] == 0.5:
            assert ran == 5, ran


@pytest.mark.parametrize("strategy", [0, 1, 5], ids=["auto", "atomic", "tile"])
def test_small_maps_are_finalised_by_one_workgroup_and_larger_ones_fall_back(strategy):
    """A map of a few hundred nodes (a depth-camera frame at the launch cells) is finalised by ONE workgroup in one launch
    (k_small_finalize) from the handle's second build on; a cloud with more nodes than that kernel has threads raises its fallback
    flag and is finalised by the regular kernels — eagerly at gndt_sync, as GNDT_ERR_CAPACITY for a replayed hipGraph."""
    import torch
    import grid_ndt_amd as g
    from grid_ndt_amd._lib import GndtError
    frame, P = scenes.depth_frame(), scenes.DEPTH_PARAMS
    n = frame.shape[0] - 1
    big = scenes.campus_frame(n + 1)                           # same point count, ~70 k nodes
    big[0] = frame[0]
    ref_small, ref_big = parity.ref_from_cloud(frame, P), parity.ref_from_cloud(big, P)
    assert ref_small["num_nodes"] < 900 < ref_big["num_nodes"]
    m = g.TwoDmap(P["grid_len"], P["z_len"], strategy=strategy)
    m.setInterval(P["slope_interval"])
    m.setCloudFirst(frame[0])
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        buf = torch.from_numpy(np.ascontiguousarray(frame[1:])).cuda()
        for k in range(3):                                     # build 0: the regular kernels; 1, 2: one workgroup
            m.create2DMap(P["demand"], buf, s)
            parity.assert_parity(m.export(), ref_small)
        ran = m.last_strategy()
        if ran in (1, 5):                                      # (AUTO may have taken TILE or ATOMIC; PARTITION has no small path)
            buf.copy_(torch.from_numpy(np.ascontiguousarray(big[1:])))
            m.create2DMap(P["demand"], buf, s)                 # not small: the fallback inside the build
            parity.assert_parity(m.export(), ref_big)
            buf.copy_(torch.from_numpy(np.ascontiguousarray(frame[1:])))
            for k in range(2):
                m.create2DMap(P["demand"], buf, s)
                parity.assert_parity(m.export(), ref_small)
            # captured with the small finalisation, replayed on the small frame and on the big cloud
            graph = torch.cuda.CUDAGraph()
            with g.graph_capture(graph, s):
                m.create2DMap(P["demand"], buf, s)
            graph.replay()
            s.synchronize()
            parity.assert_parity(m.export(), ref_small)
            buf.copy_(torch.from_numpy(np.ascontiguousarray(big[1:])))
            graph.replay()
            s.synchronize()
            with pytest.raises(GndtError) as ei:
                m.export()
            assert ei.value.code == 5
    print("strategy", strategy, "ran", m.STRATEGY_NAMES[ran])


@pytest.mark.parametrize("n_pts,spread", [(700, 40.0), (1200, 3.0), (1000, 0.9)], ids=["700_columns", "few_hundred_columns", "one_tall_stack"])
def test_small_map_finalisation_orders_many_and_few_columns(n_pts, spread):
    """k_small_finalize sorts the columns by first-seen index — by counting for up to 256 columns, by a bitonic sort above — and walks
    every column once: sparse clouds (one node per column, ~700 columns), clouds of a few hundred columns with several levels each,
    and a single stack of levels, all from the second build of a handle on (the first goes through the regular kernels)."""
    import torch
    import grid_ndt_amd as g
    rng = np.random.default_rng(n_pts)
    body = (rng.random((n_pts, 3)) - 0.5).astype(np.float32) * np.float32([spread, spread, 3.0])
    cloud = np.concatenate([np.float32([[0.01, 0.02, 0.03]]), body], 0)
    P = dict(grid_len=0.5, z_len=0.1, slope_interval=0.08, demand="slope")
    ref = parity.ref_from_cloud(cloud, P)
    assert ref["num_nodes"] <= 900
    m = g.TwoDmap(P["grid_len"], P["z_len"], strategy=1)
    m.setInterval(P["slope_interval"])
    m.setCloudFirst(cloud[0])
    dev = torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda()
    for k in range(3):
        m.create2DMap(P["demand"], dev)
        parity.assert_parity(m.export(), ref)
    print("nodes", int(ref["num_nodes"]), "columns", int(ref["num_columns"]))


def test_partition_with_node_hint_handles_dense_node_sets():
    """With max_nodes_hint the bucket count follows the node count, so node-heavy clouds stay on the LDS path."""
    for name in ("uniform_300k_cubic", "uniform_300k_z01", "site_zero_padded"):
        cloud, P, ref = _case(name)
        for strategy in (3, 4):
            m, out = parity.gpu_from_cloud(cloud, P, strategy=strategy, max_nodes_hint=int(ref["num_nodes"]))
            parity.assert_parity(out, ref)
            assert m.last_strategy() == (3 if strategy == 3 else 2), name


def test_parity_host_input_and_pointxyz_stride():
    cloud = scenes.campus_frame(50000)
    P = scenes.CAMPUS_PARAMS
    ref = parity.ref_from_cloud(cloud, P)
    _, out12 = parity.gpu_from_cloud(cloud, P, on_device=False)
    parity.assert_parity(out12, ref)
    c4 = scenes.with_stride4(cloud)                      # pcl::PointXYZ: 16-byte points
    _, out16h = parity.gpu_from_cloud(c4, P, on_device=False)
    _, out16d = parity.gpu_from_cloud(c4, P, on_device=True)
    for o in (out16h, out16d):
        parity.assert_parity(o, ref)
        for k in ("sx", "count", "first_idx", "flags"):
            assert np.array_equal(o[k], out12[k])


def test_large_pageable_host_input_goes_through_the_bounce_buffers():
    """gndt_build from a pageable host buffer of more than 16 MB is staged through two pinned 8 MB bounce buffers (the CPU fills one
    while the DMA engine empties the other); from pinned memory it is one async copy.  Both give the device build's map, row for row."""
    import torch
    cloud = scenes.terrain_cloud(1_500_000)                   # 18 MB of packed xyz: three bounce chunks, the last one short
    P = TERRAIN
    m_dev, dev = parity.gpu_from_cloud(cloud, P, on_device=True)
    _, pageable = parity.gpu_from_cloud(cloud, P, on_device=False)
    import grid_ndt_amd as g
    m = g.TwoDmap(P["grid_len"], P["z_len"])
    m.setInterval(P["slope_interval"])
    m.setCloudFirst(cloud[0])
    m.create2DMap(P["demand"], torch.from_numpy(np.ascontiguousarray(cloud[1:])).pin_memory())
    pinned = m.export()
    for o in (pageable, pinned):
        assert o["num_nodes"] == dev["num_nodes"] and o["num_columns"] == dev["num_columns"] and o["num_slopes"] == dev["num_slopes"]
        for k in ("sx", "sy", "sz", "count", "first_idx", "flags"):
            assert np.array_equal(o[k], dev[k]), k
    parity.assert_parity(pageable, parity.ref_from_cloud(cloud, P, mode=2))


def test_rebuild_is_idempotent_and_clears_previous_map():
    import torch
    import grid_ndt_amd as g
    a, b = scenes.campus_frame(60000), scenes.uniform_box(50000, seed=7)
    P = scenes.CAMPUS_PARAMS
    m = g.TwoDmap(P["grid_len"], P["z_len"])
    m.setInterval(P["slope_interval"])
    m.setCloudFirst(a[0])
    ta, tb = torch.from_numpy(a[1:]).cuda(), torch.from_numpy(b[1:]).cuda()
    m.create2DMap("slope", ta)
    first = m.export()
    m.create2DMap("slope", tb)          # a different cloud in between: its nodes must be gone afterwards
    m.create2DMap("slope", ta)
    again = m.export()
    for k in ("sx", "sy", "sz", "count", "first_idx", "flags"):
        assert np.array_equal(first[k], again[k]), k
    assert np.allclose(first["cov"], again["cov"], rtol=1e-6, atol=0) and np.allclose(first["mean"], again["mean"], rtol=1e-7)


def test_incremental_update_equals_batch_build():
    """update(F1)..update(Fk) == build(F1||..||Fk), first_idx over the concatenation."""
    import torch
    import grid_ndt_amd as g
    frames = scenes.terrain_frames(3, first_pose=5, points_per_frame=40000)
    cloud = np.concatenate([frames[:1], frames], 0)       # point 0 (origin) = first point, then all frames
    ref = parity.ref_from_cloud(cloud, TERRAIN)
    m = g.TwoDmap(TERRAIN["grid_len"], TERRAIN["z_len"])
    m.setInterval(TERRAIN["slope_interval"])
    m.setCloudFirst(cloud[0])
    for f in range(3):
        t = torch.from_numpy(frames[f * 40000:(f + 1) * 40000]).cuda()
        m.change2DMap("slope", t)
    parity.assert_parity(m.export(), ref)


def test_update_replayed_from_a_hip_graph_equals_batch_build():
    """BASELINE configs[3] (streaming frames): one incremental update is captured in a hipGraph and replayed per
    frame.  Nothing in accumulate + finalize waits for the host; the first_idx base lives on the device."""
    import torch
    import grid_ndt_amd as g
    nf, ppf = 6, 30000
    frames = scenes.terrain_frames(nf, first_pose=3, points_per_frame=ppf)
    cloud = np.concatenate([frames[:1], frames], 0)
    ref = parity.ref_from_cloud(cloud, TERRAIN)
    m = g.TwoDmap(TERRAIN["grid_len"], TERRAIN["z_len"], strategy=1, max_nodes_hint=200000, max_points_hint=nf * ppf)
    m.setInterval(TERRAIN["slope_interval"])
    m.setCloudFirst(cloud[0])
    buf = torch.empty(ppf, 3, dtype=torch.float32, device="cuda")
    host = [torch.from_numpy(frames[f * ppf:(f + 1) * ppf]).cuda() for f in range(nf)]
    buf.copy_(host[0])
    m.change2DMap("slope", buf)          # eager first frame: allocates every buffer
    m.sync()
    graph = torch.cuda.CUDAGraph()
    with g.graph_capture(graph):
        m.change2DMap("slope", buf)      # captured, not executed
    for f in range(1, nf):
        buf.copy_(host[f])
        graph.replay()
    torch.cuda.synchronize()
    parity.assert_parity(m.export(), ref)
    # running past max_points_hint is reported, not silently mis-ordered
    buf.copy_(host[0] + 1000.0)
    graph.replay()
    torch.cuda.synchronize()
    with pytest.raises(g.GndtError):
        m.sync()


def test_deferred_emit_stream_eager_and_replayed_equals_batch_build():
    """gndt_set_deferred_emit: frames stop after relabelling the touched columns; a read in the middle and at the end of the stream gives
    the oracle's map of the points so far — frames launched eagerly, then the same stream with ONE captured update replayed per
    frame (the host cannot see replays: every read of such a handle emits)."""
    import torch
    import grid_ndt_amd as g
    nf, ppf = 7, 30000
    frames = scenes.terrain_frames(nf, first_pose=3, points_per_frame=ppf)
    cloud = np.concatenate([frames[:1], frames], 0)
    ref_all = parity.ref_from_cloud(cloud, TERRAIN)
    ref_mid = parity.ref_from_cloud(cloud[:1 + 4 * ppf], TERRAIN)
    host = [torch.from_numpy(frames[f * ppf:(f + 1) * ppf]).cuda() for f in range(nf)]
    for graph_mode in (False, True):
        m = g.TwoDmap(TERRAIN["grid_len"], TERRAIN["z_len"], strategy=1, max_nodes_hint=200000, max_points_hint=nf * ppf)
        m.setInterval(TERRAIN["slope_interval"])
        m.setCloudFirst(cloud[0])
        m.set_deferred_emit(True)
        buf = torch.empty(ppf, 3, dtype=torch.float32, device="cuda")
        buf.copy_(host[0])
        m.change2DMap("slope", buf)          # the first frame is a full finalisation in either mode
        m.sync()
        graph = None
        if graph_mode:
            graph = torch.cuda.CUDAGraph()
            with g.graph_capture(graph):
                m.change2DMap("slope", buf)
        for f in range(1, nf):
            buf.copy_(host[f])
            graph.replay() if graph else m.change2DMap("slope", buf)
            if f == 3:
                torch.cuda.synchronize()
                parity.assert_parity(m.export(), ref_mid)      # frames 0 .. 3
        torch.cuda.synchronize()
        parity.assert_parity(m.export(), ref_all)
        parity.assert_parity(m.export(), ref_all)              # a second read changes nothing
        m.set_deferred_emit(False)                             # back to dense rows every frame: one more (empty-handed) check of the switch
        del m


def test_deferred_emit_replay_eager_frame_sync_replay_export():
    """a deferred gndt_update recorded into a hipGraph, then — on the same handle — a replay, an EAGER frame,
    a read, another replay and an export.  The eager frame used to clear the handle's "a captured deferred frame exists" flag,
    so the read after the last replay skipped the ordering + emit pass and returned the new node count over stale rows with
    GNDT_OK.  The flag is sticky now: every read of such a handle emits."""
    import torch
    import grid_ndt_amd as g
    nf, ppf = 6, 30000
    frames = scenes.terrain_frames(nf, first_pose=3, points_per_frame=ppf)
    cloud = np.concatenate([frames[:1], frames], 0)
    host = [torch.from_numpy(frames[f * ppf:(f + 1) * ppf]).cuda() for f in range(nf)]
    m = g.TwoDmap(TERRAIN["grid_len"], TERRAIN["z_len"], strategy=1, max_nodes_hint=200000, max_points_hint=nf * ppf)
    m.setInterval(TERRAIN["slope_interval"])
    m.setCloudFirst(cloud[0])
    m.set_deferred_emit(True)
    buf = torch.empty(ppf, 3, dtype=torch.float32, device="cuda")
    buf.copy_(host[0])
    m.change2DMap("slope", buf)              # frame 0: full finalisation
    m.sync()
    graph = torch.cuda.CUDAGraph()
    with g.graph_capture(graph):
        m.change2DMap("slope", buf)          # (recorded, not run)
    buf.copy_(host[1]); graph.replay()       # frame 1: replay
    torch.cuda.synchronize()
    buf.copy_(host[2]); m.change2DMap("slope", buf)      # frame 2: eager
    m.sync()                                              # a read in between
    parity.assert_parity(m.export(), parity.ref_from_cloud(cloud[:1 + 3 * ppf], TERRAIN))
    buf.copy_(host[3]); graph.replay()       # frame 3: replay, unseen by the host
    torch.cuda.synchronize()
    parity.assert_parity(m.export(), parity.ref_from_cloud(cloud[:1 + 4 * ppf], TERRAIN))
    buf.copy_(host[4]); graph.replay()
    buf.copy_(host[5]); m.change2DMap("slope", buf)
    torch.cuda.synchronize()
    parity.assert_parity(m.export(), parity.ref_from_cloud(cloud, TERRAIN))
    del m


def test_split_accumulate_finalize_and_stats_roundtrip():
    """accumulate(shard A) + accumulate(shard B) == build(A||B); stats export -> merge into a second
    handle reproduces the same map (the multi-GPU exchange primitive)."""
    import torch
    import grid_ndt_amd as g
    cloud = scenes.campus_frame(80000)
    P = scenes.CAMPUS_PARAMS
    ref = parity.ref_from_cloud(cloud, P)
    body = torch.from_numpy(cloud[1:]).cuda()
    cut = 33333
    m = g.TwoDmap(P["grid_len"], P["z_len"])
    m.setInterval(P["slope_interval"])
    m.setCloudFirst(cloud[0])
    m.reset("slope")
    m.accumulate("slope", body[cut:], first_idx_base=cut)   # out of order on purpose: first_idx is a min
    m.accumulate("slope", body[:cut], first_idx_base=0)
    st = m.stats_export()
    torch.cuda.synchronize()
    st = {k: v.clone() for k, v in st.items()}
    m.finalize()
    parity.assert_parity(m.export(), ref)
    m2 = g.TwoDmap(P["grid_len"], P["z_len"])
    m2.setInterval(P["slope_interval"])
    m2.setCloudFirst(cloud[0])
    m2.reset("slope")
    half = st["key"].shape[0] // 2
    for sl in (slice(0, half), slice(half, None)):
        m2.stats_merge(st["key"][sl].contiguous(), st["sums"][sl].contiguous(), st["count"][sl].contiguous(),
                       st["first_idx"][sl].contiguous())
    m2.finalize()
    parity.assert_parity(m2.export(), ref)


def test_edge_inputs():
    import torch
    import grid_ndt_amd as g
    m = g.TwoDmap(0.5, 0.1)
    m.setInterval(0.08)
    m.setCloudFirst((0, 0, 0))
    # empty cloud
    m.create2DMap("slope", torch.zeros((0, 3), dtype=torch.float32, device="cuda"))
    assert m.sync() == (0, 0, 0)
    # one and two points: nodes exist but carry no statistics (MINPOINTSIZE, map2D.h:28, 611)
    pts = np.float32([[0.1, 0.1, 0.05], [0.2, 0.2, 0.05]])
    m.create2DMap("slope", pts)
    out = m.export()
    assert out["num_nodes"] == 1 and out["count"][0] == 2 and out["flags"][0] == 0
    assert not out["mean"].any() and not out["cov"].any()
    # ragged sizes around the wave width, all in one cell (the wave-uniform path) and spread out
    for n in (1, 63, 64, 65, 127, 129, 1000):
        same = np.tile(np.float32([[3.3, -2.2, 0.77]]), (n, 1))
        cloud = np.concatenate([np.float32([[0, 0, 0]]), same], 0)
        ref = parity.ref_from_cloud(cloud, scenes.CAMPUS_PARAMS)
        for strategy in (1, 3, 4, 5):
            _, o = parity.gpu_from_cloud(cloud, scenes.CAMPUS_PARAMS, strategy=strategy)
            parity.assert_parity(o, ref)
    # key range: |nx| > 65535 must be an error, not a silent wrap (Stopwatch.h:102-110)
    far = np.float32([[0.5 * 70000, 0, 0]] * 4)
    with pytest.raises(g.GndtError) as e:
        m.create2DMap("slope", far)
    assert e.value.code == 4
    for strategy in (3, 4):
        m2 = g.TwoDmap(0.5, 0.1, strategy=strategy)
        m2.setInterval(0.08)
        m2.setCloudFirst((0, 0, 0))
        with pytest.raises(g.GndtError) as e:
            m2.create2DMap("slope", far)
            m2.sync()
        assert e.value.code == 4


def test_table_growth_without_hint():
    """No max_nodes_hint and every point in its own node: the build must grow its table and still agree."""
    cloud = scenes.uniform_box(120000, half_xy=3000.0)
    P = dict(grid_len=0.5, z_len=0.5, slope_interval=0.08, demand="slope")
    ref = parity.ref_from_cloud(cloud, P)
    _, out = parity.gpu_from_cloud(cloud, P)
    parity.assert_parity(out, ref)
    assert out["num_nodes"] > 100000


def test_full_size_properties_10M():
    """BASELINE configs[1] at full size: size-independent properties instead of the oracle."""
    import torch
    import grid_ndt_amd as g
    cloud = scenes.uniform_box(10_000_000)
    m = g.TwoDmap(0.5, 0.5, max_nodes_hint=1 << 20)
    m.setInterval(0.08)
    m.setCloudFirst(cloud[0])
    t = torch.from_numpy(cloud[1:]).cuda()
    m.create2DMap("slope", t)
    out = m.export()
    assert m.last_strategy() == 2          # AUTO picks the LDS-resident pipeline at this size
    # the two strategies must agree with each other at full size
    ma = g.TwoDmap(0.5, 0.5, max_nodes_hint=1 << 20, strategy=1)
    ma.setInterval(0.08)
    ma.setCloudFirst(cloud[0])
    ma.create2DMap("slope", t)
    oa = ma.export()
    assert ma.last_strategy() == 1
    for k in ("sx", "sy", "sz", "count", "first_idx", "flags"):
        assert np.array_equal(out[k], oa[k]), k
    assert np.abs(out["cov"] - oa["cov"]).max() <= 1e-6 * np.abs(oa["cov"]).max()
    assert np.abs(out["mean"] - oa["mean"]).max() < 1e-5
    del ma
    n = cloud.shape[0] - 1
    assert int(out["count"].astype(np.int64).sum()) == n                       # every point binned once
    keys = (out["sx"].astype(np.int64) << 42) ^ (out["sy"].astype(np.int64) << 21) ^ out["sz"].astype(np.int64)
    assert np.unique(keys).size == out["num_nodes"]                            # one row per node
    f = out["first_idx"].astype(np.int64)
    assert np.unique(f).size == f.size and f.min() == 0 and f.max() < n
    # order: columns by first-seen, nodes inside a column by first-seen
    col = (out["sx"].astype(np.int64) << 32) ^ (out["sy"].astype(np.int64) & 0xFFFFFFFF)
    new_col = np.concatenate([[True], col[1:] != col[:-1]])
    assert int(new_col.sum()) == out["num_columns"] == np.unique(col).size       # columns are contiguous
    col_first = f[new_col]
    assert np.all(np.diff(col_first) > 0)
    same = ~new_col[1:]
    assert np.all(np.diff(f)[same] > 0)
    has = (out["flags"] & 1) != 0
    assert np.array_equal(has, out["count"] >= 3)
    body = cloud[1:].astype(np.float64)
    # trace(scatter) is additive: sum over nodes <= total scatter about the global mean
    tr = (out["cov"][:, 0] + out["cov"][:, 3] + out["cov"][:, 5]).astype(np.float64).sum()
    total_tr = ((body - body.mean(0)) ** 2).sum()
    assert 0 < tr < total_tr
    # a random sample of nodes against an independent numpy fp64 recomputation
    rng = np.random.default_rng(0)
    o = cloud[0]
    pick = rng.choice(np.flatnonzero(has), 200, replace=False)
    d = (cloud[1:] - o)
    gl = np.float32(0.5)
    nx = np.maximum(1, np.ceil(np.abs(d[:, 0]) / gl)).astype(np.int32) * np.where(cloud[1:, 0] > o[0], 1, -1)
    ny = np.maximum(1, np.ceil(np.abs(d[:, 1]) / gl)).astype(np.int32) * np.where(cloud[1:, 1] > o[1], 1, -1)
    nz = np.maximum(1, np.ceil(np.abs(d[:, 2]) / gl)).astype(np.int32) * np.where(cloud[1:, 2] > o[2], 1, -1)
    for i in pick:
        sel = (nx == out["sx"][i]) & (ny == out["sy"][i]) & (nz == out["sz"][i])
        p = body[sel]
        assert p.shape[0] == out["count"][i]
        mu = p.mean(0)
        S = (p - mu).T @ (p - mu)
        ref_ut = np.array([S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]])
        assert np.abs(out["cov"][i] - ref_ut).max() <= 1e-5 * np.abs(ref_ut).max()
        assert np.abs(out["mean"][i] - mu).max() <= 1e-5 * max(1.0, np.abs(mu).max())
        ev = np.linalg.eigvalsh(S)
        # 3-point nodes are exactly coplanar: lambda_min = 0 reads 0.01 (map2D.h:131-132)
        lam = 0.0 if (out["rough"][i] == np.float32(0.01) and ev[0] < 1e-3) else out["rough"][i]
        assert abs(lam - ev[0]) <= 1e-5 * np.trace(S)


def test_single_hip_runtime_and_foreign_stream():
    """libgndt must share the process's HIP runtime with torch (streams/events/allocations are per
    runtime), and honour a non-default torch stream handed across the C ABI."""
    import torch
    import grid_ndt_amd as g
    maps = open("/proc/self/maps").read()
    libs = {line.split()[-1] for line in maps.splitlines() if "libamdhip64" in line}
    assert len(libs) == 1, libs
    cloud = scenes.campus_frame(100000)
    P = scenes.CAMPUS_PARAMS
    ref = parity.ref_from_cloud(cloud, P)
    m = g.TwoDmap(P["grid_len"], P["z_len"])
    m.setInterval(P["slope_interval"])
    m.setCloudFirst(cloud[0])
    s = torch.cuda.Stream()
    host = torch.from_numpy(cloud[1:]).pin_memory()
    with torch.cuda.stream(s):
        dev = host.to("cuda", non_blocking=True)      # the copy is only ordered with the build via stream s
        m.create2DMap("slope", dev, stream=s)
    parity.assert_parity(m.export(), ref)
    views = m.export_device()
    torch.cuda.synchronize()
    assert int(views["count"].sum().item()) == cloud.shape[0] - 1


@pytest.mark.parametrize("strategy", [0, 1])
@pytest.mark.parametrize("scene", ["terrain", "campus_true", "uniform_large"])
def test_sharded_cloud_statistics_merge_to_the_global_map(strategy, scene):
    """Three shards of one cloud, each turned into statistics by its own handle (shard_stats: the counting
    partition for strategy AUTO, the node table for ATOMIC), merged the way grid_ndt_amd/dist.py merges them
    (union of keys, sums add, first-seen takes the min), finalised from the key-sorted statistics: must equal
    the oracle's map of the whole cloud."""
    import torch
    import grid_ndt_amd as g
    if scene == "uniform_large" and strategy == 1:
        pytest.skip("the node-table shard path is covered by the smaller scenes")
    make = {"terrain": lambda: (scenes.terrain_cloud(400000), TERRAIN),
            "campus_true": lambda: (scenes.campus_frame(200000), dict(scenes.CAMPUS_PARAMS, demand="true")),
            # shards above 2^20 points: the two-level partition with the statistics epilogue
            "uniform_large": lambda: (scenes.uniform_box(3_600_001, half_xy=60.0),
                                      dict(grid_len=0.5, z_len=0.5, slope_interval=0.08, demand="slope"))}[scene]
    cloud, P = make()
    ref = parity.ref_from_cloud(cloud, P)
    body = torch.from_numpy(cloud[1:]).cuda()
    n = body.shape[0]
    cuts = [0, n // 3, 2 * n // 3 + 70001, n] if scene == "uniform_large" else [0, n // 5, n // 5 + 70001, n]
    parts = []
    for r in range(3):
        m = g.TwoDmap(P["grid_len"], P["z_len"], strategy=strategy)
        m.setInterval(P["slope_interval"])
        m.setCloudFirst(cloud[0])
        st = m.shard_stats(P["demand"], body[cuts[r]:cuts[r + 1]], first_idx_base=cuts[r])
        expect = (1,) if strategy == 1 else ((2,) if cuts[r + 1] - cuts[r] >= (1 << 20) else (3, 6))   # (6: one-level tile partition)
        assert m.last_strategy() in expect, (r, m.last_strategy())
        parts.append({k: v.clone() for k, v in st.items()})
        assert int(parts[-1]["count"].sum().item()) == cuts[r + 1] - cuts[r]
    keys = torch.cat([p["key"] for p in parts])
    union, inv = torch.unique(keys, return_inverse=True)
    c = union.shape[0]
    sums = torch.zeros((c, 9), dtype=torch.float64, device="cuda").index_add_(0, inv, torch.cat([p["sums"] for p in parts]))
    count = torch.zeros(c, dtype=torch.int32, device="cuda").index_add_(0, inv, torch.cat([p["count"] for p in parts]))
    first = torch.full((c,), 2**31 - 1, dtype=torch.int32, device="cuda").scatter_reduce_(
        0, inv, torch.cat([p["first_idx"] for p in parts]), reduce="amin")
    m = g.TwoDmap(P["grid_len"], P["z_len"], strategy=strategy)
    m.setInterval(P["slope_interval"])
    m.setCloudFirst(cloud[0])
    m._ensure(P["demand"])
    m.finalize_stats(union, sums, count, first, total_points=n)
    parity.assert_parity(m.export(), ref)
    # a first index beyond total_points is reported, not silently mis-ordered
    m.finalize_stats(union, sums, count, first, total_points=1000)
    with pytest.raises(g.GndtError):
        m.sync()


def test_global_map_exchange_on_rccl_single_rank():
    """grid_ndt_amd/dist.py end to end on the GPU (backend nccl = RCCL, world_size 1): accumulate ->
    stats export -> all_gather / all_reduce -> merge -> finalize must equal the plain build."""
    import os
    import torch
    import torch.distributed as dist
    import grid_ndt_amd as g
    from grid_ndt_amd.dist import build_global_map
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29651")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        cloud = scenes.terrain_cloud(200000)
        ref = parity.ref_from_cloud(cloud, TERRAIN)
        m = g.TwoDmap(TERRAIN["grid_len"], TERRAIN["z_len"])
        m.setInterval(TERRAIN["slope_interval"])
        m.setCloudFirst(cloud[0])
        pts = torch.from_numpy(cloud[1:]).cuda()
        build_global_map(m, "slope", pts, 0)
        parity.assert_parity(m.export(), ref)
        # the same build with the exchange inside libgndt (RCCL called from C++, gndt_build_global_device): the
        # communicator's id travels over the process group, the data does not
        from grid_ndt_amd.dist import Communicator
        comm = Communicator(0)
        m2 = g.TwoDmap(TERRAIN["grid_len"], TERRAIN["z_len"])
        m2.setInterval(TERRAIN["slope_interval"])
        m2.setCloudFirst(cloud[0])
        t = m2.build_global(comm, "slope", pts, 0, pts.shape[0], timed=True)
        print("exchange inside libgndt:", t)
        assert t["ranks"] == 1 and t["global_nodes"] == t["local_nodes"] == ref["num_nodes"]
        parity.assert_parity(m2.export(), ref)
        # a rank whose shard is empty still takes part
        m3 = g.TwoDmap(TERRAIN["grid_len"], TERRAIN["z_len"])
        m3.setInterval(TERRAIN["slope_interval"])
        m3.setCloudFirst(cloud[0])
        m3.build_global(comm, "slope", pts[:0], 0, 0)
        assert m3.sync() == (0, 0, 0)
        comm.close()
    finally:
        dist.destroy_process_group()


def test_large_terrain_both_strategies_agree():
    """A 16 M-point LiDAR-ordered terrain (BASELINE configs[2] scaled to one GPU, 0.2 m voxels): the two
    strategies must produce the same map; size-independent properties stand in for the oracle."""
    import torch
    import grid_ndt_amd as g
    cloud = scenes.terrain_cloud(16_000_000)
    t = torch.from_numpy(cloud[1:]).cuda()
    outs = {}
    for strat in (2, 1):
        m = g.TwoDmap(0.2, 0.2, strategy=strat)
        m.setInterval(0.08)
        m.setCloudFirst(cloud[0])
        m.create2DMap("slope", t)
        assert m.last_strategy() == strat
        outs[strat] = m.export()
        del m
    a, b = outs[2], outs[1]
    assert a["num_nodes"] == b["num_nodes"] and a["num_columns"] == b["num_columns"] and a["num_slopes"] == b["num_slopes"]
    for k in ("sx", "sy", "sz", "count", "first_idx", "flags"):
        assert np.array_equal(a[k], b[k]), k
    assert int(a["count"].astype(np.int64).sum()) == cloud.shape[0] - 1
    sc = np.abs(b["cov"]).max(1)
    assert (np.abs(a["cov"] - b["cov"]).max(1) <= 1e-6 * np.maximum(sc, 1e-30)).all()
    assert np.abs(a["mean"] - b["mean"]).max() <= 1e-5


def test_table_entry_points_refuse_a_partition_built_map():
    """The additive state lives in the node table; after a PARTITION build (pending or finished) the table calls must
    say so instead of reading counters that describe another pipeline, and gndt_reset makes them usable again."""
    import torch
    import grid_ndt_amd as g
    cloud = scenes.campus_frame(150000)
    P = scenes.CAMPUS_PARAMS
    m = g.TwoDmap(P["grid_len"], P["z_len"], strategy=3)
    m.setInterval(P["slope_interval"])
    m.setCloudFirst(cloud[0])
    pts = torch.from_numpy(cloud[1:]).cuda()
    m.create2DMap("slope", pts)                       # launched, still pending
    for call in (lambda: m.finalize(), lambda: m.accumulate("slope", pts[:1000]), lambda: m.change2DMap("slope", pts[:1000]),
                 lambda: m.stats_export()):
        with pytest.raises(g.GndtError):
            call()
    parity.assert_parity(m.export(), parity.ref_from_cloud(cloud, P))      # the pending build is still resolved correctly
    m.reset("slope")
    m.accumulate("slope", pts)
    m.finalize()
    parity.assert_parity(m.export(), parity.ref_from_cloud(cloud, P))


def test_partition_build_replayed_from_a_hip_graph():
    """A PARTITION build enqueues its kernels and the read-back of its flags without waiting for the host, so once its
    buffers exist it can be captured in a hipGraph and replayed on new data in the same device buffer."""
    import torch
    import grid_ndt_amd as g
    n = 1_200_000                                        # large enough for the two-level partition
    clouds = [scenes.uniform_box(n + 1, seed=0x5EED0100 + k, half_xy=40.0) for k in range(3)]
    origin = clouds[0][0]
    m = g.TwoDmap(0.5, 0.5, strategy=2, max_nodes_hint=200000)
    m.setInterval(0.08)
    m.setCloudFirst(origin)
    buf = torch.empty(n, 3, dtype=torch.float32, device="cuda")
    buf.copy_(torch.from_numpy(clouds[0][1:]))
    for _ in range(2):                                    # eager builds: every buffer exists, capacities are learnt
        m.create2DMap("slope", buf)
        m.sync()
    assert m.last_strategy() in (2, 7)                    # (7: the second eager build of this dense box takes blocked buckets)
    graph = torch.cuda.CUDAGraph()
    with g.graph_capture(graph):
        m.create2DMap("slope", buf)
    assert m.last_strategy() == 2                         # a RECORDED build never does: its replays see other clouds
    for k in (1, 2):
        buf.copy_(torch.from_numpy(clouds[k][1:]))
        graph.replay()
        torch.cuda.synchronize()
        cloud = np.concatenate([origin[None, :], clouds[k][1:]], 0)
        ref = parity.ref_from_cloud(cloud, dict(grid_len=0.5, z_len=0.5, slope_interval=0.08, demand="slope"))
        parity.assert_parity(m.export(), ref)


def test_auto_build_captured_when_its_locality_answer_is_due_for_renewal():
    """AUTO re-measures a cloud's locality every 64 builds, with a host wait.  A build captured into a hipGraph at that
    moment keeps the answer the handle has instead of waiting inside the capture."""
    import torch
    import grid_ndt_amd as g
    cloud = scenes.depth_frame()
    P = scenes.DEPTH_PARAMS
    ref = parity.ref_from_cloud(cloud, P)
    buf = torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda()
    m = g.TwoDmap(P["grid_len"], P["z_len"])
    m.setInterval(P["slope_interval"])
    m.setCloudFirst(cloud[0])
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(63):                                     # the 64th build would measure again
            m.create2DMap("slope", buf, s)
            m.sync()
        graph = torch.cuda.CUDAGraph()
        with g.graph_capture(graph, s):
            for _ in range(3):
                m.create2DMap("slope", buf, s)
        graph.replay()
        s.synchronize()
    parity.assert_parity(m.export(), ref)
    assert m.last_strategy() == 5


def test_long_stream_incremental_finalisation_equals_batch_builds():
    """gndt_update relabels only the columns a frame touched (after the first, full, finalisation).  A stream of uneven
    frames — some only revisiting old nodes, some opening columns, small ones that lift nodes over the 3-point threshold —
    must equal the batch build of the concatenation at every checkpoint; a call that borrows the order's buffers in the
    middle (pack_points) and a table that has to grow must not disturb it."""
    import torch
    import grid_ndt_amd as g
    ppf = 20000
    frames = scenes.terrain_frames(12, first_pose=2, points_per_frame=ppf)
    cuts = [0]
    rng = np.random.default_rng(5)
    while cuts[-1] < frames.shape[0]:
        cuts.append(min(frames.shape[0], cuts[-1] + int(rng.choice([37, 500, 7000, 20000, 33000]))))
    m = g.TwoDmap(TERRAIN["grid_len"], TERRAIN["z_len"], strategy=1)       # no hint: the table grows on the way
    m.setInterval(TERRAIN["slope_interval"])
    m.setCloudFirst(frames[0])
    dev = torch.from_numpy(frames).cuda()
    for k in range(len(cuts) - 1):
        m.change2DMap("slope", dev[cuts[k]:cuts[k + 1]])
        if k == 5:
            junk = torch.zeros(1000, 4, dtype=torch.float32, device="cuda")
            assert m.pack_points(junk, 16).shape[0] == 1000           # uses the order's bitmap as scratch
        if k in (2, 6, len(cuts) - 2):
            cloud = np.concatenate([frames[:1], frames[:cuts[k + 1]]], 0)
            parity.assert_parity(m.export(), parity.ref_from_cloud(cloud, TERRAIN))
    # revisiting exactly the same points again doubles every count and keeps keys, order and labels' inputs consistent
    before = m.export()
    m.change2DMap("slope", dev[:cuts[3]])
    after = m.export()
    assert np.array_equal(after["sx"], before["sx"]) and np.array_equal(after["first_idx"], before["first_idx"])
    assert int(after["count"].astype(np.int64).sum()) == int(before["count"].astype(np.int64).sum()) + cuts[3]


@pytest.mark.gpu
def test_auto_takes_the_one_pass_tile_strategy_only_where_the_sampled_locality_pays():
    """Strategy AUTO measures points per partial on a sample of the cloud (gndt_locality_sample) before it picks the
    one-pass TILE path.  A dense scan-ordered cloud (many consecutive points per node) must take it and still equal the
    partition result; a shuffled cloud must not."""
    import torch
    import grid_ndt_amd as g
    rng = np.random.default_rng(5)
    n = 600_000
    # dense raster: 300 points per 0.5 m cell in a row, rows one after the other -> long runs of one node
    t = np.arange(n)
    x = (t % 60_000) * (100.0 / 60_000) * 0.1 + 0.0003 * rng.random(n)
    y = (t // 60_000) * 0.5 + 0.2 + 0.1 * rng.random(n)
    z = 0.05 * np.sin(x) + 0.01 * rng.random(n)
    dense = np.ascontiguousarray(np.stack([x, y, z], 1).astype(np.float32))
    shuffled = np.ascontiguousarray(dense[rng.permutation(n)])
    shuffled[0] = dense[0]
    P = dict(grid_len=0.5, z_len=0.25, slope_interval=0.08, demand="slope")
    res = {}
    for name, cloud in (("dense", dense), ("shuffled", shuffled)):
        m = g.TwoDmap(P["grid_len"], P["z_len"], strategy=0)
        m.setInterval(P["slope_interval"]); m.setCloudFirst(cloud[0])
        pts = torch.from_numpy(cloud[1:]).cuda()
        ratio = m.locality_sample(pts)
        # (the kernel's last workgroup sends the totals home and leaves the device words zero: the same answer every time, also with
        #  builds — which sample again by themselves — and other tile counts in between)
        assert m.locality_sample(pts) == ratio and m.locality_sample(pts, tiles=7) > 0 and m.locality_sample(pts) == ratio
        m.create2DMap("slope", pts)
        m.sync()
        assert m.locality_sample(pts) == ratio
        res[name] = (ratio, m.last_strategy(), m.export())
        print(name, "points per partial %.1f" % ratio, "->", m.STRATEGY_NAMES[m.last_strategy()])
    assert res["dense"][0] > 48 and res["dense"][1] == 5
    assert res["shuffled"][0] < 24 and res["shuffled"][1] in (2, 3)
    ref = parity.ref_from_cloud(dense, P)
    parity.assert_parity(res["dense"][2], ref)


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["campus", "terrain"])
def test_remove_is_the_inverse_of_update(scene):
    """gndt_remove* (intent of del2DMap, map2D.h:826-915, defined in include/gndt.h): after update(A), update(B), remove(B')
    the map is the oracle's map of the remaining points — keys, counts, first-seen order and labels exactly, statistics
    within the parity tolerances — as long as no surviving node loses its first point (B' is taken from the END of the
    stream).  Nodes that only B' filled are deleted; nodes that fall below three points lose their statistics."""
    import torch
    import grid_ndt_amd as g
    cloud, P = {"campus": (scenes.campus_frame(120_000), scenes.CAMPUS_PARAMS), "terrain": (scenes.terrain_cloud(200_000), TERRAIN)}[scene]
    n = cloud.shape[0]
    cut = int(0.7 * n)
    rng = np.random.default_rng(9)
    # B' = a random 60 % of the last 30 % of the stream
    tail = np.arange(cut, n)
    gone = np.sort(rng.choice(tail, size=int(0.6 * tail.size), replace=False))
    keep = np.ones(n, bool)
    keep[gone] = False
    remaining = cloud[keep]
    ref = parity.ref_from_cloud(remaining, P)
    # a surviving node must not have lost its first point for exact equality: true by construction for nodes first seen
    # before `cut`; nodes first seen in the tail are compared only if their first point survived
    m = g.TwoDmap(P["grid_len"], P["z_len"], strategy=1)
    m.setInterval(P["slope_interval"]); m.setCloudFirst(cloud[0])
    m.change2DMap(P["demand"], torch.from_numpy(cloud[1:cut]).cuda())
    m.change2DMap(P["demand"], torch.from_numpy(cloud[cut:]).cuda())
    before = m.sync()
    m.del2DMap(P["demand"], torch.from_numpy(np.ascontiguousarray(cloud[gone])).cuda())
    out = m.export()
    assert out["num_nodes"] == ref["num_nodes"] < before[0]              # some nodes died
    # keys as sets, counts per key exact
    key = lambda d: (d["sx"].astype(np.int64) << 42) ^ (d["sy"].astype(np.int64) << 21) ^ (d["sz"].astype(np.int64) & 0x1FFFFF)
    og, orf = np.argsort(key(out)), np.argsort(key(ref))
    assert np.array_equal(key(out)[og], key(ref)[orf])
    assert np.array_equal(out["count"][og].astype(np.int64), ref["count"][orf].astype(np.int64))
    # stream index of a remaining point -> its index in the fresh build of the remaining points
    new_index = np.cumsum(keep) - 1
    first_new = new_index[np.minimum(out["first_idx"][og].astype(np.int64) + 1, n - 1)] - 1     # (+1 / -1: point 0 is the origin)
    first_kept = keep[np.minimum(out["first_idx"][og].astype(np.int64) + 1, n - 1)]
    same_first = first_kept & (first_new == ref["first_idx"][orf].astype(np.int64))
    assert same_first.mean() > 0.9                                        # most nodes kept their first point ...
    # ... and where every node of the map did, the whole export is the fresh build's, in its order
    if same_first.all():
        out2 = dict(out)
        out2["first_idx"] = (new_index[out["first_idx"].astype(np.int64) + 1] - 1).astype(np.uint32)
        parity.assert_parity(out2, ref)
    else:
        has = (ref["flags"][orf] & 1) != 0
        assert np.array_equal((out["flags"][og] & 1) != 0, has)
        dm = np.abs(out["mean"][og][has].astype(np.float64) - ref["mean64"][orf][has])
        assert dm.max() < 1e-5 * max(1.0, np.abs(ref["mean64"]).max())
        sc = np.abs(ref["cov64"][orf][has]).max(axis=1)
        dc = np.abs(out["cov"][og][has].astype(np.float64) - ref["cov64"][orf][has]).max(axis=1)
        assert np.all(dc <= 1e-5 * sc + 1e-12)
    # a point that was never added is an error
    with pytest.raises(g.GndtError):
        m.del2DMap(P["demand"], torch.from_numpy(np.float32([[1e4, 1e4, 1e3]])).cuda())


@pytest.mark.gpu
def test_remove_whole_frames_equals_the_build_without_them():
    """Frames appended last and removed again: no surviving node can have lost its first point, so the export must be the
    oracle's map of the earlier frames, order and labels included."""
    import torch
    import grid_ndt_amd as g
    ppf = 8192
    frames = scenes.terrain_frames(6, 0, points_per_frame=ppf)
    P = TERRAIN
    m = g.TwoDmap(P["grid_len"], P["z_len"], strategy=1)
    m.setInterval(P["slope_interval"]); m.setCloudFirst(frames[0])
    m.change2DMap("slope", torch.from_numpy(frames[1:4 * ppf]).cuda())
    for f in (4, 5):
        m.change2DMap("slope", torch.from_numpy(frames[f * ppf:(f + 1) * ppf]).cuda())
    for f in (5, 4):
        m.del2DMap("slope", torch.from_numpy(frames[f * ppf:(f + 1) * ppf]).cuda())
    parity.assert_parity(m.export(), parity.ref_from_cloud(frames[:4 * ppf], P))


def test_small_clouds_take_the_one_level_tile_partition_and_fall_back_when_a_bucket_outgrows_its_room():
    """Clouds that need at most 512 buckets: level 1 writes the buckets themselves (strategy 6 reported), every bucket with room
    for 4 x the mean.  A cloud with one very hot column overflows that room: the build is re-run on the counting partition and the
    handle stays there."""
    import torch
    import grid_ndt_amd as g
    for name in ("campus_200k", "bridge_ground"):
        cloud, P, ref = _case(name)
        m, out = parity.gpu_from_cloud(cloud, P, on_device=True, strategy=2)
        parity.assert_parity(out, ref)
        assert m.last_strategy() == 6, (name, m.last_strategy())
        # 16-byte points through the same path
        m2 = g.TwoDmap(P["grid_len"], P["z_len"], strategy=2)
        m2.setInterval(P["slope_interval"]); m2.setCloudFirst(cloud[0])
        p16 = torch.zeros((cloud.shape[0] - 1, 4), dtype=torch.float32, device="cuda")
        p16[:, :3] = torch.from_numpy(cloud[1:]).cuda()
        m2.create2DMap(P["demand"], p16)
        parity.assert_parity(m2.export(), ref)
        assert m2.last_strategy() == 6
    rng = np.random.default_rng(3)
    n = 300_000
    xyz = np.stack([rng.random(n) * 60 - 30, rng.random(n) * 60 - 30, 0.02 * rng.normal(size=n)], 1)
    hot = rng.random(n) < 0.5
    xyz[hot, 0] = 1.26 + 0.01 * rng.random(hot.sum()); xyz[hot, 1] = 4.03 + 0.01 * rng.random(hot.sum()); xyz[hot, 2] = rng.random(hot.sum()) * 3.0
    cloud = np.concatenate([np.float32([[0.0, 0.0, 0.0]]), xyz.astype(np.float32)], 0)
    P = dict(grid_len=0.5, z_len=0.1, slope_interval=0.08, demand="slope")
    ref = parity.ref_from_cloud(cloud, P)
    m, out = parity.gpu_from_cloud(cloud, P, on_device=True, strategy=2)
    parity.assert_parity(out, ref)
    assert m.last_strategy() == 3 and m.retry_count() >= 1
    before = m.retry_count()
    m.create2DMap("slope", torch.from_numpy(cloud[1:]).cuda())
    parity.assert_parity(m.export(), ref)
    assert m.last_strategy() == 3 and m.retry_count() == before          # (no second attempt at the one-level path)
