"""GPU tier for the cost-map flood: gndt_compute_cost (HIP, one launch per layer) against the oracle's sequential
FIFO restatement of TwoDmap::computeCost (map2D.h:1285-1397) run on the SAME grid — the grid the GPU built and
exported — so that h (fp32) and the flood state compare bit for bit."""
import numpy as np
import pytest

from oracle import oracle
from grid_ndt_amd import scenes

pytestmark = pytest.mark.gpu
FLT_MAX = np.float32(3.4028234663852886e38)


@pytest.fixture
def cost_launch_mode():
    """gndt_debug_set_option(GNDT_DEBUG_COST_ONE_WORKGROUP, .) for one test: the flood with or without the one-workgroup kernel."""
    import grid_ndt_amd as g
    def choose(one_workgroup):
        g.TwoDmap.set_debug_option(g.TwoDmap.DEBUG_COST_ONE_WORKGROUP, 1 if one_workgroup else 0)
    yield choose
    choose(True)


def _build(cloud, P, demand="slope", strategy=0):
    import torch
    import grid_ndt_amd as g
    m = g.TwoDmap(P["grid_len"], P["z_len"], strategy=strategy)
    m.setInterval(P["slope_interval"])
    m.setCloudFirst(cloud[0])
    m.create2DMap(demand, torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda())
    return m


def _check(m, cloud, P, goal, demand, robot):
    cells = m.export()
    st = m.computeCost(goal, robot=robot)
    got = m.cost_export()
    ref = oracle.compute_cost(cells, cloud[0], P["grid_len"], P["z_len"], P["slope_interval"], goal, demand=demand, robot=robot)
    assert st["rc"] == ref["rc"]
    # the only non-IEEE step is acosf (device libm vs glibc): exactness is claimed while no decision sits on the threshold
    assert ref["angle_margin_deg"] > 1e-3, "scene puts an angle within 1e-3 deg of the gate: pick another seed"
    np.testing.assert_array_equal(got["h"], ref["h"])
    np.testing.assert_array_equal(got["state"], ref["state"])
    assert (st["traversable"], st["closed"], st["check_pushes"], st["ring"]) == \
           (ref["traversable"], ref["closed"], ref["check_pushes"], ref["ring"])
    return st, got


@pytest.mark.parametrize("demand", ["slope", "true"])
@pytest.mark.parametrize("strategy", [1, 3, 4])
def test_cost_map_equals_reference_flood(demand, strategy):
    cloud = scenes.drivable_site()
    P = scenes.COST_PARAMS
    m = _build(cloud, P, demand, strategy)
    for radius in (0.25, 0.6, 1.3):
        st, got = _check(m, cloud, P, scenes.DRIVABLE_GOAL, demand, {"radius": radius})
        assert st["traversable"] > 8000 and st["closed"] > 0 and st["levels"] > 50
    # a second goal on the platform, another robot
    _check(m, cloud, P, (12.0, -1.0, 0.45 + 0.35 * np.sin(12.0 / 7.0) + 0.25 * np.cos(-1.0 / 5.0)), demand,
           {"radius": 0.25, "reachable_height": 0.1, "max_angle_deg": 20.0})


def test_cost_map_goal_statuses_and_lifetime():
    import torch
    import grid_ndt_amd as g
    cloud = scenes.drivable_site(60000)
    P = scenes.COST_PARAMS
    m = g.TwoDmap(P["grid_len"], P["z_len"])
    m.setInterval(P["slope_interval"])
    m.setCloudFirst(cloud[0])
    with pytest.raises(g.GndtError):
        m.computeCost(scenes.DRIVABLE_GOAL)                    # nothing built yet
    pts = torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda()
    m.create2DMap("slope", pts)
    assert m.computeCost((500.0, 500.0, 0.0))["rc"] == 1       # no cell: the reference silently does nothing
    got = m.cost_export()
    assert (got["h"] == FLT_MAX).all() and (got["state"] == 0).all()
    assert m.computeCost((scenes.DRIVABLE_GOAL[0], scenes.DRIVABLE_GOAL[1], 40.0))["rc"] == 2   # "Goal position wrong"
    assert m.computeCost(scenes.DRIVABLE_GOAL)["rc"] == 0
    m.create2DMap("slope", pts)                                # a new grid invalidates the flood
    with pytest.raises(g.GndtError):
        m.cost_export()


def test_cost_map_on_a_large_terrain():
    """~2M points of LiDAR-ordered terrain at 0.5 m cells: tens of thousands of slopes, hundreds of layers."""
    cloud = scenes.terrain_cloud(2_000_000)
    P = dict(grid_len=0.5, z_len=0.25, slope_interval=0.08)
    m = _build(cloud, P, "slope", 0)
    cells = m.export()
    rows = np.nonzero((cells["flags"] & 2) != 0)[0]
    goal = cells["mean"][rows[len(rows) // 3]]
    st, got = _check(m, cloud, P, goal, "slope", None)
    assert st["rc"] == 0
    reached = got["h"] < FLT_MAX
    g0 = cells["mean"][np.nonzero(got["h"] == 0)[0][0]].astype(np.float64)
    d = np.sqrt(((cells["mean"].astype(np.float64) - g0) ** 2).sum(1))
    assert (got["h"][reached] >= d[reached] * (1 - 1e-5)).all()


def test_cost_map_with_collision_rings_beyond_the_first_scratch():
    """A 1.3 m robot on 0.1 m cells: CollisionCheck's ring covers 27 x 27 columns, thousands of slopes — a std::list in the
    reference (map2D.h:351-411); rounds 1-3 listed it in a scratch of fixed size and once ended with GNDT_ERR_CAPACITY (found by
    tools/fuzz_cost.py); now 13 rounds over the whole map and no list.  Same flood as the oracle's, bit for bit."""
    cloud = scenes.drivable_site(600_000, half=6.0)                # ~40 points per 0.1 m cell
    P = dict(scenes.COST_PARAMS, grid_len=0.1, z_len=0.05)
    m = _build(cloud, P, "slope", 0)
    st, got = _check(m, cloud, P, scenes.DRIVABLE_GOAL, "slope", {"radius": 1.3})
    print("flood", st)
    assert st["rc"] == 0 and st["traversable"] + st["closed"] > 100
    st, got = _check(m, cloud, P, scenes.DRIVABLE_GOAL, "slope", {"radius": 0.25})
    assert st["rc"] == 0 and st["ring_store"] == 1


def _tall_cloud(n=600_000, seed=11):
    rng = np.random.default_rng(seed)
    body = np.empty((n, 3), np.float32)
    body[:, 0:2] = rng.random((n, 2)) * 3.0 + 0.5
    body[:, 2] = rng.random(n) * 12.0
    return np.vstack([np.zeros((1, 3), np.float32), body]).astype(np.float32)


@pytest.mark.parametrize("demand", ["true", "slope"])
def test_cost_map_with_rings_of_thousands_of_slopes(demand):
    """A tall structure on 0.1 m cells, a 0.25 m robot (ring depth 2): demand "true" lists every slope of every cell of the ring,
    gates or not (map2D.h:414-474) — ~1 400 slopes around a slope here; demand "slope" asks the gates of ~110 rows per neighbour
    cell, more than a step mask holds.  (tools/fuzz_cost.py met a flood that spent minutes on ONE such ring when a lane walked it;
    the verdicts now come from ring-depth rounds over the whole map and no ring is listed.)  Same flood as the oracle's."""
    import time
    cloud = _tall_cloud()
    P = dict(grid_len=0.1, z_len=0.1, slope_interval=0.08)
    m = _build(cloud, P, demand, 0)
    cells = m.export()
    rows = np.nonzero((cells["flags"] & 2) != 0)[0]
    assert len(rows) > 5000
    for pick in (len(rows) // 2, int(np.argmax(cells["mean"][rows, 2]))):
        goal = cells["mean"][rows[pick]]
        t = time.perf_counter()
        st = m.computeCost(goal, robot={"radius": 0.25})
        dt = time.perf_counter() - t
        print("flood", st, "%.3f s" % dt)
        assert st["rc"] == 0 and st["ring"] == 2 and st["ring_store"] == 1 and dt < 2.0
        _check(m, cloud, P, goal, demand, {"radius": 0.25})


def _sheets(n=300_000, seed=12, levels=24):
    """Thin, slightly tilted sheets 0.1 m apart over 2 x 2 m: every cell of 0.1 m holds a stack of flat nodes."""
    rng = np.random.default_rng(seed)
    body = np.empty((n, 3), np.float32)
    body[:, 0:2] = rng.random((n, 2)) * 2.0 + 0.5
    k = rng.integers(0, levels, n)
    body[:, 2] = k * 0.1 + 0.05 + rng.normal(0, 0.002, n) + 0.01 * body[:, 0]
    return np.vstack([np.zeros((1, 3), np.float32), body]).astype(np.float32)


@pytest.mark.parametrize("wg", ["1", "0"])
def test_cost_map_where_a_cell_holds_more_accessible_slopes_than_a_record_names(wg, cost_launch_mode):
    """Stacked sheets on 0.1 m cells and a robot that reaches 0.45 m up or down: a neighbour cell holds up to nine accessible slopes,
    a CostEdge record (gndt_cost.hpp) names two — such cells (nine records in ten here) are expanded from their rows inside the layer
    (kEdgeMore), next to cells expanded from their record.  Same flood as the oracle's, by the one-workgroup kernel and by one-layer
    launches."""
    cloud = _sheets()
    P = dict(grid_len=0.1, z_len=0.1, slope_interval=0.12)       # (a node 0.1 m above does not take the slope away: map2D.h:66-108)
    m = _build(cloud, P, "slope", 0)
    cells = m.export()
    rows = np.nonzero((cells["flags"] & 2) != 0)[0]
    assert len(rows) > 4000
    cost_launch_mode(wg != "0")
    robot = {"radius": 0.04, "reachable_height": 0.45, "max_angle_deg": 30.0}
    goal = cells["mean"][rows[len(rows) // 2]]
    st, _ = _check(m, cloud, P, goal, "slope", robot)
    assert st["rc"] == 0 and st["ring"] == 0 and st["traversable"] > 4000
    assert st["check_pushes"] > 20 * st["traversable"]              # (tall cells: ~12 slopes in each of the four)


def test_cost_map_for_goal_after_goal_on_one_map_then_on_the_next():
    """What a flood needs of the map and the robot alone (column index, neighbour columns, CostEdge records, collision verdicts) is
    kept for the next goal on the same map: goals, robots and maps in every order — each flood the oracle's, also right after the
    robot changed (tables worked out again), after the map was rebuilt from another cloud, and back."""
    import torch
    cloud = scenes.drivable_site()
    P = scenes.COST_PARAMS
    m = _build(cloud, P, "slope", 0)
    cells = m.export()
    rows = np.nonzero((cells["flags"] & 2) != 0)[0]
    goals = [scenes.DRIVABLE_GOAL] + [cells["mean"][rows[(7 * k + 3) * len(rows) // 40]] for k in range(4)]
    wide, narrow = {"radius": 0.6}, {"radius": 0.25}
    for goal, robot in ((goals[0], narrow), (goals[1], narrow), (goals[2], wide), (goals[3], wide), (goals[4], narrow), (goals[0], narrow)):
        st, _ = _check(m, cloud, P, goal, "slope", robot)
        assert st["rc"] == 0
    # another map on the same handle (half of the cloud: other nodes, other rows), the same goal and robot as the last flood
    half = np.ascontiguousarray(np.vstack([cloud[:1], cloud[1::2]]))
    m.create2DMap("slope", torch.from_numpy(np.ascontiguousarray(half[1:])).cuda())
    st, _ = _check(m, half, P, goals[0], "slope", narrow)
    st2, _ = _check(m, half, P, goals[1], "slope", narrow)
    m.create2DMap("slope", torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda())
    st3, _ = _check(m, cloud, P, goals[1], "slope", narrow)
    assert st3["traversable"] > 8000


@pytest.mark.parametrize("demand", ["slope", "true"])
def test_cost_map_by_one_layer_launches_only(demand, cost_launch_mode):
    """GNDT_DEBUG_COST_ONE_WORKGROUP = 0: no one-workgroup kernel walking the narrow layers, every layer its own launch (what wide layers get anyway).
    Same flood."""
    cloud = scenes.drivable_site()
    P = scenes.COST_PARAMS
    m = _build(cloud, P, demand, 0)
    cost_launch_mode(False)
    for radius in (0.25, 0.6):
        st, _ = _check(m, cloud, P, scenes.DRIVABLE_GOAL, demand, {"radius": radius})
        assert st["traversable"] > 8000 and st["levels"] > 50


def test_cost_map_with_layers_wider_than_the_one_workgroup_kernel_takes():
    """An open 100 m site on 0.25 m cells: ~150 k slopes, layers grow from one slope to far more than the 320 the one-workgroup
    kernel keeps, and shrink again at the end — the flood is handed from that kernel to one-layer launches (sized from the last layer
    seen) and back.  Same flood as the oracle's; also with one-layer launches only."""
    cloud = scenes.drivable_site(1_500_000, half=50.0)
    P = dict(grid_len=0.25, z_len=0.25, slope_interval=0.08)
    m = _build(cloud, P, "slope", 0)
    st, _ = _check(m, cloud, P, scenes.DRIVABLE_GOAL, "slope", None)
    assert st["traversable"] > 100_000 and st["levels"] > 200
    assert st["traversable"] / st["levels"] > 200          # (the MEAN layer is near the workgroup's share: the wide ones are far beyond it)
