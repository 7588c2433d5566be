"""The planner (AstarPlanar::findRoute, GlobalPlan.h:49-166) and the cost flood under it against routes DERIVED BY HAND.

tests/test_compat_cpp.py compares gndt_compat::AstarPlanar with the oracle's restatement of the same loop — two statements
of GlobalPlan.h by one author.  Here the scenes leave no choice: one-cell-wide corridors of flat floor (4-connected cells,
AccessibleNeighbors looks left / right / front / back, map2D.h:530-588), so the route IS the corridor, cell by cell; a fork
whose short branch is open (the route must take it: Slope::h is the travel cost to the goal, 0.5 m per step) and the same
fork with one cell of the short branch missing (the route must go round).  Only the key strings come from the codec
(transMortonXYZ, pinned by tests/golden/morton_known_answers.csv); the ORDER of the cells is the hand-derived part."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from tests.test_compat_cpp import _build_checker

GL, ZL, IV = 0.5, 0.25, 0.08
ORIGIN = np.float32([0.013, -0.021, 0.05])


def _cells_to_cloud(cells):
    """9 points per cell (3 x 3 inside the cell), floor at z ~ 0.06 with a ripple so that every cell has a proper normal."""
    pts = [ORIGIN]
    for k, (ix, iy) in enumerate(cells):
        for a in (0.12, 0.25, 0.38):
            for b in (0.12, 0.25, 0.38):
                x, y = ix * GL + a, iy * GL + b
                pts.append([x, y, 0.06 + 0.002 * np.sin(7 * x) * np.cos(5 * y)])
    return np.ascontiguousarray(np.float32(pts))


def _centre(cell):
    return (cell[0] * GL + 0.25, cell[1] * GL + 0.25, 0.06)


def _expected_keys(route_cells):
    import grid_ndt_amd as g
    out = []
    for c in route_cells:
        rc, key, nx, ny, sz = g.trans_morton_xyz(ORIGIN, GL, ZL, _centre(c))
        assert rc == 0
        out.append(f"{key}/{sz}")
    return out


def _line(a, b):
    (x0, y0), (x1, y1) = a, b
    assert x0 == x1 or y0 == y1
    n = max(abs(x1 - x0), abs(y1 - y0))
    return [(x0 + (x1 > x0) * k - (x1 < x0) * k, y0 + (y1 > y0) * k - (y1 < y0) * k) for k in range(n + 1)]


def _poly(corners):
    cells = []
    for a, b in zip(corners[:-1], corners[1:]):
        seg = _line(a, b)
        cells += seg if not cells else seg[1:]
    return cells


SCENES = {}
# 1. straight corridor along +x, entirely in quadrant A/B side
SCENES["straight"] = dict(cells=_poly([(2, 3), (21, 3)]), route=_poly([(2, 3), (21, 3)]))
# 2. L-shaped corridor
SCENES["corner"] = dict(cells=_poly([(2, 2), (12, 2), (12, 13)]), route=_poly([(2, 2), (12, 2), (12, 13)]))
# 3. corridor across both origin seams (cell indices have no 0: countLRFB steps -1 -> +1, map2D.h:226-255)
SCENES["seams"] = dict(cells=_poly([(-7, -4), (5, -4), (5, 6)]), route=_poly([(-7, -4), (5, -4), (5, 6)]))
# 4. fork: from (0,0) to (10,0) either straight (11 cells) or round through y = 6 (23 cells): the straight branch wins
_short, _long = _poly([(0, 0), (10, 0)]), _poly([(0, 0), (0, 6), (10, 6), (10, 0)])
SCENES["fork_open"] = dict(cells=sorted(set(_short + _long)), route=_short)
# 5. the same fork with one cell of the short branch missing: the only route left goes round
SCENES["fork_blocked"] = dict(cells=sorted(set([c for c in _short if c != (5, 0)] + _long)), route=_long)


def _run(exe, name, gpu):
    sc = SCENES[name]
    cloud = _cells_to_cloud(sc["cells"])
    start, goal = _centre(sc["route"][0]), _centre(sc["route"][-1])
    with tempfile.NamedTemporaryFile(suffix=".f32") as f:
        cloud.tofile(f.name)
        cmd = [exe, f.name, str(cloud.shape[0]), str(GL), str(ZL), str(IV), "slope"]
        cmd += [repr(float(v)) for v in goal] + [repr(float(v)) for v in start] + ["0.2"] + (["gpu"] if gpu else [])
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, PLAN_PRINT="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    route = [l for l in r.stdout.splitlines() if l.startswith("route:")][-1].split()[1:]
    h_start = float([l for l in r.stdout.splitlines() if l.startswith("h_start:")][-1].split()[1])
    want = _expected_keys(sc["route"])
    assert route == want, (name, route, want)
    # Slope::h at the start = travel cost to the goal = (cells - 1) steps of one cell (means sit at the cell centres)
    assert abs(h_start - GL * (len(want) - 1)) < 1e-3 * len(want), (name, h_start)
    return len(want)


@pytest.mark.parametrize("name", sorted(SCENES))
def test_route_is_the_hand_derived_one(native_lib, name):
    exe = _build_checker(native_lib, "plan_check")
    _run(exe, name, gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SCENES))
def test_route_on_the_gpu_built_grid_is_the_hand_derived_one(native_lib, name):
    exe = _build_checker(native_lib, "plan_check")
    _run(exe, name, gpu=True)
