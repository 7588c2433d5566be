"""examples/receiver_main.cpp: the reference node's callback (src/receiver.cpp:137-176) as a ROS-free C++ program on top
of the C ABI and gndt_compat.hpp — .pcd file -> grid -> cost map -> A* route.  CPU tier: it compiles and links against
libgndt.  GPU tier: its route equals the oracle's AstarPlanar on the same (NaN-stripped) cloud, step for step."""
import os
import subprocess

import numpy as np
import pytest

from grid_ndt_amd import scenes
from tests.test_input_side import _write_pcd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_example(name="receiver_main"):
    from grid_ndt_amd import _lib
    exe = os.path.join(ROOT, "examples", name)
    src = exe + ".cpp"
    deps = [src, os.path.join(ROOT, "include", "gndt_compat.hpp"), os.path.join(ROOT, "include", "gndt.h"), _lib.LIB_PATH]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        csrc = os.path.dirname(_lib.LIB_PATH)
        hip = _lib._hip_runtime_dir()
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                               "-o", exe, src, "-L", csrc,
                               "-l:libgndt.so", "-L", hip, "-l:libamdhip64.so", f"-Wl,-rpath,{csrc}", f"-Wl,-rpath,{hip}"])
    return exe


def test_example_builds_and_reports_a_missing_file(native_lib):
    exe = _build_example()
    r = subprocess.run([exe, "/nonexistent.pcd", "0.5", "0.25", "0.08", "slope", "0", "0", "0", "1", "1", "0", "0.25"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "cannot open" in r.stdout


@pytest.mark.gpu
def test_example_route_equals_the_oracle(tmp_path, native_lib):
    from oracle import oracle
    exe = _build_example()
    cloud = scenes.drivable_site()
    P = scenes.COST_PARAMS
    dirty = np.insert(cloud, [3, 1000, 200000], np.float32([1.0, np.nan, 2.0]), axis=0)
    rec = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad", "<f4")])           # pcl::PointXYZ records
    a = np.zeros(dirty.shape[0], rec)
    a["x"], a["y"], a["z"] = dirty[:, 0], dirty[:, 1], dirty[:, 2]
    pcd = str(tmp_path / "site.pcd")
    _write_pcd(pcd, ["x", "y", "z", "_"], [4, 4, 4, 4], ["F", "F", "F", "F"], [1, 1, 1, 1], a.tobytes(), a.shape[0], "binary")
    goal = scenes.DRIVABLE_GOAL
    start = (-20.0, 10.0, float(0.35 * np.sin(-20.0 / 7.0) + 0.25 * np.cos(10.0 / 5.0)))
    cmd = [exe, pcd, str(P["grid_len"]), str(P["z_len"]), str(P["slope_interval"]), "slope"] + \
          [repr(float(v)) for v in goal] + [repr(float(v)) for v in start] + ["0.25"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"Loaded {dirty.shape[0]} points" in r.stdout and "found the route to goal" in r.stdout
    route = [ln.split()[1:] for ln in r.stdout.splitlines() if ln.startswith("route ")]
    # the oracle on the clean cloud: grid, flood, planner
    ref = oracle.build_grid(cloud, P["grid_len"], P["z_len"], P["slope_interval"], mode=oracle.MODE_INT_SERIAL)
    c = oracle.compute_cost(ref, cloud[0], P["grid_len"], P["z_len"], P["slope_interval"], goal, start=start)
    # Same route as the oracle's planner.  (The GPU's means differ from the fp32 oracle's in the last digits, so an
    # exact tie in A*'s f could in principle be broken differently: then the route must still be a valid one of the
    # same cost.)
    ref_route = [(ref["morton"][row], int(ref["sz"][row])) for row in c["path"]]
    got_route = [(key, int(z)) for key, z, _ in route]
    assert len(got_route) > 10 and got_route[0] == ref_route[0] and got_route[-1] == ref_route[-1]
    h_start, h_ref = float(route[0][2]), float(c["h"][c["path"][0]])
    assert abs(h_start - h_ref) <= 1e-4 * max(1.0, h_ref) and float(route[-1][2]) == 0.0
    if got_route != ref_route:
        assert abs(len(got_route) - len(ref_route)) <= 2
        rows = {(ref["morton"][i], int(ref["sz"][i])): i for i in range(len(ref["sz"]))}
        for a, b in zip(got_route[:-1], got_route[1:]):
            i, j = rows[a], rows[b]
            dx, dy = abs(int(ref["sx"][i]) - int(ref["sx"][j])), abs(int(ref["sy"][i]) - int(ref["sy"][j]))
            assert sorted((min(dx, 2), min(dy, 2))) in ([0, 1], [0, 2])      # one step along one axis (2 = across the seam -1 -> 1)


def test_sharded_example_builds(native_lib):
    _build_example("sharded_build")


@pytest.mark.gpu
def test_sharded_example_one_rank_equals_the_oracle(tmp_path, native_lib):
    """examples/sharded_build.cpp: the sharded build from a plain C++ host (RCCL inside libgndt), here with one rank."""
    from tests import parity
    exe = _build_example("sharded_build")
    cloud = scenes.terrain_cloud(150_000)
    P = dict(grid_len=0.2, z_len=0.2, slope_interval=0.08, demand="slope")
    path = str(tmp_path / "cloud.f32")
    np.ascontiguousarray(cloud, np.float32).tofile(path)
    ref = parity.ref_from_cloud(cloud, P)
    want = f"nodes {ref['num_nodes']} columns {ref['num_columns']} slopes {int(np.count_nonzero(ref['flags'] & 2))}"
    for mode in ("global", "owner"):          # statistics all-reduced / points sent to the owner of their column
        r = subprocess.run([exe, path, str(cloud.shape[0]), "0.2", "0.2", "0.08", "0", "1", str(tmp_path / f"id_{mode}.bin"), "0", mode],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert want in r.stdout, r.stdout
