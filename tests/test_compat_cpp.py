"""include/gndt_compat.hpp: the reference-shaped containers rebuilt from an export.
CPU tier: materialise() from the oracle's export, container invariants (order, keys, Slope fields).
GPU tier: the same program builds through libgndt's C ABI and compares container by container."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from grid_ndt_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_checker(native_lib, name="compat_check"):
    from oracle import oracle
    from grid_ndt_amd import _lib
    oracle.build()
    exe = os.path.join(ROOT, "tests", "cpp", name)
    src = exe + ".cpp"
    deps = [src, os.path.join(ROOT, "include", "gndt_compat.hpp"), os.path.join(ROOT, "include", "gndt.h"), _lib.LIB_PATH]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        csrc = os.path.dirname(_lib.LIB_PATH)
        orc = os.path.join(ROOT, "oracle")
        hip = _lib._hip_runtime_dir()
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-o", exe, src,
                               "-L", csrc, "-l:libgndt.so", "-L", orc, "-l:liboracle.so", "-L", hip, "-l:libamdhip64.so",
                               f"-Wl,-rpath,{csrc}", f"-Wl,-rpath,{orc}", f"-Wl,-rpath,{hip}", "-fopenmp"])
    return exe


def _run(exe, cloud, P, gpu):
    with tempfile.NamedTemporaryFile(suffix=".f32") as f:
        np.ascontiguousarray(cloud, np.float32).tofile(f.name)
        cmd = [exe, f.name, str(cloud.shape[0]), str(P["grid_len"]), str(P["z_len"]), str(P["slope_interval"]), P["demand"]]
        if gpu:
            cmd.append("gpu")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


@pytest.mark.parametrize("name", ["bridge", "campus_true"])
def test_materialise_from_oracle_export(native_lib, name):
    exe = _build_checker(native_lib)
    cloud, P = {"bridge": (scenes.bridge_ground(), scenes.BRIDGE_PARAMS),
                "campus_true": (scenes.campus_frame(60000), dict(scenes.CAMPUS_PARAMS, demand="true"))}[name]
    out = _run(exe, cloud, P, gpu=False)
    assert "oracle->materialise OK" in out


@pytest.mark.gpu
def test_cpp_twodmap_create2dmap_on_gpu(native_lib):
    exe = _build_checker(native_lib)
    for cloud, P in ((scenes.bridge_ground(), scenes.BRIDGE_PARAMS), (scenes.terrain_cloud(300000), dict(grid_len=0.2, z_len=0.2, slope_interval=0.08, demand="slope"))):
        out = _run(exe, cloud, P, gpu=True)
        assert "libgndt create2DMap == oracle OK" in out


# ---- the consumers: computeCost (map2D.h:1285-1397) and AstarPlanar (GlobalPlan.h:49-166) ----
PLAN_CASES = [("slope", (-20.0, 10.0), 0.25), ("slope", (25.0, -20.0), 0.6), ("true", (5.0, 25.0), 0.25)]


def _run_plan(exe, demand, start_xy, radius, gpu):
    cloud = scenes.drivable_site()
    P = scenes.COST_PARAMS
    z = float(0.35 * np.sin(start_xy[0] / 7.0) + 0.25 * np.cos(start_xy[1] / 5.0))
    with tempfile.NamedTemporaryFile(suffix=".f32") as f:
        np.ascontiguousarray(cloud, np.float32).tofile(f.name)
        cmd = [exe, f.name, str(cloud.shape[0]), str(P["grid_len"]), str(P["z_len"]), str(P["slope_interval"]), demand]
        cmd += [repr(float(v)) for v in scenes.DRIVABLE_GOAL] + [repr(start_xy[0]), repr(start_xy[1]), repr(z), repr(radius)]
        if gpu:
            cmd.append("gpu")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


@pytest.mark.parametrize("demand,start_xy,radius", PLAN_CASES)
def test_compat_planner_follows_the_reference_route(native_lib, demand, start_xy, radius):
    exe = _build_checker(native_lib, "plan_check")
    assert "compat A* == oracle OK" in _run_plan(exe, demand, start_xy, radius, gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize("demand,start_xy,radius", PLAN_CASES)
def test_cpp_compute_cost_and_planner_on_gpu(native_lib, demand, start_xy, radius):
    exe = _build_checker(native_lib, "plan_check")
    assert "libgndt computeCost + A* == oracle OK" in _run_plan(exe, demand, start_xy, radius, gpu=True)
