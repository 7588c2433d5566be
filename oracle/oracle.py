"""ctypes front-end of the CPU oracle (oracle/ref_cpu.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under grid_ndt_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force=False):
    """Compile oracle/ref_cpu.cpp -> oracle/liboracle.so (g++, a second or two)."""
    srcs = [os.path.join(_HERE, f) for f in ("ref_cpu.cpp", "cost_cpu.cpp", "ref_codec.hpp", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.oracle_count_morton.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.oracle_count_morton.restype = C.c_int
        L.oracle_morton_to_xy.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oracle_trans_morton_xyz.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_float, C.POINTER(C.c_float),
                                              C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oracle_build.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_float), C.c_float, C.c_float,
                                   C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
        L.oracle_build.restype = C.c_void_p
        L.oracle_num_nodes.argtypes = [C.c_void_p]
        L.oracle_num_nodes.restype = C.c_size_t
        L.oracle_num_columns.argtypes = [C.c_void_p]
        L.oracle_num_columns.restype = C.c_size_t
        L.oracle_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.oracle_max_threads.restype = C.c_int
        L.oracle_export.argtypes = [C.c_void_p] + [C.c_void_p] * 17
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_set_truth.argtypes = [C.c_int]
        L.oracle_compute_cost.argtypes = [C.c_size_t] + [C.c_void_p] * 8 + [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float,
                                          C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_void_p, C.c_int64, C.c_void_p]
        L.oracle_compute_cost.restype = C.c_int
        _lib = L
    return _lib


def count_morton(a, b):
    buf = C.create_string_buffer(16)
    rc = lib().oracle_count_morton(int(a), int(b), buf, 16)
    assert rc == 0
    return buf.value.decode()


def morton_to_xy(m):
    a, b = C.c_int(), C.c_int()
    lib().oracle_morton_to_xy(int(m), C.byref(a), C.byref(b))
    return a.value, b.value


def trans_morton_xyz(origin, grid_len, z_len, p):
    o = (C.c_float * 3)(*[float(v) for v in origin])
    q = (C.c_float * 3)(*[float(v) for v in p])
    buf = C.create_string_buffer(16)
    nx, ny, sz = C.c_int(), C.c_int(), C.c_int()
    lib().oracle_trans_morton_xyz(o, float(grid_len), float(z_len), q, buf, C.byref(nx), C.byref(ny), C.byref(sz))
    return buf.value.decode(), nx.value, ny.value, sz.value


MODE_AS_SHIPPED, MODE_INT_SERIAL, MODE_INT_OPENMP = 0, 1, 2


EIGEN_JACOBI, EIGEN_GENERAL_QR = 0, 1


def set_eigen_solver(which):
    """The fp32 eigen-solve the oracle uses for OcNode::countRoughNormal (map2D.h:110-133): EIGEN_JACOBI (default: cyclic Jacobi on the
    symmetric scatter) or EIGEN_GENERAL_QR (Hessenberg + shifted QR + back-substitution: the route Eigen::EigenSolver takes)."""
    lib().oracle_set_eigen_solver(int(which))


def build_grid(cloud, grid_len, z_len, slope_interval, demand="slope", min_points=3, mode=MODE_AS_SHIPPED,
               threads=0, export=True, truth=None):
    """Run the reference path on `cloud` ([N, 3|4] float32, point 0 = origin and is not binned:
    receiver.cpp:145, 150).  Returns a dict of numpy arrays in the reference's node order."""
    cloud = np.ascontiguousarray(cloud, dtype=np.float32)
    assert cloud.ndim == 2 and cloud.shape[1] in (3, 4) and cloud.shape[0] >= 1
    stride = cloud.shape[1]
    origin = (C.c_float * 3)(*[float(v) for v in cloud[0, :3]])
    body = cloud[1:]
    dem = {"slope": 0, "true": 1}[demand] if isinstance(demand, str) else int(demand)
    L = lib()
    # the fp64 "truth" beside the reference arithmetic is for the parity gates; a timed baseline (export=False) leaves it out
    L.oracle_set_truth(int(export if truth is None else truth))
    h = L.oracle_build(body.ctypes.data if body.size else None, body.shape[0], stride, origin, float(grid_len),
                       float(z_len), float(slope_interval), dem, int(min_points), int(mode), int(threads))
    try:
        n = L.oracle_num_nodes(h)
        td, tc = C.c_double(), C.c_double()
        L.oracle_times(h, C.byref(td), C.byref(tc))
        out = {"num_nodes": n, "num_columns": L.oracle_num_columns(h), "t_division": td.value,
               "t_calculate": tc.value, "num_points": body.shape[0]}
        if export:
            arrs = {
                "sx": np.zeros(n, np.int32), "sy": np.zeros(n, np.int32), "sz": np.zeros(n, np.int32),
                "count": np.zeros(n, np.uint32), "first_idx": np.zeros(n, np.uint64),
                "mean": np.zeros((n, 3), np.float32), "cov": np.zeros((n, 6), np.float32),
                "evals": np.zeros((n, 3), np.float32), "rough": np.zeros(n, np.float32),
                "normal": np.zeros((n, 3), np.float32), "flags": np.zeros(n, np.uint32),
                "mean64": np.zeros((n, 3), np.float64), "cov64": np.zeros((n, 6), np.float64),
                "rough64": np.zeros(n, np.float64), "normal64": np.zeros((n, 3), np.float64),
                "evals64": np.zeros((n, 3), np.float64), "morton": np.zeros((n, 16), np.uint8),
            }
            order = ["sx", "sy", "sz", "count", "first_idx", "mean", "cov", "evals", "rough", "normal", "flags",
                     "mean64", "cov64", "rough64", "normal64", "evals64", "morton"]
            L.oracle_export(h, *[arrs[k].ctypes.data for k in order])
            arrs["morton"] = np.array([bytes(r).split(b"\0")[0].decode() for r in arrs["morton"]], dtype=object)
            out.update(arrs)
        return out
    finally:
        L.oracle_free(h)


def max_threads():
    return lib().oracle_max_threads()


# RobotSphere thresholds (include/robot.h:12, 38-46): radius 0.25 (receiver.cpp:33), reachable height 0.15,
# roughness 100, angle 30 degrees
ROBOT_DEFAULT = dict(radius=0.25, reachable_height=0.15, max_rough=100.0, max_angle_deg=30.0)
COST_AS_SHIPPED, COST_FLAGS = 0, 1


def compute_cost(cells, origin, grid_len, z_len, slope_interval, goal, demand="slope", robot=None, mode=COST_FLAGS, start=None):
    """TwoDmap::computeCost (map2D.h:1285-1397) on an exported grid (`cells`: dict with sx, sy, sz, count, mean,
    normal, rough, flags in reference node order).  Returns dict(rc, h, state, traversable, closed, check_pushes,
    ring, angle_margin_deg, height_margin).  With `start` (xyz) AstarPlanar::findRoute (GlobalPlan.h:49-166) runs
    afterwards and `path` holds the rows of its global_path, start slope first (empty: no route)."""
    rb = dict(ROBOT_DEFAULT)
    rb.update(robot or {})
    n = int(len(cells["sx"]))
    arr = {k: np.ascontiguousarray(cells[k], dtype=t) for k, t in
           (("sx", np.int32), ("sy", np.int32), ("sz", np.int32), ("count", np.uint32), ("mean", np.float32),
            ("normal", np.float32), ("rough", np.float32), ("flags", np.uint32))}
    h = np.zeros(n, np.float32)
    state = np.zeros(n, np.uint8)
    stats = np.zeros(4, np.int64)
    margins = np.zeros(2, np.float64)
    o = (C.c_float * 3)(*[float(v) for v in origin])
    g = (C.c_float * 3)(*[float(v) for v in goal])
    r4 = (C.c_float * 4)(float(rb["radius"]), float(rb["reachable_height"]), float(rb["max_rough"]), float(rb["max_angle_deg"]))
    dem = {"slope": 0, "true": 1}[demand] if isinstance(demand, str) else int(demand)
    path = np.zeros(max(n, 1), np.int32)
    plen = np.zeros(1, np.int64)
    st = (C.c_float * 3)(*[float(v) for v in start]) if start is not None else None
    rc = lib().oracle_compute_cost(n, *[arr[k].ctypes.data for k in ("sx", "sy", "sz", "count", "mean", "normal", "rough", "flags")],
                                   o, float(grid_len), float(z_len), float(slope_interval), dem, g, r4, int(mode),
                                   h.ctypes.data, state.ctypes.data, stats.ctypes.data, margins.ctypes.data,
                                   st, path.ctypes.data, n, plen.ctypes.data)
    return {"rc": rc, "path": path[:int(plen[0])].copy(), "h": h, "state": state, "traversable": int(stats[0]), "closed": int(stats[1]),
            "check_pushes": int(stats[2]), "ring": int(stats[3]), "angle_margin_deg": float(margins[0]),
            "height_margin": float(margins[1])}
