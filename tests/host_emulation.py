"""CPU emulation of libgndt's pipeline built from the SAME arithmetic header the kernels use
(grid_ndt_amd/csrc/gndt_math.hpp via tests/host_math_shim.cpp) plus numpy for the data movement.
It lets the CPU test tier check the kernel maths, the order-free restatement of the slope label and
the parity tolerances against the oracle.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SO = os.path.join(_HERE, "_host_math_shim.so")
_lib = None


def shim():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "host_math_shim.cpp")
        hdrs = [os.path.join(_ROOT, "grid_ndt_amd", "csrc", f) for f in ("gndt_math.hpp", "gndt_cost.hpp")]
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in [src] + hdrs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I",
                                   os.path.join(_ROOT, "grid_ndt_amd", "csrc"), "-o", _SO, src])
        _lib = C.CDLL(_SO)
        _lib.shim_mean_z.restype = C.c_float
        _lib.shim_mean_z.argtypes = [C.c_uint32, C.c_double, C.c_double]
    return _lib


def unpack(keys):
    k = keys.astype(np.uint64)
    sx = ((k >> np.uint64(43)) & np.uint64(0x1FFFFF)).astype(np.int64) - (1 << 20)
    sy = ((k >> np.uint64(22)) & np.uint64(0x1FFFFF)).astype(np.int64) - (1 << 20)
    sz = (k & np.uint64(0x3FFFFF)).astype(np.int64) - (1 << 21)
    return sx.astype(np.int32), sy.astype(np.int32), sz.astype(np.int32)


def point_columns(body, origin, grid_len, z_len):
    """(sx, sy) of every point: the column part of the kernels' own key."""
    L = shim()
    body = np.ascontiguousarray(body, np.float32)
    n, stride = body.shape
    o = (C.c_float * 3)(*[float(v) for v in origin])
    keys = np.zeros(n, np.uint64)
    ok = np.zeros(n, np.uint8)
    L.shim_point_keys(C.c_void_p(body.ctypes.data), C.c_uint64(n), C.c_int(stride), o, C.c_float(grid_len),
                      C.c_float(z_len), C.c_void_p(keys.ctypes.data), C.c_void_p(ok.ctypes.data))
    assert ok.all()
    sx, sy, _ = unpack(keys)
    return sx, sy


def accumulate(body, origin, grid_len, z_len, first_base=0, idx=None):
    """points -> (unique keys, count, first_idx, sums[.,9]) — what k_accumulate leaves in the table.
    `idx`: explicit global index of every point (records of an owner-partitioned build) instead of first_base + position."""
    L = shim()
    body = np.ascontiguousarray(body, np.float32)
    n, stride = body.shape
    o = (C.c_float * 3)(*[float(v) for v in origin])
    keys = np.zeros(n, np.uint64)
    ok = np.zeros(n, np.uint8)
    L.shim_point_keys(C.c_void_p(body.ctypes.data), C.c_uint64(n), C.c_int(stride), o, C.c_float(grid_len),
                      C.c_float(z_len), C.c_void_p(keys.ctypes.data), C.c_void_p(ok.ctypes.data))
    assert ok.all()
    uk, inv = np.unique(keys, return_inverse=True)
    cen = np.zeros((uk.size, 3), np.float64)
    L.shim_centres(C.c_void_p(uk.ctypes.data), C.c_uint64(uk.size), o, C.c_float(grid_len), C.c_float(z_len),
                   C.c_void_p(cen.ctypes.data))
    v = body[:, :3].astype(np.float64) - cen[inv]
    q = np.stack([v[:, 0], v[:, 1], v[:, 2], v[:, 0] * v[:, 0], v[:, 0] * v[:, 1], v[:, 0] * v[:, 2],
                  v[:, 1] * v[:, 1], v[:, 1] * v[:, 2], v[:, 2] * v[:, 2]], 1)
    sums = np.zeros((uk.size, 9), np.float64)
    np.add.at(sums, inv, q)
    count = np.bincount(inv, minlength=uk.size).astype(np.uint32)
    first = np.full(uk.size, np.iinfo(np.int64).max, np.int64)
    np.minimum.at(first, inv, (np.arange(n, dtype=np.int64) + first_base) if idx is None else np.asarray(idx, np.int64))
    return uk, count, first.astype(np.uint32), sums, cen


def finalize(uk, count, first, sums, cen, slope_interval, demand="slope", min_points=3):
    """table contents -> export dict in the reference's order (k_scan/k_label/sort/k_emit)."""
    L = shim()
    n = uk.size
    sx, sy, sz = unpack(uk)
    mean = np.zeros((n, 3), np.float32)
    cov = np.zeros((n, 6), np.float32)
    rough = np.zeros(n, np.float32)
    normal = np.zeros((n, 3), np.float32)
    count = np.ascontiguousarray(count, np.uint32)
    sums = np.ascontiguousarray(sums)
    cen = np.ascontiguousarray(cen)
    L.shim_finalize(C.c_void_p(count.ctypes.data), C.c_void_p(sums.ctypes.data), C.c_void_p(cen.ctypes.data),
                    C.c_uint64(n), C.c_int(min_points), C.c_void_p(mean.ctypes.data), C.c_void_p(cov.ctypes.data),
                    C.c_void_p(rough.ctypes.data), C.c_void_p(normal.ctypes.data))
    has = count >= min_points
    mean_z = np.where(has, mean[:, 2], np.float32(0)).astype(np.float32)
    lookup = {int(k): i for i, k in enumerate(uk)}
    colkey = uk & ~np.uint64(0x3FFFFF)
    col_first = {}
    for i in range(n):
        ck = int(colkey[i])
        f = int(first[i])
        if ck not in col_first or f < col_first[ck]:
            col_first[ck] = f
    flags = has.astype(np.uint32)
    iv = np.float32(slope_interval)

    def pack(x, y, z):
        return ((x + (1 << 20)) << 43) | ((y + (1 << 20)) << 22) | (z + (1 << 21))

    for i in range(n):
        if not has[i]:
            continue
        slope, down = True, False
        if demand == "slope":
            up = False
            z = int(sz[i])
            for target, is_up in ((L.shim_level_above(z), True), (L.shim_level_below(z), False)):
                j = lookup.get(pack(int(sx[i]), int(sy[i]), target))
                if j is None:
                    continue
                visited = first[j] < first[i] and has[j]
                oz = mean_z[j] if visited else np.float32(0)
                if np.abs(np.float32(oz - mean_z[i])) > iv:
                    if is_up:
                        up = True
                    else:
                        down = True
            slope = not up
        if slope:
            flags[i] |= 2
            if down:
                flags[i] |= 4
    cf = np.array([col_first[int(c)] for c in colkey], np.int64)
    order = np.lexsort((first.astype(np.int64), cf))
    out = {"sx": sx[order], "sy": sy[order], "sz": sz[order], "count": count[order], "first_idx": first[order],
           "mean": mean[order], "cov": cov[order], "rough": rough[order], "normal": normal[order], "flags": flags[order],
           "num_nodes": n, "num_columns": len(col_first), "num_slopes": int(np.count_nonzero(flags & 2))}
    return out


def build(cloud, grid_len, z_len, slope_interval, demand="slope"):
    t = accumulate(cloud[1:], cloud[0, :3], grid_len, z_len)
    return finalize(*t, slope_interval, demand)


def cost_levelsync(cells, grid_len, slope_interval, goal_key, demand="slope", robot=None):
    """The level-synchronous flood of gndt_cost.hpp run on the host (same per-slope code as the kernels).
    `goal_key` = (sx, sy, sz) of the goal.  Returns dict(rc, h, state, traversable, closed, check_pushes, ring,
    levels, ring_overflow)."""
    rb = dict(radius=0.25, reachable_height=0.15, max_rough=100.0, max_angle_deg=30.0)
    rb.update(robot or {})
    n = int(len(cells["sx"]))
    arr = {k: np.ascontiguousarray(cells[k], dtype=t) for k, t in
           (("sx", np.int32), ("sy", np.int32), ("sz", np.int32), ("mean", np.float32), ("normal", np.float32),
            ("rough", np.float32), ("flags", np.uint32))}
    h = np.zeros(n, np.float32)
    state = np.zeros(n, np.uint8)
    stats = np.zeros(8, np.int64)
    r4 = (C.c_float * 4)(float(rb["radius"]), float(rb["reachable_height"]), float(rb["max_rough"]), float(rb["max_angle_deg"]))
    L = shim()
    L.shim_cost.restype = C.c_int
    L.shim_cost.argtypes = [C.c_uint64] + [C.c_void_p] * 7 + [C.c_float, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                            C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_void_p]
    dem = {"slope": 0, "true": 1}[demand] if isinstance(demand, str) else int(demand)
    rc = L.shim_cost(n, *[arr[k].ctypes.data for k in ("sx", "sy", "sz", "mean", "normal", "rough", "flags")],
                     float(slope_interval), dem, float(grid_len), int(goal_key[0]), int(goal_key[1]), int(goal_key[2]), r4,
                     h.ctypes.data, state.ctypes.data, stats.ctypes.data)
    return {"rc": rc, "h": h, "state": state, "traversable": int(stats[0]), "closed": int(stats[1]),
            "check_pushes": int(stats[2]), "ring": int(stats[3]), "levels": int(stats[4]), "ring_overflow": int(stats[5]),
            "records": int(stats[6]), "records_more": int(stats[7])}


def collide_all(cells, grid_len, slope_interval, demand="slope", robot=None, ring_cap=1 << 16):
    """CollisionCheck for every slope of `cells`, by the walk over the ring (cost_collide: the restatement of map2D.h:351-474) and by
    the rounds over the whole map the device runs instead (gndt_cost.hpp).  Returns (ring depth, walk, rounds): uint8 arrays, 1
    collide / 0 free / 255 not a slope (walk 2: the ring did not fit ring_cap)."""
    rb = dict(radius=0.25, reachable_height=0.15, max_rough=100.0, max_angle_deg=30.0)
    rb.update(robot or {})
    n = int(len(cells["sx"]))
    arr = {k: np.ascontiguousarray(cells[k], dtype=t) for k, t in
           (("sx", np.int32), ("sy", np.int32), ("sz", np.int32), ("mean", np.float32), ("normal", np.float32),
            ("rough", np.float32), ("flags", np.uint32))}
    walk = np.zeros(n, np.uint8)
    rounds = np.zeros(n, np.uint8)
    r4 = (C.c_float * 4)(float(rb["radius"]), float(rb["reachable_height"]), float(rb["max_rough"]), float(rb["max_angle_deg"]))
    L = shim()
    L.shim_collide_all.restype = C.c_int
    L.shim_collide_all.argtypes = [C.c_uint64] + [C.c_void_p] * 7 + [C.c_float, C.c_int, C.c_float, C.POINTER(C.c_float), C.c_int,
                                   C.c_void_p, C.c_void_p]
    dem = {"slope": 0, "true": 1}[demand] if isinstance(demand, str) else int(demand)
    ring = L.shim_collide_all(n, *[arr[k].ctypes.data for k in ("sx", "sy", "sz", "mean", "normal", "rough", "flags")],
                              float(slope_interval), dem, float(grid_len), r4, int(ring_cap), walk.ctypes.data, rounds.ctypes.data)
    return ring, walk, rounds
