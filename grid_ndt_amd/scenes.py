"""Synthetic scenes for bench.py, the tools and the parity tests.

All randomness comes from a counter-based splitmix64 written here, so a scene is a pure function of
(seed, index) and is bit-identical on every box.  Every generator returns a C-contiguous float32
array [N, 3]; point 0 is the cloud's origin by the reference's convention (receiver.cpp:145).
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(counter, seed):
    """splitmix64 output for state = seed + (counter+1)*golden; counter is a uint64 array."""
    with np.errstate(over="ignore"):
        z = (np.asarray(counter, dtype=np.uint64) + np.uint64(1)) * _GOLD + np.uint64(seed)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def u01(counter, seed):
    """24-bit-mantissa uniform in [0,1) as float64 (exactly representable in float32)."""
    return (splitmix64(counter, seed) >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


def normal01(counter, seed):
    a = u01(counter, seed ^ 0xA5A5A5A5)
    b = u01(counter, seed ^ 0x5A5A5A5A5A)
    return np.sqrt(-2.0 * np.log(1.0 - a)) * np.cos(2.0 * np.pi * b)


# --------------------------------------------------------------------------------------------------
# S2: N i.i.d. uniform points in [-100,100)^2 x [-1,1)  (BASELINE.json configs[1])
# --------------------------------------------------------------------------------------------------
def uniform_box(n, seed=0x5EED0002, half_xy=100.0, half_z=1.0, chunk=1 << 22):
    out = np.empty((n, 3), np.float32)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        i = np.arange(s, e, dtype=np.uint64)
        out[s:e, 0] = (u01(i * np.uint64(3), seed) * 2.0 - 1.0) * half_xy
        out[s:e, 1] = (u01(i * np.uint64(3) + np.uint64(1), seed) * 2.0 - 1.0) * half_xy
        out[s:e, 2] = (u01(i * np.uint64(3) + np.uint64(2), seed) * 2.0 - 1.0) * half_z
    return out


# --------------------------------------------------------------------------------------------------
# bridge_ground: the reference's own deterministic scene, restated as a table of lattice sweeps
# (src/test/genePcd.cpp:29-199).  Loop counters are float, the step and the bounds are double, the
# cloud is pre-allocated 600 x 600 and the unfilled tail stays (0,0,0) (genePcd.cpp:30-33).
# Parameters of the scene: launch/parameters.txt:53-59 (res 0.1, z_res 0.05, interval 0.08).
# --------------------------------------------------------------------------------------------------
def _sweep(lo, hi, inclusive):
    vals = []
    v = np.float32(lo)
    while (float(v) <= hi) if inclusive else (float(v) < hi):
        vals.append(v)
        v = np.float32(float(v) + 0.025)
    return np.array(vals, np.float32)


def bridge_ground(pad_to=360000):
    parts = []

    def plane_xy(xs, ys, zfun):
        X, Y = np.meshgrid(xs, ys, indexing="ij")
        Z = zfun(X.astype(np.float64)).astype(np.float32)
        parts.append(np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1))

    def wall_y(xs, y, zs):  # x outer, z inner
        X, Z = np.meshgrid(xs, zs, indexing="ij")
        parts.append(np.stack([X.ravel(), np.full(X.size, y, np.float32), Z.ravel()], 1))

    def wall_x(x, ys, zs):  # y outer, z inner
        Y, Z = np.meshgrid(ys, zs, indexing="ij")
        parts.append(np.stack([np.full(Y.size, x, np.float32), Y.ravel(), Z.ravel()], 1))

    # ramps a, b (genePcd.cpp:37-52)
    plane_xy(_sweep(1, 5.025, True), _sweep(1, 5, True), lambda x: x * 0.5 + 0.5)
    plane_xy(_sweep(np.float32(11 - 0.025), 15, True), _sweep(1, 5, True), lambda x: x * (-0.5) + 8.5)
    # deck c, d (:54-61)
    plane_xy(_sweep(5, 11, False), _sweep(1, 5, False), lambda x: np.full_like(x, 3.0))
    # ground with holes (:71-128)
    one = lambda x: np.full_like(x, 1.0)
    plane_xy(_sweep(0, 5, True), _sweep(0, 6, True), one)
    plane_xy(_sweep(6, 10, True), _sweep(0, 6, True), one)
    plane_xy(_sweep(11, 16, True), _sweep(0, 6, True), one)
    plane_xy(_sweep(5, 6, True), _sweep(0, 1, True), one)
    plane_xy(_sweep(5, 6, True), _sweep(5, 6, True), one)
    plane_xy(_sweep(10, 11, True), _sweep(0, 1, True), one)
    plane_xy(_sweep(10, 11, True), _sweep(5, 6, True), one)
    # wall panels e, k, f, l (:131-162) and g, h, i, j (:165-196)
    zs = _sweep(1, 3, True)
    wall_y(_sweep(5, 6, True), np.float32(1), zs)
    wall_y(_sweep(10, 11, True), np.float32(1), zs)
    wall_y(_sweep(5, 6, True), np.float32(5), zs)
    wall_y(_sweep(10, 11, True), np.float32(5), zs)
    ys = _sweep(1, 5, True)
    for x in (5, 6, 10, 11):
        wall_x(np.float32(x), ys, zs)
    pts = np.concatenate(parts, 0).astype(np.float32)
    if pad_to and pad_to > pts.shape[0]:
        pts = np.concatenate([pts, np.zeros((pad_to - pts.shape[0], 3), np.float32)], 0)
    return np.ascontiguousarray(pts)


BRIDGE_PARAMS = dict(grid_len=0.1, z_len=0.05, slope_interval=0.08, demand="slope")


# --------------------------------------------------------------------------------------------------
# value-noise heightfield shared by S1 / S3 / S4
# --------------------------------------------------------------------------------------------------
def _lattice(ix, iy, octave, seed):
    key = (ix.astype(np.int64) & 0xFFFFF).astype(np.uint64) | ((iy.astype(np.int64) & 0xFFFFF).astype(np.uint64) << np.uint64(20)) \
        | (np.uint64(octave) << np.uint64(40))
    return u01(key, seed) * 2.0 - 1.0


def heightfield(x, y, seed, amplitude=10.0, wavelength=80.0, octaves=4):
    h = np.zeros_like(x, dtype=np.float64)
    for o in range(octaves):
        w = wavelength / (1 << o)
        a = amplitude / (1 << o)
        fx, fy = x / w, y / w
        ix, iy = np.floor(fx), np.floor(fy)
        tx, ty = fx - ix, fy - iy
        tx = tx * tx * (3 - 2 * tx)
        ty = ty * ty * (3 - 2 * ty)
        ix, iy = ix.astype(np.int64), iy.astype(np.int64)
        v00 = _lattice(ix, iy, o, seed)
        v10 = _lattice(ix + 1, iy, o, seed)
        v01 = _lattice(ix, iy + 1, o, seed)
        v11 = _lattice(ix + 1, iy + 1, o, seed)
        h += a * ((v00 * (1 - tx) + v10 * tx) * (1 - ty) + (v01 * (1 - tx) + v11 * tx) * ty)
    return h


# --------------------------------------------------------------------------------------------------
# S3 / S4: outdoor terrain seen by a moving spinning LiDAR, emitted pose-major, ring-major, azimuth
# order.  `first_pose`/`n_poses` select frames (S4 streams them one by one); S3 is the concatenation.
# --------------------------------------------------------------------------------------------------
RINGS, AZ = 64, 2048
FRAME_POINTS = RINGS * AZ  # 131072


def _pose_xy(p, half, lane):
    """Boustrophedon sweep over [-half, half)^2 with `lane` metres between poses and rows."""
    per_row = max(1, int((2 * half) // lane))
    row, col = p // per_row, p % per_row
    col = np.where(row % 2 == 0, col, per_row - 1 - col)
    row = row % per_row
    return -half + (col + 0.5) * lane, -half + (row + 0.5) * lane


def terrain_frames(n_poses, first_pose=0, seed=0x5EED0003, half=200.0, lane=14.0, max_range=50.0,
                   points_per_frame=FRAME_POINTS, veg_fraction=0.05):
    out = np.empty((n_poses * points_per_frame, 3), np.float32)
    j = np.arange(points_per_frame, dtype=np.int64)
    ring, az = j // AZ, j % AZ
    # ground-hit radius grows geometrically with the ring index: 2 m .. max_range
    rho0 = 2.0 * (max_range / 2.0) ** (ring / (RINGS - 1.0))
    for f in range(n_poses):
        p = first_pose + f
        px, py = _pose_xy(np.int64(p), half, lane)
        ctr = (np.uint64(p) * np.uint64(points_per_frame) + j.astype(np.uint64)) * np.uint64(4)
        rho = rho0 * (1.0 + 0.02 * (u01(ctr, seed) - 0.5))
        ang = (az + u01(ctr + np.uint64(1), seed)) * (2.0 * np.pi / AZ)
        x = np.clip(px + rho * np.cos(ang), -half, np.nextafter(np.float32(half), np.float32(0)))
        y = np.clip(py + rho * np.sin(ang), -half, np.nextafter(np.float32(half), np.float32(0)))
        z = heightfield(x, y, seed) + 0.03 * normal01(ctr + np.uint64(2), seed)
        veg = u01(ctr + np.uint64(3), seed)
        z = z + np.where(veg < veg_fraction, 3.0 * veg / veg_fraction, 0.0)
        s = f * points_per_frame
        out[s:s + points_per_frame, 0] = x
        out[s:s + points_per_frame, 1] = y
        out[s:s + points_per_frame, 2] = z
    return out


TERRAIN_PARAMS = dict(grid_len=0.2, z_len=0.2, slope_interval=0.08, demand="slope")   # BASELINE configs[2] / [3]: 0.2 m cubic voxels


def terrain_cloud(n, seed=0x5EED0003, **kw):
    """S3: the first n points of the pose stream."""
    ppf = kw.get("points_per_frame", FRAME_POINTS)
    poses = (n + ppf - 1) // ppf
    return np.ascontiguousarray(terrain_frames(poses, 0, seed, **kw)[:n])


# --------------------------------------------------------------------------------------------------
# S1: single ~200 k frame, campus-like (freiburg2_16 stand-in; dataset unavailable, README.md:18)
# --------------------------------------------------------------------------------------------------
def campus_frame(n=200000, seed=0x5EED0001):
    i = np.arange(n, dtype=np.uint64) * np.uint64(4)
    # scan-ordered: x sweeps slowly, y quickly (rows of a raster), then jitter
    rows = int(np.sqrt(n))
    r, c = (np.arange(n) // rows), (np.arange(n) % rows)
    x = -10.0 + 100.0 * (r + u01(i, seed)) / max(1, (n + rows - 1) // rows)
    y = -70.0 + 100.0 * (c + u01(i + np.uint64(1), seed)) / rows
    z = 0.6 * np.sin(x / 15.0) + 0.4 * np.cos(y / 11.0) + 0.02 * normal01(i + np.uint64(2), seed)
    wall = u01(i + np.uint64(3), seed)
    z = np.where(wall < 0.10, 3.0 * wall / 0.10, z)
    x = np.where(wall < 0.10, np.round(x / 12.5) * 12.5, x)  # walls on a few vertical planes
    return np.ascontiguousarray(np.stack([x, y, z], 1).astype(np.float32))


CAMPUS_PARAMS = dict(grid_len=0.5, z_len=0.1, slope_interval=0.08, demand="slope")  # parameters.txt:46-51


# --------------------------------------------------------------------------------------------------
# S5: two-storey site (site125 stand-in), 15 % of the points left at (0,0,0) like the converters'
# pre-allocated clouds (pose2pcd.cpp:143-146 etc.)
# --------------------------------------------------------------------------------------------------
def site_two_storey(n=20_000_000, seed=0x5EED0005, zero_fraction=0.15, chunk=1 << 22):
    out = np.zeros((n, 3), np.float32)
    n_real = int(n * (1.0 - zero_fraction))
    for s in range(0, n_real, chunk):
        e = min(n_real, s + chunk)
        i = np.arange(s, e, dtype=np.uint64) * np.uint64(5)
        # raster order over a 60 m x 60 m footprint
        t = (np.arange(s, e) + 0.5) / n_real
        x = -30.0 + 60.0 * t + 0.05 * (u01(i, seed) - 0.5)
        y = -30.0 + 60.0 * u01(i + np.uint64(1), seed)
        kind = u01(i + np.uint64(2), seed)
        floor0 = -0.4 + 0.01 * normal01(i + np.uint64(3), seed)
        floor1 = 0.05 + 2.6 + 0.01 * normal01(i + np.uint64(3), seed)
        ramp = -0.4 + 2.6 * np.clip((y + 10.0) / 20.0, 0.0, 1.0)
        on_ramp = (np.abs(x) < 2.0)
        z = np.where(kind < 0.45, floor0, np.where(kind < 0.85, np.where(np.abs(x) > 6.0, floor1, floor0), 0.0))
        z = np.where(on_ramp, ramp + 0.01 * normal01(i + np.uint64(4), seed), z)
        wall = kind >= 0.85
        z = np.where(wall, -0.4 + 5.6 * u01(i + np.uint64(4), seed), z)
        xw = np.round(x / 10.0) * 10.0
        x = np.where(wall, xw, x)
        out[s:e, 0], out[s:e, 1], out[s:e, 2] = x, y, z
    # origin must be a real point: keep point 0 real, zeros are the tail (as in the converters)
    return out


# --------------------------------------------------------------------------------------------------
# Cost-map scene: gently rolling ground with everything the flood has to react to —
# kerbs higher than the robot's reachable height (0.15 m, robot.h:38), a raised platform reached by a
# ramp, free-standing walls, overhanging slabs 0.3 m above the ground (closer than the robot's diameter:
# collision), and a sparsely sampled region whose normals are noisy (angle gate).  The area straddles the
# origin so the flood crosses all four quadrant seams (countLRFB, map2D.h:226-255).
# --------------------------------------------------------------------------------------------------
COST_PARAMS = dict(grid_len=0.5, z_len=0.25, slope_interval=0.08, demand="slope")
DRIVABLE_GOAL = (-5.2, -3.3, float(0.35 * np.sin(-5.2 / 7.0) + 0.25 * np.cos(-3.3 / 5.0)))


def drivable_site(n=400000, seed=0x5EED0006, half=30.0):
    i = np.arange(n, dtype=np.uint64)
    x = (u01(i, seed) * 2 - 1) * half
    y = (u01(i, seed + 1) * 2 - 1) * half
    sel = u01(i, seed + 2)
    # thin the north-east corner: few points per cell -> noisy normals, some cells without statistics
    thin = (x > 0.5 * half) & (y > 0.5 * half) & (sel > 0.08)
    z = 0.35 * np.sin(x / 7.0) + 0.25 * np.cos(y / 5.0) + 0.004 * normal01(i, seed + 3)
    # platform (+0.45 m) with a ramp on its west side
    plat = (x > 8) & (x < 16) & (y > -6) & (y < 4)
    z = np.where(plat, z + 0.45, z)
    ramp = (x > 3) & (x <= 8) & (y > -3) & (y < 1)
    z = np.where(ramp, z + 0.45 * (x - 3) / 5.0, z)
    # kerb: a 0.3 m step along y = -12 for x < 0
    z = np.where((y < -12) & (x < 0), z - 0.3, z)
    # walls: points spread over 0..2 m above the ground in two strips
    wall = ((np.abs(x + 10) < 0.3) & (y > 0) & (y < 15)) | ((np.abs(y - 18) < 0.3) & (x > -20) & (x < 5))
    z = np.where(wall & (sel < 0.7), z + 2.0 * u01(i, seed + 4), z)
    # overhanging slab 0.3 m above the ground (a second surface in the same columns)
    slab = (x > -25) & (x < -18) & (y > -8) & (y < -2)
    z = np.where(slab & (sel < 0.5), z + 0.3, z)
    keep = ~thin
    pts = np.stack([x, y, z], 1)[keep].astype(np.float32)
    origin = np.array([[0.013, -0.021, float(0.35 * np.sin(0.013 / 7.0) + 0.25 * np.cos(-0.021 / 5.0))]], np.float32)
    return np.concatenate([origin, pts], 0)


def with_stride4(cloud):
    """pcl::PointXYZ layout: 16-byte points, 4th float is padding (set to 1.0 like PCL's data[3])."""
    out = np.ones((cloud.shape[0], 4), np.float32)
    out[:, :3] = cloud
    return out


# --------------------------------------------------------------------------------------------------
# S1 (second stand-in): one depth-camera frame — what BASELINE configs[0]'s freiburg2 .pcd files are (an organised 640 x 480
# cloud of an office, NaN pixels removed, row-major order kept).  At the reference's launch cells (0.5 / 0.1 m) such a frame
# has a few hundred nodes of hundreds of points each: the regime of dense cells and hot columns, unlike the campus raster above.
# --------------------------------------------------------------------------------------------------
DEPTH_PARAMS = dict(grid_len=0.5, z_len=0.1, slope_interval=0.08, demand="slope")  # parameters.txt:46-51


def depth_frame(seed=0x5EED0006, holes=0.3):
    rng = np.random.default_rng(seed)
    W, H = 640, 480
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    dx, dy = (u - 319.5) / 525.0, (v - 239.5) / 525.0               # pinhole, looking along +y (x right, z up)
    below = dy > 1e-3
    t_floor = np.where(below, 1.2 / np.maximum(dy, 1e-3), np.inf)   # floor 1.2 m under the camera
    t_desk = np.where(below, 0.5 / np.maximum(dy, 1e-3), np.inf)    # a desk top 0.5 m under it, 1 .. 2 m away, 1.6 m wide
    t_desk = np.where((t_desk > 1.0) & (t_desk < 2.0) & (np.abs(dx * t_desk) < 0.8), t_desk, np.inf)
    t = np.minimum(np.minimum(t_floor, 3.5), t_desk)                 # back wall at 3.5 m
    t = t * (1.0 + 0.002 * rng.standard_normal(t.shape))
    pts = np.stack([dx * t, t, -dy * t], -1).reshape(-1, 3).astype(np.float32)
    keep = rng.random(pts.shape[0]) > holes                          # invalid pixels dropped, order kept
    return np.ascontiguousarray(pts[keep])
