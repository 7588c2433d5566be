"""tests/golden/bridge_ground_expected.npz — the reference's own deterministic scene (genePcd.cpp) evaluated by a numpy-only
script (tests/golden/make_bridge_ground_expected.py, SURVEY Appendix A rules, fp64 two-pass statistics, numpy eigvalsh):
independent of oracle/*.cpp and of libgndt.  The oracle (CPU tier) and the HIP path (GPU tier) must both reproduce it."""
import os

import numpy as np
import pytest

from grid_ndt_amd import scenes
from tests import parity

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bridge_ground_expected.npz")
GOLD_CAMPUS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "campus_100k_expected.npz")
GOLD_TERRAIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "terrain_frame_expected.npz")
GOLD_TRUE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "terrain_true_expected.npz")
GOLD_SITE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "site_zero_padded_expected.npz")
TRUE_PARAMS = dict(grid_len=0.2, z_len=0.1, slope_interval=0.08, demand="true")
SITE_PARAMS = dict(grid_len=0.1, z_len=0.1, slope_interval=0.08, demand="slope")


def fp32_cases():
    """(golden, cloud, parameters) of every numpy-only fp32 golden: the reference's scene, a non-lattice frame, a LiDAR frame,
    demand "true" (round 4) and the zero-padded site (round 4: 15 % of the points in one node)."""
    return ((GOLD, scenes.bridge_ground(), scenes.BRIDGE_PARAMS), (GOLD_CAMPUS, scenes.campus_frame(100_000), scenes.CAMPUS_PARAMS),
            (GOLD_TERRAIN, scenes.terrain_frames(1, 5), scenes.TERRAIN_PARAMS), (GOLD_TRUE, scenes.terrain_cloud(120_000), TRUE_PARAMS),
            (GOLD_SITE, scenes.site_two_storey(150_000), SITE_PARAMS))


def check_fp32_half(got, gold_path, what, bit_exact_statistics):
    """Round 3: the golden's fp32 half (PCL's dense path restated with numpy float32 running sums in arrival order, labels from
    the fp32 centroids).  The oracle must reproduce mean / scatter BIT FOR BIT and every label; the HIP path — fp64 cell-local
    sums — must give every label and the fp32 values within the north_star tolerance."""
    g = np.load(gold_path)
    n = g["sx"].shape[0]
    assert int(got["num_nodes"]) == n and int(got["num_columns"]) == int(g["num_columns"]), what
    for k in ("sx", "sy", "sz", "count", "first_idx"):
        assert np.array_equal(np.asarray(got[k]).astype(np.int64), g[k].astype(np.int64)), (what, k)
    assert np.array_equal(np.asarray(got["flags"]).astype(np.int64) & 7, g["flags32"].astype(np.int64)), (what, "labels, all nodes")
    mean, cov = np.asarray(got["mean"], np.float32), np.asarray(got["cov"], np.float32)
    if bit_exact_statistics:
        assert np.array_equal(mean.view(np.uint32), g["mean32"].view(np.uint32)), (what, "fp32 centroid bits")
        assert np.array_equal(cov.view(np.uint32), g["scatter32"].view(np.uint32)), (what, "fp32 scatter bits")
    else:
        has = (g["flags32"] & 1) != 0
        assert np.abs(mean[has].astype(np.float64) - g["mean32"][has]).max() <= 1e-5 * max(1.0, float(np.abs(g["mean32"]).max())), what
    return {"nodes": n, "labels_checked": n, "bit_exact_statistics": bool(bit_exact_statistics)}


def check_against_golden(got, what):
    g = np.load(GOLD)
    n = g["sx"].shape[0]
    assert int(got["num_nodes"]) == n and int(got["num_columns"]) == int(g["num_columns"]), what
    for k in ("sx", "sy", "sz"):                                       # keys AND order (columns / nodes first-seen)
        assert np.array_equal(np.asarray(got[k]).astype(np.int64), g[k].astype(np.int64)), (what, k)
    assert np.array_equal(np.asarray(got["count"]).astype(np.int64), g["count"].astype(np.int64)), what
    assert np.array_equal(np.asarray(got["first_idx"]).astype(np.int64), g["first_idx"].astype(np.int64)), what
    gf, f = np.asarray(got["flags"]).astype(np.int64), g["flags"].astype(np.int64)
    assert np.array_equal(gf & 1, f & 1), what                         # has_stats
    sure = g["label_margin"] > 1e-5                                    # label decisions not within 1e-5 of the threshold
    assert sure.mean() > 0.99
    assert np.array_equal((gf & 6)[sure], (f & 6)[sure]), (what, "slope / down labels")
    has = (f & 1) != 0
    mean = np.asarray(got["mean"]).astype(np.float64)[has]
    assert np.abs(mean - g["mean64"][has]).max() <= 1e-5 * max(1.0, np.abs(g["mean64"]).max()), what
    S = g["scatter64"][has]
    scale = np.abs(S).max(axis=1)
    d = np.abs(np.asarray(got["cov"]).astype(np.float64)[has] - S).max(axis=1)
    # fp32 inputs near |p| resolve a scatter only down to count * ulp(p)^2 (lattice points: many nodes are exactly planar)
    floor = g["count"][has].astype(np.float64) * (2.0 ** -23 * np.maximum(np.abs(g["mean64"][has]).max(axis=1), 1e-3)) ** 2
    assert np.all(d <= np.maximum(1e-5 * scale, floor)), (what, float((d / np.maximum(scale, 1e-300)).max()))
    sl = (f & 2) != 0
    rough = np.asarray(got["rough"]).astype(np.float64)[sl]
    lam = g["lambda_min64"][sl]
    tr = g["scatter64"][sl][:, 0] + g["scatter64"][sl][:, 3] + g["scatter64"][sl][:, 5]
    fl = g["count"][sl].astype(np.float64) * (2.0 ** -23 * np.maximum(np.abs(g["mean64"][sl]).max(axis=1), 1e-3)) ** 2
    tol = np.maximum(1e-5 * tr, fl) + 1e-12
    shown_as_001 = rough == np.float64(np.float32(0.01))                # map2D.h:131-132: an exact 0 is displayed as 0.01
    ok = np.where(shown_as_001, (np.abs(lam) <= tol) | (np.abs(lam - 0.01) <= tol), np.abs(rough - lam) <= tol)
    assert np.all(ok), (what, "lambda_min", int(np.count_nonzero(~ok)))
    return {"nodes": n, "labels_checked": int(sure.sum())}


def test_oracle_reproduces_the_numpy_golden():
    """Pins the oracle (all three modes share the arithmetic; mode 0 is the as-shipped containers) to an independent statement."""
    cloud = scenes.bridge_ground()
    ref = parity.ref_from_cloud(cloud, scenes.BRIDGE_PARAMS)
    print(check_against_golden(ref, "oracle mode 0"))


def test_oracle_fp32_statistics_and_labels_are_bit_identical_to_the_numpy_float32_restatement():
    """The oracle's stand-in for pcl::compute3DCentroid / computeCovarianceMatrix (map2D.h:621-622) and for OcNode::isSlope's
    fp32 comparisons, pinned by a restatement that shares no code with it: on the reference's own scene and on a non-lattice one."""
    for gold, cloud, P in fp32_cases():
        for mode in (0, 1, 2):
            ref = parity.ref_from_cloud(cloud, P, mode=mode, threads=3 if mode == 2 else 0)
            print(check_fp32_half(ref, gold, f"oracle mode {mode}", True))


def test_kernel_arithmetic_on_the_host_reproduces_the_numpy_golden():
    from tests import host_emulation as he
    P = scenes.BRIDGE_PARAMS
    emu = he.build(scenes.bridge_ground(), P["grid_len"], P["z_len"], P["slope_interval"], P["demand"])
    print(check_against_golden(emu, "gndt_math.hpp on the host"))


@pytest.mark.gpu
@pytest.mark.parametrize("strategy", [0, 1, 5], ids=["auto", "atomic", "tile"])
def test_hip_path_reproduces_the_numpy_golden(strategy):
    _, out = parity.gpu_from_cloud(scenes.bridge_ground(), scenes.BRIDGE_PARAMS, strategy=strategy)
    print(check_against_golden(out, f"libgndt strategy {strategy}"))
    print(check_fp32_half(out, GOLD, f"libgndt strategy {strategy}", False))
    _, out = parity.gpu_from_cloud(scenes.campus_frame(100_000), scenes.CAMPUS_PARAMS, strategy=strategy)
    print(check_fp32_half(out, GOLD_CAMPUS, f"libgndt strategy {strategy}, campus", False))
    _, out = parity.gpu_from_cloud(scenes.terrain_frames(1, 5), scenes.TERRAIN_PARAMS, strategy=strategy)
    print(check_fp32_half(out, GOLD_TERRAIN, f"libgndt strategy {strategy}, terrain frame", False))


@pytest.mark.gpu
@pytest.mark.parametrize("strategy", [0, 1, 3, 4], ids=["auto", "atomic", "partition_exact", "partition_two_level"])
def test_hip_path_reproduces_the_true_demand_and_zero_padded_goldens(strategy):
    """Round 4: every label of the numpy-only goldens for demand "true" and for the zero-padded site, exactly — no dense-cloud gate,
    no margin mask — and the centroids within the north_star tolerance."""
    for gold, cloud, P in fp32_cases()[3:]:
        _, out = parity.gpu_from_cloud(cloud, P, strategy=strategy)
        print(check_fp32_half(out, gold, f"libgndt strategy {strategy}, {os.path.basename(gold)}", False))
