#!/usr/bin/env python3
"""Generates tests/golden/bridge_ground_expected.npz: what the grid-build path must produce for the reference's own
deterministic scene (src/test/genePcd.cpp:29-199 as restated in grid_ndt_amd/scenes.py, parameters launch/parameters.txt:53-59),
computed HERE with numpy only — independent of oracle/*.cpp and of libgndt — by the rule sheet of SURVEY.md Appendix A:

  A.1  key of a point: fp32 subtract / abs / correctly rounded divide / ceil, 0 -> 1, sign + iff p > o (strict)
  A.2  point 0 is the origin and is not binned; columns in first-seen order, nodes of a column in first-seen order
  A.3  per node with >= 3 points: mean, un-normalised scatter  (here in fp64, two-pass: the exact values the fp32
       reference approximates)
  A.4  lambda_min of the scatter (numpy.linalg.eigvalsh)
  A.5  slope label with the reference's visiting-order rule, evaluated on the fp64 means; `label_margin` is how far the
       closest threshold decision of the node was from flipping (labels are only compared where it exceeds 1e-5)

Round 3 — the fp32 half, so that the oracle's stand-in for PCL is pinned by something that is not the oracle:
  A.3' mean32 / scatter32: what pcl::compute3DCentroid and pcl::computeCovarianceMatrix (call sites include/map2D.h:621-622)
       compute on a dense cloud — the points of a node in ARRIVAL order, running sums in fp32 (numpy.cumsum with dtype
       float32 adds strictly left to right), centroid = sum / float(n), then the six products of the fp32 differences, again
       summed in fp32 in arrival order, not divided by n.
  A.5' flags32: the same visiting-order rule evaluated on the fp32 centroids with fp32 arithmetic (what the reference's
       OcNode::isSlope sees, map2D.h:66-108): compared EVERYWHERE, no margin mask.
and a second scene that is not lattice data: the campus stand-in of BASELINE configs[0] at 100 000 points
(tests/golden/campus_100k_expected.npz: keys, counts, the fp32 fields and labels only, to keep the fixture small), and a
third that is one scan-ordered LiDAR frame of the S3 / S4 terrain (tests/golden/terrain_frame_expected.npz, same fields).

Run from the repository root:  python tests/golden/make_bridge_ground_expected.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def fp32_sequential(pts, inv, cnt, has):
    """mean32 [N,3] and scatter32 [N,6] per node, PCL's dense path: fp32 running sums in arrival order."""
    order = np.argsort(inv, kind="stable")                  # points grouped by node, arrival order kept inside a node
    starts = np.concatenate([[0], np.cumsum(cnt)])
    mean32 = np.zeros((cnt.size, 3), np.float32)
    S32 = np.zeros((cnt.size, 6), np.float32)
    pairs = ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))
    for k in np.flatnonzero(has):
        p = pts[order[starts[k]:starts[k + 1]]]              # fp32 [n, 3]
        # (PCL's accumulators start at +0: adding +0 turns a sum of nothing but -0 terms into +0 and changes nothing else)
        tot = np.cumsum(p, axis=0, dtype=np.float32)[-1] + np.float32(0.0)
        m = tot / np.float32(p.shape[0])                     # fp32 divide
        mean32[k] = m
        d = p - m                                            # fp32
        for j, (a, b) in enumerate(pairs):
            S32[k, j] = np.cumsum(d[:, a] * d[:, b], dtype=np.float32)[-1] + np.float32(0.0)
    return mean32, S32


def evaluate(cloud, P, with_fp64=True):
    demand_true = P.get("demand", "slope") == "true"
    o = cloud[0].astype(np.float32)
    pts = cloud[1:]
    lens = np.float32([P["grid_len"], P["grid_len"], P["z_len"]])
    d = np.abs(pts - o)                                   # fp32
    n = np.ceil(d / lens).astype(np.int64)                # fp32 divide (correctly rounded), ceil
    n[n == 0] = 1
    s = np.where(pts > o, n, -n)                          # signed indices sx, sy, sz
    idx = np.arange(pts.shape[0], dtype=np.int64)
    # nodes: unique (sx, sy, sz); first-seen index = min point index
    key = (s[:, 0] + (1 << 20)) << 43 | (s[:, 1] + (1 << 20)) << 22 | (s[:, 2] + (1 << 21))
    uk, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    first = np.full(uk.size, np.iinfo(np.int64).max)
    np.minimum.at(first, inv, idx)
    col = uk & ~np.int64(0x3FFFFF)
    ucol, cinv = np.unique(col, return_inverse=True)
    cfirst = np.full(ucol.size, np.iinfo(np.int64).max)
    np.minimum.at(cfirst, cinv, first)
    order = np.lexsort((first, cfirst[cinv]))             # column first-seen, then node first-seen
    # fp64 statistics, two-pass
    p64 = pts.astype(np.float64)
    mean = np.zeros((uk.size, 3))
    np.add.at(mean, inv, p64)
    mean /= cnt[:, None]
    dv = p64 - mean[inv]
    prod = np.stack([dv[:, 0] * dv[:, 0], dv[:, 0] * dv[:, 1], dv[:, 0] * dv[:, 2], dv[:, 1] * dv[:, 1], dv[:, 1] * dv[:, 2],
                     dv[:, 2] * dv[:, 2]], 1)
    S = np.zeros((uk.size, 6))
    np.add.at(S, inv, prod)
    has = cnt >= 3
    mean[~has] = 0.0
    S[~has] = 0.0
    M = np.zeros((uk.size, 3, 3))
    M[:, 0, 0], M[:, 0, 1], M[:, 0, 2], M[:, 1, 1], M[:, 1, 2], M[:, 2, 2] = S.T
    M[:, 1, 0], M[:, 2, 0], M[:, 2, 1] = M[:, 0, 1], M[:, 0, 2], M[:, 1, 2]
    lam = np.linalg.eigvalsh(M)[:, 0]
    lam[~has] = 0.0
    # labels, visiting order = `order`
    sx = ((uk >> 43) & 0x1FFFFF) - (1 << 20)
    sy = ((uk >> 22) & 0x1FFFFF) - (1 << 20)
    sz = (uk & 0x3FFFFF) - (1 << 21)
    where = {int(k): i for i, k in enumerate(uk)}
    flags = has.astype(np.uint32)
    margin = np.full(uk.size, np.inf)
    iv = float(np.float32(P["slope_interval"]))
    for i in order:
        if not has[i]:
            continue
        z = int(sz[i])
        up = down = False
        for target, is_up in (((1 if z == -1 else z + 1), True), ((-1 if z == 1 else z - 1), False)):
            j = where.get(int((sx[i] + (1 << 20)) << 43 | (sy[i] + (1 << 20)) << 22 | (target + (1 << 21))))
            if j is None:
                continue
            visited = first[j] < first[i] and has[j]
            oz = mean[j, 2] if visited else 0.0
            diff = abs(oz - mean[i, 2])
            margin[i] = min(margin[i], abs(diff - iv))
            if diff > iv:
                if is_up:
                    up = True
                else:
                    down = True
        if demand_true:                                   # map2D.h:644-659: always a Slope, `down` left as constructed (false)
            flags[i] |= 2
        elif not up:
            flags[i] |= 2
            if down:
                flags[i] |= 4
    # ---- the fp32 half: PCL's dense path and the reference's own label arithmetic ----
    mean32, S32 = fp32_sequential(pts, inv, cnt, has)
    flags32 = has.astype(np.uint32)
    iv32 = np.float32(P["slope_interval"])
    for i in order:
        if not has[i]:
            continue
        z = int(sz[i])
        up = down = False
        for target, is_up in (((1 if z == -1 else z + 1), True), ((-1 if z == 1 else z - 1), False)):
            j = where.get(int((sx[i] + (1 << 20)) << 43 | (sy[i] + (1 << 20)) << 22 | (target + (1 << 21))))
            if j is None:
                continue
            visited = first[j] < first[i] and has[j]
            oz = mean32[j, 2] if visited else np.float32(0.0)
            if np.abs(np.float32(oz - mean32[i, 2])) > iv32:
                if is_up:
                    up = True
                else:
                    down = True
        if demand_true:
            flags32[i] |= 2
        elif not up:
            flags32[i] |= 2
            if down:
                flags32[i] |= 4
    res = dict(sx=sx[order].astype(np.int32), sy=sy[order].astype(np.int32), sz=sz[order].astype(np.int32),
               count=cnt[order].astype(np.uint32), first_idx=first[order].astype(np.uint32),
               mean32=mean32[order], scatter32=S32[order], flags32=flags32[order],
               num_columns=np.int64(ucol.size), params=np.float64([P["grid_len"], P["z_len"], P["slope_interval"]]))
    if with_fp64:
        res.update(mean64=mean[order], scatter64=S[order], lambda_min64=lam[order], flags=flags[order],
                   label_margin=np.minimum(margin[order], 1e9).astype(np.float32))
    return res


def main():
    from grid_ndt_amd import scenes
    here = os.path.dirname(os.path.abspath(__file__))
    for name, cloud, P, full in (("bridge_ground_expected.npz", scenes.bridge_ground(), scenes.BRIDGE_PARAMS, True),
                                 ("campus_100k_expected.npz", scenes.campus_frame(100_000), scenes.CAMPUS_PARAMS, False),
                                 # one scan-ordered LiDAR frame of the S3 / S4 terrain (131 072 points, vegetation: columns of
                                 # several levels, so the up / down comparisons of isSlope are exercised on most columns)
                                 ("terrain_frame_expected.npz", scenes.terrain_frames(1, 5), scenes.TERRAIN_PARAMS, False),
                                 # round 4: demand "true" (every node with statistics becomes a Slope, `down` never assigned,
                                 # map2D.h:644-659) on a terrain cloud at the launch z resolution ...
                                 ("terrain_true_expected.npz", scenes.terrain_cloud(120_000), dict(grid_len=0.2, z_len=0.1, slope_interval=0.08, demand="true"), False),
                                 # ... and the two-storey site with the converters' (0,0,0) padding (15 % of the points in ONE node:
                                 # its fp32 running sums are the reference's own worst case), 0.1 m cubic cells
                                 ("site_zero_padded_expected.npz", scenes.site_two_storey(150_000), dict(grid_len=0.1, z_len=0.1, slope_interval=0.08, demand="slope"), False)):
        res = evaluate(cloud, P, with_fp64=full)
        out = os.path.join(here, name)
        np.savez_compressed(out, **res)
        print(out, os.path.getsize(out), "bytes;", res["sx"].size, "nodes,", int(res["num_columns"]), "columns,",
              int(np.count_nonzero(res["flags32"] & 2)), "slopes (fp32 labels)")


if __name__ == "__main__":
    main()
