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

Run from the repository root:  python tests/golden/make_bridge_ground_expected.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from grid_ndt_amd import scenes
    cloud = scenes.bridge_ground()
    P = scenes.BRIDGE_PARAMS
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
        if not up:
            flags[i] |= 2
            if down:
                flags[i] |= 4
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bridge_ground_expected.npz")
    np.savez_compressed(out, sx=sx[order].astype(np.int32), sy=sy[order].astype(np.int32), sz=sz[order].astype(np.int32),
                        count=cnt[order].astype(np.uint32), first_idx=first[order].astype(np.uint32), mean64=mean[order],
                        scatter64=S[order], lambda_min64=lam[order], flags=flags[order],
                        label_margin=np.minimum(margin[order], 1e9).astype(np.float32),
                        num_columns=np.int64(ucol.size), params=np.float64([P["grid_len"], P["z_len"], P["slope_interval"]]))
    print(out, os.path.getsize(out), "bytes;", uk.size, "nodes,", ucol.size, "columns,", int(np.count_nonzero(flags & 2)), "slopes")


if __name__ == "__main__":
    main()
