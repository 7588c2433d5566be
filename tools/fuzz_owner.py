#!/usr/bin/env python3
"""Randomised campaign for the owner-partitioned build itself (gndt_build_owned_device with W thread-group ranks on ONE GPU):
random clouds (tools/fuzz_campaign.py's generators), world sizes 1 .. 8, shard boundaries with empty and tiny shards, cells; the
assembled map must be the single-GPU map of the same cloud row for row (keys, order, counts, first-seen indices, labels) with the
statistics inside the parity tolerances, then gathered on a random root it must export the same.  Test infrastructure.

    python3 tools/fuzz_owner.py [--seconds 300] [--seed 1] [--max-points 3000000]   -> JSON summary; exit 1 on a failure
    (a rank that raises ends the process with code 3 on purpose: the others would wait for it at a barrier)"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-points", type=int, default=3_000_000)
    ap.add_argument("--only-job", type=int, default=-1, help="replay the random sequence and run only this job (verbose)")
    a = ap.parse_args()
    import grid_ndt_amd as g
    from tests import parity
    import torch
    from grid_ndt_amd.dist import Communicator
    from tests.test_gpu_owner import _ranks, _threads, global_build_with_threads, owner_build_with_threads

    def gathered(cloud, P, W, bounds, root):
        maps, comms = _ranks(cloud, P, W), Communicator.threads(W)
        pts = torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda()
        n = int(pts.shape[0])

        def rank(r):
            torch.cuda.set_device(0)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                maps[r].build_owned(comms[r], "slope", pts[bounds[r]:bounds[r + 1]], bounds[r], n, s)
                maps[r].gather_owned(comms[r], root, s)
                return maps[r].export()

        res, errs = _threads(W, rank)
        for c in comms:
            c.close()
        if errs:
            raise errs[0][1]
        return res[max(root, 0)]
    from tools.fuzz_campaign import CELLS, make_cloud
    g.build_native()
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    stats = {"jobs": 0, "ranks": 0, "points": 0, "empty_shards": 0, "modes": {}, "failures": []}
    while time.time() < t_end and len(stats["failures"]) < 3:
        cells = CELLS[int(rng.integers(0, len(CELLS)))]
        P = dict(grid_len=cells[0], z_len=cells[1], slope_interval=0.08, demand="slope")
        n = int(np.exp(rng.uniform(np.log(50), np.log(a.max_points))))
        cloud, _ = make_cloud(rng, n, cells)
        nb = cloud.shape[0] - 1
        W = int(rng.integers(1, 9))
        if rng.random() < 0.5:
            bounds = [0] + np.sort(rng.integers(0, nb + 1, size=W - 1)).tolist() + [nb]
        else:
            bounds = [nb * r // W for r in range(W + 1)]
        desc = dict(job=stats["jobs"], seed=a.seed, cells=cells, points=nb, W=W, bounds=bounds)
        if a.only_job >= 0 and stats["jobs"] != a.only_job:          # the same draws, no work
            mode = ("owner", "global", "gather")[int(rng.integers(0, 3))]
            if mode == "global":
                rng.integers(0, W)
            elif mode == "gather":
                rng.integers(-1, W)
            stats["jobs"] += 1
            if stats["jobs"] > a.only_job:
                break
            continue
        try:
            _, one = parity.gpu_from_cloud(cloud, P, on_device=True)
            mode = ("owner", "global", "gather")[int(rng.integers(0, 3))]
            desc["mode"] = mode
            stats["modes"][mode] = stats["modes"].get(mode, 0) + 1
            if mode == "global":                   # gndt_build_global_device: the whole map on every rank
                outs = global_build_with_threads(cloud, P, W, bounds)
                glob = outs[int(rng.integers(0, W))]
                for o in outs:
                    if not all(np.array_equal(o[k], outs[0][k]) for k in ("sx", "sy", "sz", "count", "first_idx", "flags", "mean", "cov")):
                        stats["failures"].append(dict(desc, differs=["ranks disagree"]))
            elif mode == "gather":                 # owner build + gndt_gather_owned_map_device to a random root (or to everybody)
                glob = gathered(cloud, P, W, bounds, int(rng.integers(-1, W)))
            else:
                glob, infos = owner_build_with_threads(cloud, P, W, bounds=bounds)
            if (glob["num_nodes"], glob["num_columns"], glob["num_slopes"]) != (one["num_nodes"], one["num_columns"], one["num_slopes"]):
                stats["failures"].append(dict(desc, differs=["totals"], got=[int(glob[k]) for k in ("num_nodes", "num_columns", "num_slopes")],
                                              want=[int(one[k]) for k in ("num_nodes", "num_columns", "num_slopes")]))
                if a.only_job >= 0:
                    np.save("gpurun_out/fuzz_owner_cloud.npy", cloud)
                    kg = set(zip(glob["sx"].tolist(), glob["sy"].tolist(), glob["sz"].tolist()))
                    miss = [(int(x), int(y), int(z), int(c), int(f)) for x, y, z, c, f in zip(one["sx"], one["sy"], one["sz"], one["count"], one["first_idx"]) if (x, y, z) not in kg]
                    print("missing nodes (sx, sy, sz, count, first_idx):", miss[:20], file=sys.stderr)
                stats["jobs"] += 1
                continue
            bad = [k for k in ("sx", "sy", "sz", "count", "first_idx", "flags") if not np.array_equal(glob[k], one[k])]
            scale = np.abs(one["cov"]).max(axis=1, keepdims=True) + 1e-30
            if one["num_nodes"] and (np.abs(glob["cov"] - one["cov"]) / scale).max() >= 1e-5:
                # (sub-resolution scatters: compare against the fp64-accumulated single-GPU value in absolute terms as well)
                d = np.abs(glob["cov"] - one["cov"]).max(axis=1)
                pm = np.maximum(np.abs(one["mean"]).max(axis=1), 1e-3)
                floor = one["count"] * (2.0 ** -23 * pm) ** 2
                if np.any(d > np.maximum(1e-5 * scale[:, 0], floor)):
                    bad.append("cov")
            if one["num_nodes"] and not np.allclose(glob["mean"], one["mean"], rtol=0, atol=2e-6 * max(1.0, float(np.abs(one["mean"]).max()))):
                bad.append("mean")
            if bad:
                stats["failures"].append(dict(desc, differs=bad))
        except Exception as e:
            stats["failures"].append(dict(desc, error=f"{type(e).__name__}: {e}"))
        stats["jobs"] += 1
        stats["ranks"] += W
        stats["points"] += nb
        stats["empty_shards"] += sum(1 for r in range(W) if bounds[r + 1] == bounds[r])
    print(json.dumps(stats, indent=1))
    sys.exit(1 if stats["failures"] else 0)


if __name__ == "__main__":
    main()
