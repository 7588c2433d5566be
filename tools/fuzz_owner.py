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
    a = ap.parse_args()
    import grid_ndt_amd as g
    from tests import parity
    from tests.test_gpu_owner import owner_build_with_threads
    from tools.fuzz_campaign import CELLS, make_cloud
    g.build_native()
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    stats = {"jobs": 0, "ranks": 0, "points": 0, "empty_shards": 0, "failures": []}
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
        try:
            _, one = parity.gpu_from_cloud(cloud, P, on_device=True)
            glob, infos = owner_build_with_threads(cloud, P, W, bounds=bounds)
            bad = [k for k in ("sx", "sy", "sz", "count", "first_idx", "flags") if not np.array_equal(glob[k], one[k])]
            if (glob["num_nodes"], glob["num_columns"], glob["num_slopes"]) != (one["num_nodes"], one["num_columns"], one["num_slopes"]):
                bad.append("totals")
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
