#!/usr/bin/env python3
"""One rank's share of the owner-partitioned build, on one GPU (RCCL communicator of one rank, no torchrun: can be run under
rocprofv3).  python3 tools/measure_owner.py [--points 12500000] [--steps 10] -> JSON."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=12_500_000)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    import torch
    import grid_ndt_amd as g
    from grid_ndt_amd import scenes
    from grid_ndt_amd.dist import Communicator
    cloud = scenes.terrain_cloud(a.points + 1)
    pts = torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda()
    n = int(pts.shape[0])
    comm = Communicator.single()
    m = g.TwoDmap(0.2, 0.2)
    m.setInterval(0.08)
    m.setCloudFirst(cloud[0])
    plain = g.TwoDmap(0.2, 0.2)
    plain.setInterval(0.08)
    plain.setCloudFirst(cloud[0])
    for _ in range(3):
        m.build_owned(comm, "slope", pts, 0, n)
        plain.create2DMap("slope", pts)
        plain.sync()
    acc = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        _, info = m.build_owned(comm, "slope", pts, 0, n)
        for k, v in info.items():
            if k.endswith("_ms"):
                acc[k] = acc.get(k, 0.0) + v / a.steps
    torch.cuda.synchronize()
    t_owned = (time.perf_counter() - t0) / a.steps * 1e3
    t0 = time.perf_counter()
    for _ in range(a.steps):
        plain.create2DMap("slope", pts)
        plain.sync()
    torch.cuda.synchronize()
    t_plain = (time.perf_counter() - t0) / a.steps * 1e3
    print(json.dumps({"points": n, "nodes": info["global_nodes"], "columns": info["global_columns"],
                      "owned_build_ms": round(t_owned, 4), "plain_build_synced_ms": round(t_plain, 4),
                      "stages_ms": {k: round(v, 4) for k, v in acc.items()},
                      "what": "gndt_build_owned_device with a communicator of ONE rank (split, hand-over to itself, build from records, "
                              "column all-gather, global rows) against gndt_build_device + gndt_sync on the same cloud"}))


if __name__ == "__main__":
    main()
