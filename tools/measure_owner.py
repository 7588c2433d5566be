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
    ap.add_argument("--threads", type=int, default=0,
                    help="W > 1: the whole W-rank job on THIS GPU (thread-group communicators, gndt_comm_create_threads): the ranks "
                         "share the GPU, so the wall time is the SUM of their work, not what W GPUs would take")
    a = ap.parse_args()
    if a.threads > 1:
        return threads_job(a)
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


def threads_job(a):
    import threading
    import torch
    import grid_ndt_amd as g
    from grid_ndt_amd import scenes
    from grid_ndt_amd.dist import Communicator
    W = a.threads
    cloud = scenes.terrain_cloud(a.points + 1)
    pts = torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda()
    n = int(pts.shape[0])
    comms = Communicator.threads(W)
    maps = []
    for _ in range(W):
        m = g.TwoDmap(0.2, 0.2)
        m.setInterval(0.08)
        m.setCloudFirst(cloud[0])
        maps.append(m)
    bounds = [n * r // W for r in range(W + 1)]
    infos = [None] * W
    walls = [0.0] * W
    start = threading.Barrier(W)

    def rank(r):
        torch.cuda.set_device(0)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            shard = pts[bounds[r]:bounds[r + 1]]
            for _ in range(3):
                maps[r].build_owned(comms[r], "slope", shard, bounds[r], n, s)
            acc = {}
            start.wait()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                _, info = maps[r].build_owned(comms[r], "slope", shard, bounds[r], n, s)
                for k, v in info.items():
                    if k.endswith("_ms"):
                        acc[k] = acc.get(k, 0.0) + v / a.steps
            walls[r] = (time.perf_counter() - t0) / a.steps * 1e3
            info.update(acc)
            infos[r] = info

    th = [threading.Thread(target=rank, args=(r,)) for r in range(W)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    plain = g.TwoDmap(0.2, 0.2)
    plain.setInterval(0.08)
    plain.setCloudFirst(cloud[0])
    for _ in range(3):
        plain.create2DMap("slope", pts)
        plain.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        plain.create2DMap("slope", pts)
        plain.sync()
    t_plain = (time.perf_counter() - t0) / a.steps * 1e3
    sent = sum(i["bytes_sent"] for i in infos)
    print(json.dumps({"points": n, "ranks_as_threads_on_one_gpu": W, "global_nodes": infos[0]["global_nodes"], "global_columns": infos[0]["global_columns"],
                      "wall_ms_per_build_all_ranks_sharing_the_gpu": round(max(walls), 4), "plain_single_build_ms": round(t_plain, 4),
                      "kept_fraction": round(1.0 - sent / 16.0 / max(1, sum(i["owned_points"] for i in infos)), 4),
                      "owned_points": [i["owned_points"] for i in infos], "bytes_sent_per_rank": [i["bytes_sent"] for i in infos],
                      "stage_ms_rank0": {k: round(v, 4) for k, v in infos[0].items() if k.endswith("_ms")},
                      "what": "gndt_build_owned_device with W thread-group ranks on ONE GPU: exchange volumes, ownership balance and the sum of "
                              "the ranks' work (they time-slice the GPU); not a scaling measurement"}))


if __name__ == "__main__":
    main()
