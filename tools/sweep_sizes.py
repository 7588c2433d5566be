#!/usr/bin/env python3
"""Cliff hunt: builds over a grid of cloud sizes and voxel sizes (uniform and LiDAR-ordered terrain), steady-state time
per build, the strategy that ran and the throughput.  Anything that falls back to the atomic path or loses an order of
magnitude against its neighbours is a sizing bug.   python tools/sweep_sizes.py > gpurun_out/sweep.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import grid_ndt_amd as g
    from grid_ndt_amd import scenes
    g.build_native()
    rows = []
    big_u = scenes.uniform_box(30_000_001)
    big_t = scenes.terrain_cloud(30_000_000)
    for scene, cloud in (("uniform", big_u), ("terrain", big_t)):
        for n in (100_000, 1_000_000, 3_000_000, 10_000_000, 30_000_000):
            pts = torch.from_numpy(np.ascontiguousarray(cloud[1:n + 1])).cuda()
            for gl in (0.05, 0.1, 0.2, 0.5, 1.0):
                m = g.TwoDmap(gl, gl)
                m.setInterval(0.08)
                m.setCloudFirst(cloud[0])
                try:
                    for _ in range(3):
                        m.create2DMap("slope", pts)
                        nodes, cols, slopes = m.sync()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(5):
                        m.create2DMap("slope", pts)
                    nodes, cols, slopes = m.sync()
                    dt = (time.perf_counter() - t0) / 5
                    rows.append({"scene": scene, "points": n, "voxel": gl, "nodes": int(nodes), "ms": round(dt * 1e3, 3),
                                 "Mpts_s": round(n / dt / 1e6), "strategy": m.STRATEGY_NAMES[m.last_strategy()]})
                except Exception as e:      # noqa: BLE001
                    rows.append({"scene": scene, "points": n, "voxel": gl, "error": str(e)[:120]})
                del m
            del pts
    print(json.dumps(rows, indent=0))


if __name__ == "__main__":
    main()
