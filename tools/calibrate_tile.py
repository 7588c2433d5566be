#!/usr/bin/env python3
"""Where does the one-pass TILE strategy beat the partition pipeline?  Dense raster clouds with a controlled number of
consecutive points per node (run length), 4 M points each: sampled points-per-partial, ms per build for strategy TILE (5) and
PARTITION (2).  The crossover sets the default of GNDT_DEBUG_TILE_RATIO (gndt_handle.hpp; gndt_debug_set_option).  Run on the GPU box; JSON to stdout."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def raster(n, run, rng):
    """`run` consecutive points per 0.5 m cell along x, rows of 4000 cells, one row after the other."""
    t = np.arange(n)
    per_row = 4000 * run
    x = (t % per_row) * (0.5 / run) + 0.2 * (0.5 / run) * rng.random(n)
    y = (t // per_row) * 0.5 + 0.1 + 0.3 * rng.random(n)
    z = 0.05 * np.sin(x * 0.3) + 0.02 * rng.random(n)
    return np.ascontiguousarray(np.stack([x, y, z], 1).astype(np.float32))


def main():
    import torch
    import grid_ndt_amd as g
    g.build_native()
    rng = np.random.default_rng(1)
    out = []
    for run in (2, 4, 8, 16, 32, 64, 128, 256):
        cloud = raster(4_000_000, run, rng)
        pts = torch.from_numpy(cloud[1:]).cuda()
        row = {"run": run}
        for strat in (5, 2):
            m = g.TwoDmap(0.5, 0.25, strategy=strat)
            m.setInterval(0.08)
            m.setCloudFirst(cloud[0])
            if strat == 5:
                row["points_per_partial"] = round(m.locality_sample(pts), 2)
            for _ in range(3):
                m.create2DMap("slope", pts)
            m.sync()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                m.create2DMap("slope", pts)
            nodes, _, _ = m.sync()
            torch.cuda.synchronize()
            row[m.STRATEGY_NAMES[m.last_strategy()] + "_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 4)
            row["nodes"] = int(nodes)
        out.append(row)
        print(json.dumps(row), file=sys.stderr)
    print(json.dumps({"what": "TILE vs PARTITION on 4 M-point dense rasters, 0.5 / 0.25 m cells", "rows": out}))


if __name__ == "__main__":
    main()
