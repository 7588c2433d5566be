#!/usr/bin/env python3
"""Measure the cost-map flood (gndt_compute_cost) on one MI355X next to the oracle's CPU executions of
TwoDmap::computeCost, and print a JSON document (kept under profiles/).  Not part of bench.py's contract.

    python tools/measure_cost.py [--points 8000000] > gpurun_out/cost.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=8_000_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--layer-launches-only", action="store_true", help="without the one-workgroup kernel (gndt_debug_set_option)")
    a = ap.parse_args()
    import torch
    import grid_ndt_amd as g
    from oracle import oracle
    from grid_ndt_amd import scenes
    g.build_native()
    if a.layer_launches_only or os.environ.get("GNDT_COST_WG") == "0":       # (the variable is this TOOL's, for tools/ab_cost*.sh; the library reads none)
        g.TwoDmap.set_debug_option(g.TwoDmap.DEBUG_COST_ONE_WORKGROUP, 0)
    out = {"device": g.device_info(0)}

    def one(name, cloud, P, goal, as_shipped):
        m = g.TwoDmap(P["grid_len"], P["z_len"])
        m.setInterval(P["slope_interval"])
        m.setCloudFirst(cloud[0])
        m.create2DMap("slope", torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda())
        cells = m.export()
        if goal is None:
            rows = np.nonzero((cells["flags"] & 2) != 0)[0]
            goal = cells["mean"][rows[len(rows) // 3]]
        st = m.computeCost(goal)
        # the first flood on a NEW map works the per-map tables out (column index, neighbour columns, CostEdge records, collision
        # verdicts); the floods for further goals on it keep them
        pts = torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda()
        first = []
        for _ in range(3):
            m.create2DMap("slope", pts)
            m.sync()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = m.computeCost(goal)
            first.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            st = m.computeCost(goal)
        torch.cuda.synchronize()
        gpu_ms = (time.perf_counter() - t0) / a.steps * 1e3
        got = m.cost_export()
        t0 = time.perf_counter()
        ref = oracle.compute_cost(cells, cloud[0], P["grid_len"], P["z_len"], P["slope_interval"], goal, mode=oracle.COST_FLAGS)
        cpu_flags_s = time.perf_counter() - t0
        rec = {"points": int(cloud.shape[0] - 1), "nodes": int(cells["num_nodes"]), "slopes": int(cells["num_slopes"]),
               "traversable": st["traversable"], "closed": st["closed"], "levels": st["levels"], "ring": st["ring"],
               "gpu_ms": round(gpu_ms, 3), "gpu_first_flood_on_a_map_ms": round(sorted(first)[1], 3),
               "gpu_us_per_level": round(gpu_ms * 1e3 / max(st["levels"], 1), 2),
               "gpu_Mslopes_per_s": round(st["traversable"] / gpu_ms / 1e3, 2),
               "cpu_oracle_flags_ms": round(cpu_flags_s * 1e3, 1),
               "parity_h_bit_exact": bool((got["h"] == ref["h"]).all()), "parity_state_exact": bool((got["state"] == ref["state"]).all())}
        if as_shipped:
            t0 = time.perf_counter()
            oracle.compute_cost(cells, cloud[0], P["grid_len"], P["z_len"], P["slope_interval"], goal, mode=oracle.COST_AS_SHIPPED)
            rec["cpu_oracle_as_shipped_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        out[name] = rec

    one("drivable_site_400k", scenes.drivable_site(), scenes.COST_PARAMS, scenes.DRIVABLE_GOAL, True)
    P = dict(grid_len=0.5, z_len=0.25, slope_interval=0.08)
    one("terrain_2M", scenes.terrain_cloud(2_000_000), P, None, True)
    one(f"terrain_{a.points // 1_000_000}M", scenes.terrain_cloud(a.points), P, None, False)
    # wide layers: an open site of 200 x 200 m on 0.25 m cells (hundreds of thousands of slopes, layers of thousands)
    one("open_site_6M", scenes.drivable_site(6_000_000, half=100.0), dict(grid_len=0.25, z_len=0.25, slope_interval=0.08), scenes.DRIVABLE_GOAL, False)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
