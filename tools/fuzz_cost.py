#!/usr/bin/env python3
"""Randomised campaign for the cost-map flood (gndt_compute_cost): random clouds and cells (tools/
fuzz_campaign.py's generators), a goal on a random point of the cloud, random robots; h (fp32) and the flood state against the
oracle's sequential restatement of TwoDmap::computeCost on the SAME exported grid, bit for bit — skipped only where the oracle
reports a decision within 1e-3 deg / 1e-6 m of a gate (acosf is the one non-IEEE step).  Test infrastructure.

    python3 tools/fuzz_cost.py [--seconds 300] [--seed 1] [--max-points 600000]   -> JSON summary; exit 1 on a failure"""
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
    ap.add_argument("--max-points", type=int, default=600_000)
    ap.add_argument("--trace", action="store_true", help="print every flood's description to stderr before it runs (to find one that hangs)")
    a = ap.parse_args()
    import torch
    import grid_ndt_amd as g
    from oracle import oracle
    from tools.fuzz_campaign import CELLS, make_cloud
    g.build_native()
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    stats = {"maps": 0, "floods": 0, "compared": 0, "on_a_gate": 0, "by_rc": {}, "traversable": 0, "failures": []}
    while time.time() < t_end and len(stats["failures"]) < 3:
        cells = CELLS[int(rng.integers(0, len(CELLS)))]
        demand = "true" if rng.random() < 0.3 else "slope"
        n = int(np.exp(rng.uniform(np.log(500), np.log(a.max_points))))
        cloud, _ = make_cloud(rng, n, cells)
        m = g.TwoDmap(cells[0], cells[1], strategy=int(rng.choice([0, 1, 2, 5])))
        m.setInterval(0.08)
        m.setCloudFirst(cloud[0])
        m.create2DMap(demand, torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda())
        grid = m.export()
        stats["maps"] += 1
        robot_before = None
        for _ in range(int(rng.integers(1, 5))):
            goal = cloud[int(rng.integers(1, cloud.shape[0]))].astype(np.float32)
            if rng.random() < 0.1:
                goal = goal + np.float32([500.0, 0.0, 0.0])          # off the map
            if robot_before is None or rng.random() < 0.5:       # (half of the floods: the robot of the flood before — its tables are kept)
                robot = {"radius": float(rng.choice([0.2, 0.25, 0.6, 1.3])), "reachable_height": float(rng.choice([0.05, 0.1, 0.3, 0.7])),
                         "max_angle_deg": float(rng.choice([15.0, 20.0, 30.0, 45.0]))}
            else:
                robot = robot_before
            robot_before = robot
            desc = dict(seed=a.seed, map=stats["maps"], cells=cells, demand=demand, points=int(cloud.shape[0] - 1), nodes=int(grid["num_nodes"]),
                        goal=[float(v) for v in goal], robot=robot)
            if a.trace:
                print(json.dumps(desc), file=sys.stderr, flush=True)
            try:
                st = m.computeCost(goal, robot=robot)
                got = m.cost_export()
                ref = oracle.compute_cost(grid, cloud[0], cells[0], cells[1], 0.08, goal, demand=demand, robot=robot)
                stats["floods"] += 1
                stats["by_rc"][str(ref["rc"])] = stats["by_rc"].get(str(ref["rc"]), 0) + 1
                if st["rc"] != ref["rc"]:
                    stats["failures"].append(dict(desc, differs="rc", got=int(st["rc"]), want=int(ref["rc"])))
                    continue
                if ref["rc"] != 0:
                    continue
                if ref["angle_margin_deg"] <= 1e-3 or ref["height_margin"] <= 1e-6:
                    stats["on_a_gate"] += 1
                    continue
                stats["compared"] += 1
                stats["traversable"] += int(ref["traversable"])
                bad = []
                if not np.array_equal(got["h"], ref["h"]):
                    bad.append(f"h differs on {int(np.count_nonzero(got['h'] != ref['h']))} rows")
                if not np.array_equal(got["state"], ref["state"]):
                    bad.append(f"state differs on {int(np.count_nonzero(got['state'] != ref['state']))} rows")
                if (st["traversable"], st["closed"], st["check_pushes"], st["ring"]) != (ref["traversable"], ref["closed"], ref["check_pushes"], ref["ring"]):
                    bad.append(f"counters {(st['traversable'], st['closed'], st['check_pushes'], st['ring'])} != {(ref['traversable'], ref['closed'], ref['check_pushes'], ref['ring'])}")
                if bad:
                    stats["failures"].append(dict(desc, differs=bad, margins=[ref["angle_margin_deg"], ref["height_margin"]]))
            except Exception as e:
                stats["failures"].append(dict(desc, error=f"{type(e).__name__}: {e}"))
        del m
    print(json.dumps(stats, indent=1))
    sys.exit(1 if stats["failures"] else 0)


if __name__ == "__main__":
    main()
