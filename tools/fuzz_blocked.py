#!/usr/bin/env python3
"""Randomised campaign for BLOCKED buckets (gndt_blocked.hpp): dense uniform boxes of random extent, height, cell sizes, density, origin
and demand — some shuffled, some with a zero-padded tail, some followed by a cloud of another extent — built three times on one handle
and once more with another cloud of the same scene; every map against the oracle (dense gates), and which builds took blocked buckets.
Test infrastructure.      python3 tools/fuzz_blocked.py [--seconds 150] [--seed 1]   -> JSON summary; exit 1 on a failure"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=150.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import torch
    import grid_ndt_amd as g
    from tests import parity
    g.build_native()
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    stats = {"cases": 0, "builds": 0, "blocked_builds": 0, "hashed_re_runs_after_a_miss": 0, "nodes": 0, "failures": []}
    while time.time() < t_end and len(stats["failures"]) < 3:
        gl = float(rng.choice([0.1, 0.25, 0.5, 1.0]))
        zl = float(rng.choice([0.1, 0.25, 0.5, 1.0]))
        cols = int(rng.integers(60, 420))                      # columns along x (and about as many along y)
        levels = float(rng.choice([1.5, 3.0, 5.0, 9.0, 20.0, 40.0]))
        per_col = float(rng.choice([6.0, 20.0, 60.0, 150.0]))
        n = int(min(6_000_000, max(1_100_000, cols * cols * per_col)))
        demand = str(rng.choice(["slope", "true"]))
        origin = (rng.random(3) * 10 - 5).astype(np.float32)
        ext = np.float32([cols * gl / 2, cols * gl / 2 * float(rng.choice([0.6, 1.0])), levels * zl / 2])

        def cloud_of(seed, scale=1.0):
            r = np.random.default_rng(seed)
            body = origin + (r.random((n, 3), dtype=np.float32) * 2 - 1) * ext * np.float32(scale)
            return np.concatenate([origin[None, :], body.astype(np.float32)], 0)

        s0 = int(rng.integers(1 << 30))
        c0 = cloud_of(s0)
        if rng.random() < 0.15:
            c0[-int(n * 0.1):] = 0.0                           # the converters' zero padding: one node with a tenth of the points
        P = dict(grid_len=gl, z_len=zl, slope_interval=0.08, demand=demand)
        desc = dict(seed=a.seed, case=stats["cases"], cells=(gl, zl), cols=cols, levels=levels, points=n, demand=demand, cloud_seed=s0)
        try:
            m = g.TwoDmap(gl, zl)
            m.setInterval(0.08)
            m.setCloudFirst(c0[0])
            ref0 = parity.ref_from_cloud(c0, P, mode=2)
            t0 = torch.from_numpy(np.ascontiguousarray(c0[1:])).cuda()
            seq = []
            for k in range(3):
                m.create2DMap(demand, t0)
                out = m.export()
                seq.append(m.STRATEGY_NAMES[m.last_strategy()])
                rep = parity.compare(out, ref0, demand, dense=True, interval=0.08)
                stats["builds"] += 1
                stats["nodes"] += int(out["num_nodes"])
                if not rep["ok"]:
                    raise AssertionError(f"build {k} ({seq}): {rep['fail'][:3]}")
            # another cloud of the scene, or one of another extent (leaves the box: re-run hashed)
            scale = float(rng.choice([1.0, 1.0, 0.7, 1.4]))
            c1 = cloud_of(int(rng.integers(1 << 30)), scale)
            c1[0] = c0[0]
            before = m.retry_count()
            m.create2DMap(demand, torch.from_numpy(np.ascontiguousarray(c1[1:])).cuda())
            out = m.export()
            seq.append(m.STRATEGY_NAMES[m.last_strategy()])
            rep = parity.compare(out, parity.ref_from_cloud(c1, P, mode=2), demand, dense=True, interval=0.08)
            stats["builds"] += 1
            if not rep["ok"]:
                raise AssertionError(f"other cloud x{scale} ({seq}): {rep['fail'][:3]}")
            stats["blocked_builds"] += sum(1 for x in seq if x == "partition_blocked")
            if "partition_blocked" in seq[:3] and seq[3] != "partition_blocked":
                stats["hashed_re_runs_after_a_miss"] += int(m.retry_count() > before)
            del m
        except Exception as e:                                  # noqa: BLE001
            desc["error"] = repr(e)[:400]
            stats["failures"].append(desc)
        stats["cases"] += 1
    stats["seconds"] = a.seconds
    print(json.dumps(stats, indent=1))
    sys.exit(1 if stats["failures"] else 0)


if __name__ == "__main__":
    main()
