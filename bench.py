#!/usr/bin/env python3
"""bench.py — NDT grid-build throughput (bin + mean/cov + eigen + labels + ordering) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one full build of the map from a device-resident cloud (`create2DMap`).  At N = 1 the
workload is BASELINE.json configs[1]: 10 M uniform-random points, 0.5 m cubic voxels.
At N > 1 every rank holds its own 10 M-point shard (weak scaling); see --mode.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
BYTES_PER_POINT = 12           # one read of packed fp32 xyz          
BYTES_PER_NODE = 76            # one write of the node's result row   


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=10_000_000, help="points per rank")
    ap.add_argument("--grid-len", type=float, default=0.5)
    ap.add_argument("--z-len", type=float, default=0.5)
    ap.add_argument("--strategy", type=int, default=0)
    ap.add_argument("--mode", choices=["replicas", "global"], default="replicas",
                    help="N>1: independent per-rank maps (no collective) or one global map via RCCL stats exchange")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000)
    ap.add_argument("--check", action="store_true", help="also check a 300 k-point build against the oracle")
    ap.add_argument("--stamps", action="store_true",
                    help="diagnostic: after the timed run, one extra build with in-kernel phase stamps (stderr)")
    ap.add_argument("--nodes-hint", type=int, default=1 << 20)
    return ap.parse_args()


def cpu_baseline(cloud, P, sample):
    """Oracle (faithful restatement of the reference path) timed on this box's host cores, on the
    first `sample` points of the same workload.  Reported, never the thing measured as `value`."""
    from oracle import oracle
    sub = np.ascontiguousarray(cloud[:sample + 1])
    threads = oracle.max_threads()
    t0 = time.perf_counter()
    r = oracle.build_grid(sub, P["grid_len"], P["z_len"], P["slope_interval"], "slope", mode=oracle.MODE_INT_OPENMP,
                          threads=threads, export=False)
    dt = r["t_division"] + r["t_calculate"]
    wall = time.perf_counter() - t0
    n_ser = min(sample, 200_000)
    r0 = oracle.build_grid(np.ascontiguousarray(cloud[:n_ser + 1]), P["grid_len"], P["z_len"], P["slope_interval"], "slope",
                           mode=oracle.MODE_AS_SHIPPED, export=False)
    dt0 = r0["t_division"] + r0["t_calculate"]
    return {"value": round(sample / dt / 1e6, 4), "unit": "Mpoints/s", "cores": threads, "kind": "port",
            "sample": f"first {sample} points of the workload, oracle mode 2 (OpenMP, integer keys); "
                      f"division {r['t_division']:.2f}s + calculate {r['t_calculate']:.2f}s (wall {wall:.1f}s)",
            "as_shipped_serial": {"value": round(n_ser / dt0 / 1e6, 4), "unit": "Mpoints/s", "cores": 1,
                                  "sample": f"first {n_ser} points, oracle mode 0 (strings + multimap, as the reference runs)"}}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (libgndt has no CPU path)"
    torch.cuda.set_device(local)
    use_dist = world > 1 or ("RANK" in os.environ and a.mode == "global")   # torchrun with one rank exercises the exchange too
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    import grid_ndt_amd as g
    from tests import scenes
    g.build_native()
    P = dict(grid_len=a.grid_len, z_len=a.z_len, slope_interval=0.08, demand="slope")
    n = a.points
    cloud = scenes.uniform_box(n + 1, seed=0x5EED0002 + rank)     # point 0 = origin (receiver.cpp:145)
    global_mode = use_dist and a.mode == "global"
    if global_mode:
        origin = scenes.uniform_box(1, seed=0x5EED0002)[0]
    else:
        origin = cloud[0]
    dev = torch.device(f"cuda:{local}")
    pts = torch.from_numpy(cloud[1:]).to(dev)
    torch.cuda.synchronize()

    m = g.TwoDmap(P["grid_len"], P["z_len"], device=local, max_nodes_hint=a.nodes_hint, strategy=a.strategy)
    m.setInterval(P["slope_interval"])
    m.setCloudFirst(origin)
    stream = torch.cuda.current_stream()

    def step():
        if global_mode:
            from grid_ndt_amd import dist as gdist
            gdist.build_global_map(m, "slope", pts, rank * n, stream)
        else:
            m.create2DMap("slope", pts, stream)

    for _ in range(a.warmup):
        step()
    m.sync()
    # Timed region: HIP events only around the dominant kernel of the strategy in use (two per build, on the launch
    # stream); the full per-phase breakdown comes from a few extra, untimed builds afterwards (events between all
    # kernels cost ~5 % of the step, which would be charged to `value`).
    m.set_profiling(2)
    phase_sum = {}
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()              # builds are launched back to back: nothing waits for the host between steps
    nodes, cols, slopes = m.sync()   # the last build's overflow flags are checked here, inside the timed region
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    # HIP events were recorded on the launch stream around every phase of every step (one event set per build,
    # a ring of 32); they are read once, after the timed region: mean duration per phase over the timed builds
    live = {k: v for k, v in m.phase_times_ms().items() if v >= 0}
    m.set_profiling(1)
    for _ in range(3):
        step()
    m.sync()
    for k, v in m.phase_times_ms().items():
        if v >= 0:
            phase_sum[k] = v * a.steps
    m.set_profiling(0)
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    nodes, cols, slopes = m.sync()

    if rank == 0:
        ms_step = dt / a.steps * 1e3
        default_workload = (n == 10_000_000 and a.grid_len == 0.5 and a.z_len == 0.5 and world == 1)
        total_points = n * world
        value = total_points / (dt / a.steps) / 1e6
        phases = {k: round(v / a.steps, 4) for k, v in phase_sum.items()}
        strat = m.STRATEGY_NAMES.get(m.last_strategy(), "?")
        # Dominant kernel = the longest phase.  Its algorithmic bytes (DESIGN.md "Roofline accounting"):
        # every kernel that streams the cloud moves 12 B/point; k_bucket_build also emits the nodes
        # (12 B/point + 76 B/node); node-proportional kernels move 76 B/node.
        kernel_of = {"accumulate": "k_accumulate", "hist": "k_part_hist", "scatter": "k_part_scatter",
                     "level1": "k_part2_level1", "level2": "k_part2_level2",
                     "bucket_build": "k_bucket_build", "columns": "k_tab_columns", "rows": "k_tab_rows",
                     "emit": "k_emit_rows"}
        cand = {k: v for k, v in phases.items() if k in kernel_of}
        dom = max(cand, key=cand.get) if cand else None
        # the dominant kernel's duration is the one measured live in the timed region when it is the phase the
        # two live events bracket (it is, unless the untimed breakdown says another phase is longer)
        acc_ms = live.get(dom, cand.get(dom, float("nan"))) if dom else float("nan")
        timed_live = dom in live
        if dom in ("accumulate", "hist", "scatter", "level1", "level2"):
            alg_bytes = BYTES_PER_POINT * n
        elif dom == "bucket_build":
            alg_bytes = BYTES_PER_POINT * n + BYTES_PER_NODE * nodes
        else:
            alg_bytes = BYTES_PER_NODE * nodes
        achieved = alg_bytes / (acc_ms * 1e-3) / 1e9 if acc_ms == acc_ms and acc_ms > 0 else None
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the
        # figure is the one the committed rocprofv3 --pmc passes of THIS workload measured (profiles/),
        # corrected as MI355X_MICROARCH.md prescribes; null for any other workload.
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_d_two_level_pmc.json")
        if default_workload and os.path.exists(pmc_path):
            for name, rec in json.load(open(pmc_path))["kernels"].items():
                if kernel_of.get(dom) and kernel_of[dom] in name:
                    traffic = rec["hbm_bytes_corrected"]
        path_bytes = BYTES_PER_POINT * n + BYTES_PER_NODE * nodes
        out = {
            "metric": "NDT grid-build throughput (bin + mean/cov + eigen + labels + ordering)",
            "value": round(value, 3), "unit": "Mpoints/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "S2: 10M uniform-random points in [-100,100)^2 x [-1,1), 0.5 m cubic voxels, "
                                   "demand=slope (BASELINE.json configs[1])" if n == 10_000_000 and a.grid_len == 0.5 and a.z_len == 0.5
                       else f"uniform box, {n} points/rank, grid {a.grid_len}/{a.z_len}",
                       "points_per_gpu": n, "nodes": int(nodes), "columns": int(cols), "slopes": int(slopes),
                       "multi_gpu_mode": a.mode if use_dist else "single", "strategy": strat},
            "roofline": {"bound": "hbm", "kernel": kernel_of.get(dom),
                         "achieved": round(achieved, 2) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5) if achieved else None, "traffic": traffic,
                         "traffic_source": "profiles/r01_d_two_level_pmc.json (rocprofv3 --pmc, separate passes)" if traffic else None,
                         "kernel_ms_measured_in_timed_region": timed_live, "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(acc_ms, 4) if acc_ms == acc_ms else None},
            "path_roofline": {"bytes": path_bytes, "achieved_GBps": round(path_bytes / (ms_step * 1e-3) / 1e9, 2),
                              "frac": round(path_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "phase_ms": phases,
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cloud, P, min(a.cpu_sample, n))
        elif world == 1:
            out["cpu_baseline"] = None
        if a.check:
            from tests import parity
            small = scenes.uniform_box(300_001)
            ref = parity.ref_from_cloud(small, P)
            _, o = parity.gpu_from_cloud(small, P, device=local)
            out["check"] = parity.compare(o, ref)["ok"]
        if a.stamps and m.last_strategy() in (2, 3):
            os.environ["GNDT_STAMPS"] = "1"
            m.create2DMap("slope", pts, stream)
            cyc, nb = m.debug_bucket_phases()
            tot = sum(list(cyc.values())[:6])
            print("k_bucket_build phase stamps (mean shader cycles per bucket, %d buckets): " % nb +
                  ", ".join(f"{k}={v:.0f} ({100 * v / tot:.0f}%)" for k, v in cyc.items()), file=sys.stderr)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
