#!/usr/bin/env python3
"""bench.py — NDT grid-build throughput (bin + mean/cov + eigen + labels + ordering) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload S1|S2|S2z|S3|S5] [--mode owner|global|replicas]

A "step" is one full build of the map from a device-resident cloud (`create2DMap`, receiver.cpp:150-160).

N = 1 (default): BASELINE.json configs[1] — S2, 10 M uniform-random points, 0.5 m cubic voxels.
N > 1 (default): BASELINE.json configs[2] — S3, ONE global map of a 100 M-point LiDAR-ordered terrain cloud, 0.2 m cubic
         voxels, sharded over the ranks as contiguous index ranges (strong scaling: the total is fixed).  `--mode owner`
         (default): every point goes to the rank that owns its column (RCCL all-to-all of 16-B records), each rank builds
         its columns, 8 B per column give every row its place in the global map — the map stays sharded by owner.
         `--mode global`: per-node statistics all-reduced, the whole map on every rank.  `--mode replicas` builds one
         independent map per rank instead (frame-level batches: no collective, weak scaling).

With N > 1 and no RANK in the environment this script starts the N ranks itself (`python -m torch.distributed.run`
as a child process, before anything here touches the GPU) and relays their output; under torchrun it is a rank.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
BYTES_PER_POINT = 12           # one read of packed fp32 xyz          
BYTES_PER_NODE = 76            # one write of the node's result row   

WORKLOADS = {
    "S1": dict(desc="S1: single 200 k-point campus frame (freiburg2_16 stand-in), grid 0.5 / z 0.1 m (BASELINE.json configs[0])",
               points=200_000, grid_len=0.5, z_len=0.1, hint=0),
    "S2": dict(desc="S2: 10M uniform-random points in [-100,100)^2 x [-1,1), 0.5 m cubic voxels, demand=slope (BASELINE.json configs[1])",
               points=10_000_000, grid_len=0.5, z_len=0.5, hint=1 << 20),
    "S2z": dict(desc="S2 variant: 10M uniform-random points, grid 0.5 / z 0.1 m (the reference's launch default)",
                points=10_000_000, grid_len=0.5, z_len=0.1, hint=3_400_000),
    "S3": dict(desc="S3: LiDAR-ordered outdoor terrain, 0.2 m cubic voxels, demand=slope (BASELINE.json configs[2])",
               points=100_000_000, grid_len=0.2, z_len=0.2, hint=0),
    "S4": dict(desc="S4: streaming 10 Hz LiDAR frames of 131072 points over the S3 terrain, incremental update per frame, 0.2 m cubic "
                    "voxels (BASELINE.json configs[3])", points=131_072, grid_len=0.2, z_len=0.2, hint=4_000_000),
    "S5": dict(desc="S5: two-storey site (site125 stand-in), 15 % of the points at (0,0,0), 0.1 m cubic voxels (BASELINE.json configs[4])",
               points=20_000_000, grid_len=0.1, z_len=0.1, hint=0),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None, help="default: S2 at N = 1, S3 at N > 1")
    ap.add_argument("--points", type=int, default=0, help="binned points of the workload, all ranks together (0 = its BASELINE size)")
    ap.add_argument("--strategy", type=int, default=0)
    ap.add_argument("--mode", choices=["replicas", "global", "owner"], default=None,
                    help="N>1: ONE map of a sharded cloud — 'owner' (default: points travel to the owner of their column, map sharded "
                         "by owner) or 'global' (statistics all-reduced, whole map on every rank) — or independent per-rank maps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-build latency / no-hint / H2D-D2H measurements")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (the other BASELINE.json configurations, untimed by `value`) "
                                                              "and the `host_path` block (the seam's host side, stage by stage)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="points of the CPU baseline's sample (0 = the whole workload, capped at 10 M)")
    ap.add_argument("--check", action="store_true", help="also check a 300 k-point build against the oracle")
    ap.add_argument("--stamps", action="store_true",
                    help="diagnostic: after the timed run, one extra build with in-kernel phase stamps (stderr)")
    ap.add_argument("--nodes-hint", type=int, default=-1, help="max_nodes_hint (-1 = the workload's default)")
    ap.add_argument("--graph", action="store_true", help="S4: replay one captured update per frame (hipGraph) instead of eager launches")
    ap.add_argument("--no-anchor", action="store_true", help="N>1: skip the single-GPU build of the same cloud on rank 0 (single_gpu_anchor)")
    ap.add_argument("--no-modes", action="store_true", help="N>1, --mode owner: skip the three extra steps of --mode global (modes block)")
    ap.add_argument("--watchdog", type=int, default=900, help="seconds a stage (warm-up + timed loop, gather, anchor, modes) may take before "
                                                             "the rank exits non-zero: a hung collective must not hang the job")
    ap.add_argument("--force-multi-extras", action="store_true",
                    help="testing on a one-GPU box (under torchrun with ONE rank): run the anchor / gather / modes stages of the N > 1 line all the same")
    ap.add_argument("--launch-check", action="store_true",
                    help="CPU-only check of the self-launch path: the ranks rendezvous over gloo, shard a small cloud "
                         "exactly as the timed run would, and rank 0 prints what every rank got")
    return ap.parse_args()


# what the N > 1 line carries beside the N = 1 keys (tests/test_bench_launcher.py)
MULTI_GPU_KEYS = ("single_gpu_anchor", "speedup_vs_single_gpu", "gather_ms", "modes", "exchange", "comm_selftest")


def watchdog(seconds):
    """Arms (seconds > 0) or disarms the stage watchdog: a C thread of the interpreter that dumps the tracebacks and _exit(1)s
    the rank — also while the main thread sits in a collective inside libgndt or RCCL.  torchrun then ends the other ranks."""
    import faulthandler
    faulthandler.cancel_dump_traceback_later()
    if seconds > 0:
        faulthandler.dump_traceback_later(seconds, exit=True)


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start the ranks as a child job.  The parent never touches the GPU — it
    does not even count the devices (without amdsmi that goes through hipGetDeviceCount and initialises the runtime): a rank
    that finds no device of its own says so and exits non-zero; the parent only relays the child job's exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def make_cloud(name, nb, rank, world, global_mode):
    """`nb` = binned points of the workload (the cloud has nb + 1: point 0 is the origin, receiver.cpp:145).
    -> (origin xyz, this rank's points [n,3] float32 WITHOUT the origin point, index of its first point in the
    accumulated stream, total binned points of the job)."""
    from grid_ndt_amd import scenes
    if name == "S3":
        ppf = scenes.FRAME_POINTS
        origin = scenes.terrain_frames(1, 0)[0].copy()
        if global_mode:                                       # contiguous index ranges of ONE cloud: global indices 1 .. nb
            lo, hi = 1 + rank * nb // world, 1 + (rank + 1) * nb // world
        else:                                                 # replicas: every rank its own stretch of the pose stream
            lo, hi = 1 + rank * (nb + 1), (rank + 1) * (nb + 1)
            if rank:
                origin = scenes.terrain_frames(1, (lo - 1) // ppf)[(lo - 1) % ppf].copy()
        f0, f1 = lo // ppf, (hi + ppf - 1) // ppf
        pts = scenes.terrain_frames(f1 - f0, f0)[lo - f0 * ppf: hi - f0 * ppf]
        return origin, np.ascontiguousarray(pts), (lo - 1) if global_mode else 0, (nb if global_mode else nb * world)
    seed_shift = 0 if global_mode else rank
    if name in ("S2", "S2z"):
        cloud = scenes.uniform_box(nb + 1, seed=0x5EED0002 + seed_shift)
    elif name == "S1":
        cloud = scenes.campus_frame(nb + 1, seed=0x5EED0001 + seed_shift)
    else:
        cloud = scenes.site_two_storey(nb + 1, seed=0x5EED0005 + seed_shift)
    origin = cloud[0].copy()
    if global_mode:
        lo, hi = 1 + rank * nb // world, 1 + (rank + 1) * nb // world
        return origin, np.ascontiguousarray(cloud[lo:hi]), lo - 1, nb
    return origin, np.ascontiguousarray(cloud[1:]), 0, nb * world


def shard_and_anchor(name, nb, rank, world, one_cloud, want_anchor):
    """make_cloud for this rank; with `want_anchor` rank 0 generates the WHOLE cloud once (the single-GPU anchor builds it) and
    takes its shard out of it.  -> (origin, shard, first index, job points, whole cloud on rank 0 or None)."""
    if want_anchor and rank == 0:
        origin, full, _, job = make_cloud(name, nb, 0, 1, True)
        hi = nb // world
        return origin, np.ascontiguousarray(full[:hi]), 0, job, full
    origin, pts, base, job = make_cloud(name, nb, rank, world, one_cloud)
    return origin, pts, base, job, None


def cpu_baseline(origin, pts, P, sample):
    """Oracle (faithful restatement of the reference path) timed on this box's host cores on the first `sample`
    points of the same workload: reported beside `value`, never the thing measured as `value`.
    OpenMP port (oracle mode 2): threads pinned (OMP_PROC_BIND=close, OMP_PLACES=cores, set before the library loads),
    arrays first touched by the threads that fill them, best of 3 at every thread count of the series."""
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    from oracle import oracle
    sub = np.ascontiguousarray(np.concatenate([origin[None, :], pts[:sample]], 0))
    threads = oracle.max_threads()
    series = sorted({t for t in (1, 16, 64, threads) if t <= threads})
    runs = {}
    t_wall = time.perf_counter()
    for th in series:
        best = None
        for _ in range(3 if th > 1 else 1):
            r = oracle.build_grid(sub, P["grid_len"], P["z_len"], P["slope_interval"], "slope", mode=oracle.MODE_INT_OPENMP,
                                  threads=th, export=False)
            if best is None or r["t_division"] + r["t_calculate"] < best[0] + best[1]:
                best = (r["t_division"], r["t_calculate"])
        runs[th] = best
    t_wall = time.perf_counter() - t_wall
    n_ser = min(sample, 200_000)
    r0 = oracle.build_grid(np.ascontiguousarray(sub[:n_ser + 1]), P["grid_len"], P["z_len"], P["slope_interval"], "slope",
                           mode=oracle.MODE_AS_SHIPPED, export=False)
    dt0 = r0["t_division"] + r0["t_calculate"]
    top = max(series, key=lambda th: sample / sum(runs[th]))          # the fastest thread count IS the baseline
    d, c = runs[top]
    d1, c1 = runs[1]
    whole = sample >= pts.shape[0]
    return {"value": round(sample / (d + c) / 1e6, 4), "unit": "Mpoints/s", "cores": top, "kind": "port",
            "sample": (f"all {sample} points of the workload" if whole else f"first {sample} points of the workload") +
                      f", oracle mode 2 (OpenMP, integer keys), best of 3, threads pinned; division {d:.3f}s + calculate {c:.3f}s "
                      f"(the whole series took {t_wall:.1f}s)",
            "host_threads_available": threads,
            "thread_series": {str(th): {"Mpoints/s": round(sample / sum(runs[th]) / 1e6, 3), "division_s": round(runs[th][0], 4),
                                        "calculate_s": round(runs[th][1], 4)} for th in series},
            "same_code_one_core": {"value": round(sample / (d1 + c1) / 1e6, 4), "unit": "Mpoints/s", "cores": 1},
            "as_shipped_serial": {"value": round(n_ser / dt0 / 1e6, 4), "unit": "Mpoints/s", "cores": 1,
                                  "sample": f"first {n_ser} points, oracle mode 0 (strings + multimap, as the reference runs)"}}


def full_size_parity(m, torch, origin, host_pts, P, pts, stream, dense=False):
    """EVERY node of the workload's map against the oracle (OpenMP port, all host threads, exported with its fp64 truth): keys,
    counts, first-seen order and labels exact, mean / covariance / lambda_min / normal under the gates of tests/parity.py.
    The oracle is the checker here, untimed and outside `value`; a failure makes the bench exit non-zero."""
    from oracle import oracle
    from tests import parity
    t0 = time.perf_counter()
    cloud = np.ascontiguousarray(np.concatenate([origin[None, :], host_pts], 0))
    ref = oracle.build_grid(cloud, P["grid_len"], P["z_len"], P["slope_interval"], P.get("demand", "slope"), mode=oracle.MODE_INT_OPENMP,
                            threads=oracle.max_threads(), export=True)
    del cloud
    t_ref = time.perf_counter() - t0
    m.create2DMap(P.get("demand", "slope"), pts, stream)
    got = m.export()
    rep = parity.compare(got, ref, P.get("demand", "slope"), dense=dense, interval=P["slope_interval"])
    keep = ("ok", "num_nodes", "label_mismatch_has_stats", "label_mismatch_slope", "label_mismatch_down", "labels_within_margin",
            "labels_on_the_margin", "mean_err", "cov_err", "cov_err_truth", "rough_err", "normal_err", "normals_checked",
            "cov_nodes_over_1e-5_vs_fp32", "cov_widening_max", "cov_nodes_below_input_resolution", "fail")
    out = {k: rep[k] for k in keep if k in rep}
    out["labels_within_margin"] = int(rep.get("labels_within_margin", 0))      # (0 unless the dense gate is on: labels are exact)
    out["dense_gate"] = bool(dense)
    out["oracle_s"] = round(t_ref, 2)
    out["what"] = ("parity.compare(gndt export, oracle mode 2 export) over ALL nodes of this workload: keys / counts / first-seen order / "
                   "labels exact; covariance <= 1e-5 max|C| vs the fp32-sequential restatement and <= 2e-6 vs fp64 truth")
    return out


def _timed_builds(g, torch, cloud, P, steps, hint=0, strategy=0, demand="slope"):
    """ms per back-to-back build of `cloud` (device-resident) on a warmed-up handle + what a FRESH handle's first build costs."""
    pts = torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = g.TwoDmap(P["grid_len"], P["z_len"], max_nodes_hint=hint, strategy=strategy)
    m.setInterval(P["slope_interval"])
    m.setCloudFirst(cloud[0])
    m.create2DMap(demand, pts)
    m.sync()
    first_ms = (time.perf_counter() - t0) * 1e3
    first_retries = m.retry_count()
    for _ in range(3):
        m.create2DMap(demand, pts)
        m.sync()
    torch.cuda.synchronize()
    r0 = m.retry_count()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.create2DMap(demand, pts)
    nodes, cols, slopes = m.sync()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    n = cloud.shape[0] - 1
    # ... and a second fresh handle, now that the process has launched every kernel this cloud needs once (first_build_ms above
    # includes their first launches; this one is handle creation + allocations + the build: tools/first_build_breakdown.py).
    # The first handle is dropped before it: a second stream ALIVE in the process costs a hardware queue once (5.6 ms), which is the
    # process's, not a build's.
    strategy_name = m.STRATEGY_NAMES[m.last_strategy()]
    retries = int(m.retry_count() - r0)
    del m
    t0 = time.perf_counter()
    m2 = g.TwoDmap(P["grid_len"], P["z_len"], max_nodes_hint=hint, strategy=strategy)
    m2.setInterval(P["slope_interval"])
    m2.setCloudFirst(cloud[0])
    m2.create2DMap(demand, pts)
    m2.sync()
    second_ms = (time.perf_counter() - t0) * 1e3
    second_retries = m2.retry_count()
    del m2
    # ... and a third one that was warmed up and reserved BEFORE the clock starts (gndt_warmup: what a node that builds one map per
    # process — the reference's receiver, receiver.cpp:137-160 — does in main() before the first cloud arrives)
    # Three such handles, one after the other (each fresh, warmed, ONE build timed): the number is one host-side call + one wait of
    # ~0.1 ms, and a box whose host cores are busy doubles a single shot (0.126 | 0.249 on two boxes for the same sources) — the
    # median is reported, the three samples beside it.
    warmed, warmed_retries = [], 0
    for _ in range(3):
        m3 = g.TwoDmap(P["grid_len"], P["z_len"], max_nodes_hint=hint, strategy=strategy, max_points_hint=n)
        m3.setInterval(P["slope_interval"])
        m3.setCloudFirst(cloud[0])
        m3.warmup(n, demand)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m3.create2DMap(demand, pts)
        m3.sync()
        warmed.append((time.perf_counter() - t0) * 1e3)
        warmed_retries = max(warmed_retries, m3.retry_count())
        del m3
    warmed_ms = sorted(warmed)[1]
    return {"points": int(n), "first_build_warmed_ms": round(warmed_ms, 3), "first_build_warmed_samples_ms": [round(w, 3) for w in warmed],
            "first_build_warmed_re_runs": int(warmed_retries), "nodes": int(nodes), "ms_per_build": round(dt * 1e3, 4), "Mpoints_per_s": round(n / dt / 1e6, 1),
            "first_build_second_handle_ms": round(second_ms, 3), "first_build_second_handle_re_runs": int(second_retries),
            "strategy": strategy_name, "path_frac": round((12 * n + 76 * nodes) / dt / (HBM_PEAK_GBS * 1e9), 5),
            "retries": retries, "steps": steps,
            "first_build_ms": round(first_ms, 3), "first_build_re_runs": int(first_retries)}


def _stream_latency(g, torch, frames, ppf, nframes, graph_mode, deferred=False):
    """S4: one gndt_update per 131 072-point frame; per-frame latency (launch -> device idle) and the back-to-back rate."""
    dev_frames = [torch.from_numpy(frames[f * ppf:(f + 1) * ppf]).cuda() for f in range(nframes)]
    m = g.TwoDmap(0.2, 0.2, strategy=1, max_nodes_hint=4_000_000, max_points_hint=nframes * ppf)
    m.setInterval(0.08)
    m.setCloudFirst(frames[0])
    buf = torch.empty(ppf, 3, dtype=torch.float32, device="cuda")
    if deferred:
        m.set_deferred_emit(True)
    buf.copy_(dev_frames[0])
    m.change2DMap("slope", buf)
    m.sync()
    graph = None
    if graph_mode:
        graph = torch.cuda.CUDAGraph()
        with g.graph_capture(graph):
            m.change2DMap("slope", buf)
    lat = []
    half = nframes // 2
    torch.cuda.synchronize()
    for f in range(1, half):                       # first half: one awaited frame at a time
        buf.copy_(dev_frames[f])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        graph.replay() if graph else m.change2DMap("slope", buf)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(half, nframes):                 # second half: back to back
        buf.copy_(dev_frames[f])
        graph.replay() if graph else m.change2DMap("slope", buf)
    torch.cuda.synchronize()                      # (a replayed graph runs on torch's stream, not on the one it was captured on, which is all m.sync() waits for)
    b2b = (time.perf_counter() - t0) / (nframes - half) * 1e3
    t0 = time.perf_counter()
    nodes, _, _ = m.sync()
    read_ms = (time.perf_counter() - t0) * 1e3
    lat = np.sort(np.array(lat[2:]))
    out = {"p50_ms": round(float(np.percentile(lat, 50)), 4), "p99_ms": round(float(np.percentile(lat, 99)), 4),
           "back_to_back_ms_per_frame": round(b2b, 4), "nodes_at_the_end": int(nodes)}
    if deferred:
        out["read_ms"] = round(read_ms, 4)
        out["what"] = ("gndt_set_deferred_emit: a frame relabels the touched columns and stops; read_ms = the gndt_sync that orders and emits "
                       "the dense rows for all the frames since the last read")
    return out


def measure_configs(g, torch, s2_cloud_host):
    """The BASELINE.json configurations other than the bench line, each on ONE GPU, untimed by `value`:
    configs[0] S1 (campus stand-in, the reference's own bridge_ground scene, an organised depth frame), configs[1]'s launch-default
    z variant, configs[2] / [4] at a size whose generation fits the default run (said in the key), configs[3] S4 eager and replayed."""
    from grid_ndt_amd import scenes
    out = {}
    out["S1_campus_200k"] = _timed_builds(g, torch, scenes.campus_frame(200_001), scenes.CAMPUS_PARAMS, steps=30)
    out["S1_bridge_ground_360k"] = _timed_builds(g, torch, scenes.bridge_ground(), scenes.BRIDGE_PARAMS, steps=30)
    out["S1_depth_frame_215k"] = _timed_builds(g, torch, scenes.depth_frame(), scenes.DEPTH_PARAMS, steps=30)
    if s2_cloud_host is not None:
        out["S2_first_build"] = {k: v for k, v in _timed_builds(g, torch, s2_cloud_host, dict(grid_len=0.5, z_len=0.5, slope_interval=0.08), steps=3).items()
                                 if k in ("first_build_ms", "first_build_re_runs", "first_build_second_handle_ms", "first_build_warmed_ms", "first_build_warmed_samples_ms", "first_build_warmed_re_runs", "points", "nodes")}
        out["S2z_10M_z01"] = _timed_builds(g, torch, s2_cloud_host, dict(grid_len=0.5, z_len=0.1, slope_interval=0.08), steps=10, hint=3_400_000)
    r = _timed_builds(g, torch, scenes.terrain_cloud(8_000_001), dict(grid_len=0.2, z_len=0.2, slope_interval=0.08), steps=10)
    r["note"] = "8 M of configs[2]'s 100 M points (scene generation time); the full size on one GPU: bench.py --workload S3 --points 100000000"
    out["S3_terrain_8M"] = r
    r = _timed_builds(g, torch, scenes.site_two_storey(5_000_001), dict(grid_len=0.1, z_len=0.1, slope_interval=0.08), steps=10)
    r["note"] = "5 M of configs[4]'s 20 M points, 15 % of them at (0,0,0); the full size: bench.py --workload S5"
    out["S5_site_5M"] = r
    nframes, ppf = 48, scenes.FRAME_POINTS
    frames = scenes.terrain_frames(nframes, 0)
    out["S4_stream_131k_frames"] = {"frames": nframes, "budget_ms": 100.0, "eager": _stream_latency(g, torch, frames, ppf, nframes, False),
                                    "hip_graph_replay": _stream_latency(g, torch, frames, ppf, nframes, True),
                                    "deferred_emit": _stream_latency(g, torch, frames, ppf, nframes, False, deferred=True),
                                    "deferred_emit_replayed": _stream_latency(g, torch, frames, ppf, nframes, True, deferred=True)}
    return out


def build_host_path_tool():
    """tools/host_path (C++: gndt_compat.hpp over the C ABI), built in-tree like the examples."""
    from grid_ndt_amd import _lib
    exe = os.path.join(ROOT, "tools", "host_path")
    src = exe + ".cpp"
    deps = [src, os.path.join(ROOT, "include", "gndt_compat.hpp"), os.path.join(ROOT, "include", "gndt.h"), _lib.LIB_PATH]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        csrc = os.path.dirname(_lib.LIB_PATH)
        hip = _lib._hip_runtime_dir()
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                               "-o", exe, src, "-L", csrc, "-l:libgndt.so", "-L", hip, "-l:libamdhip64.so", f"-Wl,-rpath,{csrc}", f"-Wl,-rpath,{hip}"])
    return exe


def measure_host_path():
    """The HOST side of the seam (src/receiver.cpp:140-175) for configs[0]-sized clouds, stage by stage, through the C++ mirror of
    the reference's API (include/gndt_compat.hpp): host cloud -> gndt_build + gndt_sync -> gndt_export_host -> containers ->
    computeCost -> findRoute, eager (containers rebuilt) and lazy (consumers served from the exported rows), beside what the
    oracle's as-shipped restatement of the reference's two loops takes on this box.  A child process: plain C++, no torch."""
    import tempfile
    from grid_ndt_amd import scenes
    from oracle import oracle
    exe = build_host_path_tool()
    out = {}
    site = scenes.drivable_site(400_000)
    start = (-20.0, 10.0, float(0.35 * np.sin(-20.0 / 7.0) + 0.25 * np.cos(10.0 / 5.0)))
    cases = (("S1_campus_200k", scenes.campus_frame(200_001), scenes.CAMPUS_PARAMS, (10.0, -10.0, 0.7), (30.0, -30.0, 0.4)),
             ("S1_bridge_ground_360k", scenes.bridge_ground(), scenes.BRIDGE_PARAMS, (9.5, 3.0, 1.0), (9.5, 3.0, 3.0)),      # parameters.txt:53-59
             ("drivable_site_400k", site, scenes.COST_PARAMS, scenes.DRIVABLE_GOAL, start))
    with tempfile.TemporaryDirectory() as td:
        for name, cloud, P, goal, st in cases:
            fn = os.path.join(td, name + ".f32")
            np.ascontiguousarray(cloud, np.float32).tofile(fn)
            cmd = [exe, fn, str(cloud.shape[0]), repr(P["grid_len"]), repr(P["z_len"]), repr(P["slope_interval"]), P.get("demand", "slope")] + \
                  [repr(float(v)) for v in goal] + [repr(float(v)) for v in st] + ["0.25", "7"]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            try:
                rec = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception:
                rec = {"error": (r.stdout + r.stderr)[-300:]}
            t0 = time.perf_counter()
            ro = oracle.build_grid(cloud, P["grid_len"], P["z_len"], P["slope_interval"], P.get("demand", "slope"), mode=oracle.MODE_AS_SHIPPED, export=False)
            rec["oracle_as_shipped_ms"] = {"division": round(ro["t_division"] * 1e3, 1), "calculate": round(ro["t_calculate"] * 1e3, 1),
                                           "what": "the reference's two loops (receiver.cpp:150-160) as restated by the oracle, 1 core, this box"}
            out[name] = rec
    out["what"] = ("tools/host_path.cpp: medians of 7 (ms); build_sync = gndt_build from a pageable host buffer + gndt_sync; export = gndt_export_host "
                   "(pinned mirror); containers = materialise_into (eager) or the column index (lazy); never part of `value`")
    return out



def measure_cost_flood(g, torch):
    """The immediate consumer of the grid: gndt_compute_cost — TwoDmap::computeCost, include/map2D.h:1285-1397,
    with CollisionCheck :351-474 — on three finished grids, each checked against the oracle's flag-based restatement of the reference's
    FIFO flood run on the same exported grid (h and state bit for bit).  Untimed by `value`."""
    from grid_ndt_amd import scenes
    from oracle import oracle
    out = {}
    cases = (("drivable_site_400k", scenes.drivable_site(400_000), scenes.COST_PARAMS, scenes.DRIVABLE_GOAL, None),
             ("bridge_ground_360k_own_parameters", scenes.bridge_ground(), scenes.BRIDGE_PARAMS, (9.5, 3.0, 1.0), {"radius": 0.25}),   # parameters.txt:53-59
             ("terrain_2M", scenes.terrain_cloud(2_000_000), dict(grid_len=0.5, z_len=0.25, slope_interval=0.08), None, None))
    for name, cloud, P, goal, robot in cases:
        demand = P.get("demand", "slope")
        m = g.TwoDmap(P["grid_len"], P["z_len"])
        m.setInterval(P["slope_interval"])
        m.setCloudFirst(cloud[0])
        m.create2DMap(demand, torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda())
        cells = m.export()
        if goal is None:
            rows = np.nonzero((cells["flags"] & 2) != 0)[0]
            goal = cells["mean"][rows[len(rows) // 3]]
        st = m.computeCost(goal, robot=robot)
        # the first flood on a new map works out what depends on map and robot alone; floods for further goals keep it
        pts = torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda()
        first = []
        for _ in range(3):
            m.create2DMap(demand, pts)
            m.sync()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = m.computeCost(goal, robot=robot)
            first.append((time.perf_counter() - t0) * 1e3)
        del pts
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = m.computeCost(goal, robot=robot)
            ts.append((time.perf_counter() - t0) * 1e3)
        got = m.cost_export()
        t0 = time.perf_counter()
        ref = oracle.compute_cost(cells, cloud[0], P["grid_len"], P["z_len"], P["slope_interval"], goal, demand=demand, robot=robot, mode=oracle.COST_FLAGS)
        cpu_ms = (time.perf_counter() - t0) * 1e3
        ms = float(np.median(ts))
        out[name] = {"points": int(cloud.shape[0] - 1), "nodes": int(cells["num_nodes"]), "slopes": int(cells["num_slopes"]),
                     "ring_depth": int(st["ring"]), "layers": int(st["levels"]), "traversable": int(st["traversable"]), "closed": int(st["closed"]),
                     "gpu_ms": round(ms, 3), "gpu_first_flood_on_a_map_ms": round(float(np.median(first)), 3),
                     "us_per_layer": round(ms * 1e3 / max(int(st["levels"]), 1), 2),
                     # this block's cpu_baseline leg: the oracle's flood, timed beside the GPU's and used as its checker
                     "cpu_baseline": {"value": round(cpu_ms, 1), "unit": "ms", "cores": 1, "kind": "port",
                                      "sample": "the whole flood on the same exported grid, oracle/cost_cpu.cpp (flag-based restatement of map2D.h:1285-1397)"},
                     "h_bit_exact": bool((got["h"] == ref["h"]).all()), "state_exact": bool((got["state"] == ref["state"]).all())}
        del m
    out["what"] = ("gndt_compute_cost (host call to host return): gpu_first_flood_on_a_map_ms = the first flood after a build (median of 3), "
                   "gpu_ms = a flood for a further goal on the same map (median of 5; column index, neighbour records and collision verdicts kept); cpu_baseline = the "
                   "oracle's flag-based restatement of the reference's FIFO flood, one core of this box, which also checks the GPU's h and state; "
                   "never part of `value`")
    return out


def committed_traffic(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes — only when that profile
    was taken from THIS source tree (hash of the kernel sources stored in the profile); otherwise null."""
    from grid_ndt_amd import _lib
    prof_dir = os.path.join(ROOT, "profiles")
    cur = _lib.source_hash()
    best = None
    for fn in sorted(os.listdir(prof_dir)) if os.path.isdir(prof_dir) else []:
        if not fn.endswith("_pmc.json"):
            continue
        try:
            doc = json.load(open(os.path.join(prof_dir, fn)))
        except Exception:
            continue
        if doc.get("source_hash") != cur:
            continue
        for name, rec in doc.get("kernels", {}).items():
            if kernel_substr and kernel_substr in name:
                best = (rec.get("hbm_bytes_corrected"), f"profiles/{fn} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, "
                                                         f"source_hash {cur[:12]})")
    return best if best else (None, None)


def launch_check(a, rank, world):
    """No GPU: proves that `bench.py --gpus N` reaches N ranks with the right arguments and that the shards of the
    default N > 1 workload tile the cloud (tests/test_bench_launcher.py)."""
    import torch
    import torch.distributed as dist
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    dist.init_process_group("gloo")
    wname = a.workload or ("S3" if world > 1 else "S2")
    mode = a.mode or ("owner" if world > 1 else "single")
    total = a.points or 300_000
    one_cloud = mode in ("global", "owner")
    origin, pts, base, job, anchor = shard_and_anchor(wname, total, rank, world, one_cloud, want_anchor=one_cloud and world > 1 and not a.no_anchor)
    mine = torch.tensor([base, pts.shape[0], job], dtype=torch.int64)
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    if rank == 0:
        out = {"launch_check": True, "n_gpus": world, "workload": wname, "mode": mode,
               "shards": [[int(v) for v in g] for g in got], "origin": [float(v) for v in origin]}
        if one_cloud and world > 1:       # what the timed run adds to its line, and the cloud rank 0 builds alone for the anchor
            out["multi_gpu_keys"] = [k for k in MULTI_GPU_KEYS if not (k == "single_gpu_anchor" and a.no_anchor)
                                     and not (k == "speedup_vs_single_gpu" and a.no_anchor) and not (k == "gather_ms" and mode != "owner")
                                     and not (k == "modes" and (mode != "owner" or a.no_modes))]
            out["anchor_points"] = None if anchor is None else int(anchor.shape[0])
            out["anchor_is_the_ranks_cloud"] = None if anchor is None else bool(
                np.array_equal(anchor[base:base + pts.shape[0]], pts) and anchor.shape[0] == job)
        print(json.dumps(out))
    dist.destroy_process_group()


def run_stream(a):
    """--workload S4: a step is ONE frame of 131072 points added to the map (gndt_update_device: accumulate into the HBM node
    table, relabel the touched columns, re-order and re-emit the rows behind the first column that changed size).  `value` counts the timed frames' points against the
    wall time of the back-to-back loop; the per-frame latency (launch -> results ready) is measured in a second, synchronised
    pass over the same frames."""
    import torch
    import grid_ndt_amd as g
    from grid_ndt_amd import scenes
    g.build_native()
    W = WORKLOADS["S4"]
    ppf = scenes.FRAME_POINTS
    nframes = a.warmup + a.steps + 1
    frames = scenes.terrain_frames(nframes, 0)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dev_frames = [torch.from_numpy(frames[f * ppf:(f + 1) * ppf]).to(dev) for f in range(nframes)]
    hint = W["hint"] if a.nodes_hint < 0 else a.nodes_hint

    def run(sync_each):
        m = g.TwoDmap(W["grid_len"], W["z_len"], strategy=a.strategy or 1, max_nodes_hint=hint, max_points_hint=nframes * ppf)
        m.setInterval(0.08)
        m.setCloudFirst(frames[0])
        buf = torch.empty(ppf, 3, dtype=torch.float32, device=dev)
        buf.copy_(dev_frames[0][:ppf])
        m.change2DMap("slope", buf)      # frame 0 (its point 0 is the origin AND a point of the stream, as tools/measure_configs.py did)
        m.sync()
        graph = None
        if a.graph:
            graph = torch.cuda.CUDAGraph()
            with g.graph_capture(graph):
                m.change2DMap("slope", buf)
        lat = []
        for f in range(1, a.warmup + 1):
            buf.copy_(dev_frames[f])
            graph.replay() if graph else m.change2DMap("slope", buf)
        m.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(a.warmup + 1, nframes):
            buf.copy_(dev_frames[f])
            if sync_each:
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            graph.replay() if graph else m.change2DMap("slope", buf)
            if sync_each:
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - t1) * 1e3)
        nodes, cols, slopes = m.sync()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, lat, (nodes, cols, slopes), m

    dt, _, (nodes, cols, slopes), m = run(False)
    strat = m.STRATEGY_NAMES.get(m.last_strategy(), "?")
    del m
    _, lat, _, _ = run(True)
    lat = np.sort(np.array(lat))
    ms_step = dt / a.steps * 1e3
    path_bytes = BYTES_PER_POINT * ppf + BYTES_PER_NODE * nodes      # (upper bound: a frame re-emits the rows behind the first column that changed size, most of the map)
    out = {"metric": "NDT grid-build throughput (bin + mean/cov + eigen + labels + ordering)", "value": round(ppf / (dt / a.steps) / 1e6, 3),
           "unit": "Mpoints/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "io_dtype": "f32", "data": "synthetic",
           "config": {"workload": W["desc"], "points_per_frame": ppf, "frames_timed": a.steps, "nodes_at_the_end": int(nodes), "columns": int(cols),
                      "slopes": int(slopes), "strategy": strat, "hip_graph_replay": bool(a.graph), "max_nodes_hint": int(hint)},
           "frame_latency_ms": {"p50": round(float(np.percentile(lat, 50)), 4), "p99": round(float(np.percentile(lat, 99)), 4),
                                "max": round(float(lat.max()), 4), "budget": 100.0,
                                "what": "one frame, launch -> device idle (host-timed, synchronised before and after)"},
           "roofline": None, "path_roofline": {"bytes": int(path_bytes), "frac": round(path_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
           "cpu_baseline": None}
    print(json.dumps(out))


def main():
    a = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # before anything initialises HIP / HSA in this process
    if a.workload == "S4":
        return run_stream(a)
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(a))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.launch_check:
        return launch_check(a, rank, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (libgndt has no CPU path)"
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if local >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} wants GPU {local} but only {torch.cuda.device_count()} visible", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    mode = a.mode or ("owner" if world > 1 else "single")
    use_dist = world > 1 or ("RANK" in os.environ and mode in ("global", "owner"))   # torchrun with one rank exercises the exchange too
    if use_dist:
        watchdog(a.watchdog)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    global_mode = use_dist and mode in ("global", "owner")      # ONE cloud sharded over the ranks
    owner_mode = use_dist and mode == "owner"

    import grid_ndt_amd as g
    g.build_native()
    wname = a.workload or ("S3" if world > 1 else "S2")
    W = WORKLOADS[wname]
    total = a.points or W["points"]
    hint = W["hint"] if a.nodes_hint < 0 else a.nodes_hint
    P = dict(grid_len=W["grid_len"], z_len=W["z_len"], slope_interval=0.08, demand="slope")
    t_gen = time.perf_counter()
    several = world > 1 or a.force_multi_extras
    want_anchor = global_mode and several and not a.no_anchor
    watchdog(0)
    origin, host_pts, first_base, job_points, anchor_host = shard_and_anchor(wname, total, rank, world, global_mode, want_anchor)
    t_gen = time.perf_counter() - t_gen
    watchdog(a.watchdog)
    n = host_pts.shape[0]
    dev = torch.device(f"cuda:{local}")
    torch.cuda.synchronize()
    t_h2d = time.perf_counter()
    pts = torch.from_numpy(host_pts).to(dev)
    torch.cuda.synchronize()
    t_h2d = time.perf_counter() - t_h2d

    def new_map(nodes_hint):
        mm = g.TwoDmap(P["grid_len"], P["z_len"], device=local, max_nodes_hint=nodes_hint, strategy=a.strategy)
        mm.setInterval(P["slope_interval"])
        mm.setCloudFirst(origin)
        return mm

    m = new_map(hint)
    stream = torch.cuda.current_stream()
    exch = {}
    comm = None
    one_cloud = global_mode                     # the ranks hold contiguous index ranges of ONE cloud (stays true if the mode steps down)
    comm_error = None
    if global_mode:
        from grid_ndt_amd import dist as gdist
        try:
            comm = gdist.Communicator(local)    # libgndt's own RCCL communicator: the exchange runs inside the library
        except Exception as e:                  # (reported and agreed on below: every rank steps down together)
            comm_error = f"{type(e).__name__}: {e}"

    def step(mm=None, timed=False):
        mm = mm or m
        if owner_mode:
            _, t = mm.build_owned(comm, "slope", pts, first_base, job_points, stream)
            for k in ("split_ms", "exchange_ms", "build_ms", "order_ms"):
                exch[k] = exch.get(k, 0.0) + t[k]
            exch.update({k: t[k] for k in ("owned_points", "local_nodes", "local_columns", "global_nodes", "global_columns", "global_slopes",
                                           "bytes_sent", "bytes_received")})
        elif global_mode:
            t = mm.build_global(comm, "slope", pts, first_base, job_points, stream, timed=timed)
            if t:
                for k in ("shard_ms", "exchange_ms", "finalize_ms"):
                    exch[k] = exch.get(k, 0.0) + t[k]
                exch.update(global_nodes=t["global_nodes"], local_nodes=t["local_nodes"], bytes_reduced_per_rank=t["bytes_reduced"])
        else:
            mm.create2DMap("slope", pts, stream)

    # ---- N > 1: one untimed build through the exchange before anything is measured.  If the library's RCCL path reports an
    # error (every rank gets one at the same collective: gndt.h, GNDT_ERR_PEER), all ranks agree on it over torch.distributed
    # and step down together: owner -> global -> the shards built with no exchange at all (said so in the line, never silent).
    # A rank that hangs instead is ended by the watchdog. ----
    fallback = []
    selftest = None
    if global_mode and comm is not None:
        # one verified round of every collective libgndt's exchange uses, before anything is built on it: the first multi-rank run
        # of a node is otherwise also the first test of the transport (reported per primitive; a wrong answer is an error below)
        from grid_ndt_amd._lib import GndtError
        try:
            selftest = m.comm_selftest(comm, stream)
        except GndtError as e:
            selftest = {"error": str(e)}
            comm_error = comm_error or f"communicator self-test: {e}"
    if global_mode and several:
        from grid_ndt_amd._lib import GndtError
        while global_mode:
            err = comm_error
            try:
                if mode in os.environ.get("GNDT_BENCH_FAIL_MODES", "").split(","):     # tests only: the step-down itself
                    raise RuntimeError(f"injected failure of --mode {mode} (GNDT_BENCH_FAIL_MODES)")
                if err is None:
                    step()
                    m.sync()
            except (GndtError, RuntimeError) as e:
                err = f"{type(e).__name__}: {e}"
            okt = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            if int(okt.item()):
                break
            fallback.append({"mode": mode, "error": err or "another rank reported an error"})
            if rank == 0:
                print(f"bench.py: --mode {mode} failed on its first build ({fallback[-1]['error']}): stepping down", file=sys.stderr)
            mode = "global" if (mode == "owner" and comm is not None) else "shards_without_exchange"
            owner_mode = False
            global_mode = mode == "global"
            if not global_mode:                 # each rank's shard as a map of its own: origin and indices stay those of the one cloud
                want_anchor = False
            exch.clear()
            m = new_map(hint)
    for _ in range(a.warmup):
        step()
        m.sync()                     # every warm-up build is resolved: what it teaches the handle (node count, table size a
                                     # re-run needed) is in place before the next one, so the timed builds run in steady state
    exch.clear()
    # Timed region: HIP events only around the dominant kernel of the strategy in use (two per build, on the launch
    # stream); the full per-phase breakdown comes from a few extra, untimed builds afterwards (events between all
    # kernels cost ~5 % of the step, which would be charged to `value`).
    m.set_profiling(2)
    retries_before = m.retry_count()
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
    # Builds are enqueued back to back and only the last one's overflow flags are looked at (inside the timed region).  The
    # warm-up builds have taught the handle this cloud's sizes; if the last timed build still had to be re-run, the ones
    # before it ran short as well and the timing is not that of complete builds.
    retries_timed = m.retry_count() - retries_before
    if use_dist:                         # a re-run on ANY rank invalidates the step time
        rt = torch.tensor([retries_timed], dtype=torch.int64, device=dev)
        dist.all_reduce(rt, op=dist.ReduceOp.SUM)
        retries_timed = int(rt.item())
    if retries_timed and rank == 0:
        print(f"bench.py: ERROR {retries_timed} build re-run(s) inside the timed region: the step time is not that of complete builds "
              "(the line is printed with \"valid\": false and the exit code is 3)", file=sys.stderr)
    if global_mode:                       # stage times from a few extra, untimed steps (their events make the call wait)
        exch.clear()
        for _ in range(3):
            step(timed=True)
        exch = {k: (v / 3 if k.endswith("_ms") else v) for k, v in exch.items()}
    exch_timed = dict(exch)
    live = {k: v for k, v in m.phase_times_ms().items() if v >= 0}
    m.set_profiling(1)
    for _ in range(3):
        step()
    m.sync()
    phases = {k: round(v, 4) for k, v in m.phase_times_ms().items() if v >= 0}
    m.set_profiling(0)
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    nodes, cols, slopes = m.sync()
    n_local = n
    multi = {}
    if selftest is not None:
        multi["comm_selftest"] = selftest
    if fallback:
        multi["fallback"] = fallback
    rank0_nodes = nodes
    if mode == "shards_without_exchange":      # one map per rank: the totals are sums (nodes of columns two shards share count twice)
        tot = torch.tensor([nodes, cols, slopes], dtype=torch.int64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        nodes, cols, slopes = (int(v) for v in tot.tolist())
    if owner_mode:                       # the handle holds this rank's columns; the map of the whole cloud has:
        n_local = int(exch_timed.get("owned_points", n))
        local_nodes = nodes
        nodes, cols, slopes = int(exch_timed["global_nodes"]), int(exch_timed["global_columns"]), int(exch_timed["global_slopes"])
        # ---- ONE map for the consumers (untimed by `value`): the rows of all ranks gathered on rank 0 and scattered by global row ----
        watchdog(a.watchdog)
        g_ms = []
        for _ in range(3):
            step()
            m.sync()
            torch.cuda.synchronize()
            dist.barrier()
            t1 = time.perf_counter()
            m.gather_owned(comm, 0, stream)
            torch.cuda.synchronize()
            g_ms.append((time.perf_counter() - t1) * 1e3)
        tg = torch.tensor([float(np.median(g_ms))], dtype=torch.float64, device=dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        multi["gather_ms"] = {"value": round(float(tg.item()), 4), "rows": int(nodes), "bytes_per_row": 84,
                              "what": "gndt_gather_owned_map_device to rank 0 (ncclSend / ncclRecv of the packed rows + scatter by global row), "
                                      "median of 3, max over ranks; after it rank 0's handle holds the whole map"}
        if rank == 0:
            an, ac, asl = m.sync()
            multi["gather_ms"]["assembled_map_matches_totals"] = bool((an, ac, asl) == (nodes, cols, slopes))
    if global_mode and several and owner_mode and not a.no_modes:
        # ---- BASELINE configs[2] names the "cell-stat all-reduce": the same cloud through --mode global, three steps ----
        watchdog(a.watchdog)
        from grid_ndt_amd._lib import GndtError
        try:
            mg = new_map(hint)
            mg.build_global(comm, "slope", pts, first_base, job_points, stream)
            mg.sync()
            torch.cuda.synchronize()
            dist.barrier()
            t1 = time.perf_counter()
            for _ in range(3):
                mg.build_global(comm, "slope", pts, first_base, job_points, stream)
            gn, _, _ = mg.sync()
            torch.cuda.synchronize()
            tgl = torch.tensor([(time.perf_counter() - t1) / 3 * 1e3], dtype=torch.float64, device=dev)
            dist.all_reduce(tgl, op=dist.ReduceOp.MAX)
            multi["modes"] = {"owner": {"ms_per_step": round(dt / a.steps * 1e3, 4)},
                              "global": {"ms_per_step": round(float(tgl.item()), 4), "nodes": int(gn), "steps": 3,
                                         "what": "gndt_build_global_device: key all-gather + packed sum / min all-reduce of the per-node statistics, "
                                                 "the whole map finalised on every rank"}}
            del mg
        except GndtError as e:           # (every rank gets an error at the same collective: gndt.h, GNDT_ERR_PEER)
            multi["modes"] = {"owner": {"ms_per_step": round(dt / a.steps * 1e3, 4)}, "global": {"error": str(e)}}
    if want_anchor:
        # ---- the same cloud built by rank 0 ALONE: the N = 1 point of the scaling series, measured in this very run ----
        watchdog(max(a.watchdog, 1800))
        dist.barrier()
        if rank == 0:
            full = torch.from_numpy(anchor_host).to(dev)
            ma = new_map(hint)
            for _ in range(2):
                ma.create2DMap("slope", full, stream)
                ma.sync()
            r0 = ma.retry_count()
            asteps = max(3, min(a.steps, 10))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(asteps):
                ma.create2DMap("slope", full, stream)
            an, ac, asl = ma.sync()
            torch.cuda.synchronize()
            a_ms = (time.perf_counter() - t1) / asteps * 1e3
            multi["single_gpu_anchor"] = {"ms_per_step": round(a_ms, 4), "Mpoints_per_s": round(job_points / a_ms / 1e3, 3), "steps": asteps,
                                          "points_total": int(full.shape[0]), "nodes": int(an), "columns": int(ac), "slopes": int(asl),
                                          "retries_in_timed_region": int(ma.retry_count() - r0),
                                          "same_map_as_the_ranks": bool((an, ac, asl) == (nodes, cols, slopes)),
                                          "what": "gndt_build_device of the WHOLE cloud on rank 0's GPU while the other ranks wait (not part of `value`)"}
            multi["speedup_vs_single_gpu"] = round(a_ms / (dt / a.steps * 1e3), 4)
            del full, ma
        dist.barrier()
    watchdog(0)

    # ---- extras (untimed, rank 0 of a single-GPU run): what a 10 Hz callback sees, the no-hint path, PCIe legs ----
    extras = {}
    if world == 1 and not a.no_extras and not global_mode:
        lat = []
        for _ in range(7):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step()
            m.sync()
            lat.append((time.perf_counter() - t1) * 1e3)
        extras["single_build_latency_ms"] = {"median": round(float(np.median(lat)), 4), "min": round(min(lat), 4),
                                             "what": "one build, launch -> gndt_sync returns (host-timed, stream idle before)"}
        if hint:
            m2 = new_map(0)
            for _ in range(3):
                step(m2)
            m2.sync()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                step(m2)
            m2.sync()
            torch.cuda.synchronize()
            extras["no_hint_auto"] = {"ms_per_step": round((time.perf_counter() - t1) / a.steps * 1e3, 4),
                                      "what": "same workload, max_nodes_hint = 0 (node count learned from the previous build)"}
            del m2
        t1 = time.perf_counter()
        res = m.export()
        extras["d2h_export_ms"] = round((time.perf_counter() - t1) * 1e3, 3)
        extras["h2d_cloud_ms"] = round(t_h2d * 1e3, 3)
        extras["pcie_note"] = "pageable host memory, torch copy; never part of `value`"
        del res
        # host buffer -> map on the device through the C ABI (gndt_build = stage + build, then gndt_sync): pageable and pinned
        hb = {}
        for label, arr in (("pageable", host_pts), ("pinned", torch.from_numpy(host_pts).pin_memory())):
            for _ in range(2):
                m.create2DMap("slope", arr)
                m.sync()
            ts = []
            for _ in range(5):
                t1 = time.perf_counter()
                m.create2DMap("slope", arr)
                m.sync()
                ts.append((time.perf_counter() - t1) * 1e3)
            hb[label + "_ms"] = round(float(np.median(ts)), 3)
        hb["what"] = "gndt_build from a HOST buffer (H2D staging + build) + gndt_sync, median of 5; PCIe-inclusive, never `value`"
        extras["host_build"] = hb
    if world == 1 and not a.no_configs and not global_mode and rank == 0:
        watchdog(0)
        s2_host = np.concatenate([origin[None, :], host_pts], 0) if wname == "S2" and total == WORKLOADS["S2"]["points"] else None
        del pts
        torch.cuda.empty_cache()
        extras["configs"] = measure_configs(g, torch, s2_host)
        del s2_host
        try:
            extras["host_path"] = measure_host_path()
        except Exception as e:            # (a missing g++ must not cost the bench line)
            extras["host_path"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            extras["cost_flood"] = measure_cost_flood(g, torch)
        except Exception as e:
            extras["cost_flood"] = {"error": f"{type(e).__name__}: {e}"}
        pts = torch.from_numpy(host_pts).to(dev)

    parity_failed = False
    if rank == 0:
        ms_step = dt / a.steps * 1e3
        value = job_points / (dt / a.steps) / 1e6
        strat = m.STRATEGY_NAMES.get(m.last_strategy(), "?")
        # Dominant kernel = the longest phase.  Its algorithmic bytes (DESIGN.md "Roofline accounting"):
        # every kernel that streams the cloud moves 12 B/point; the bucket kernel also emits the nodes
        # (12 B/point + 76 B/node); node-proportional kernels move 76 B/node.
        kernel_of = m.KERNEL_OF_PHASE_BLOCKED if m.last_strategy() == 7 else m.KERNEL_OF_PHASE
        cand = {k: v for k, v in phases.items() if k in kernel_of}
        dom = max(cand, key=cand.get) if cand else None
        acc_ms = live.get(dom, cand.get(dom, float("nan"))) if dom else float("nan")
        timed_live = dom in live
        k_nodes = local_nodes if owner_mode else rank0_nodes    # what ONE launch of the kernel on rank 0 processed
        if dom in m.POINT_PHASES:
            alg_bytes = BYTES_PER_POINT * n_local
        elif dom in m.POINT_AND_NODE_PHASES:
            alg_bytes = BYTES_PER_POINT * n_local + BYTES_PER_NODE * k_nodes
        else:
            alg_bytes = BYTES_PER_NODE * k_nodes
        achieved = alg_bytes / (acc_ms * 1e-3) / 1e9 if acc_ms == acc_ms and acc_ms > 0 else None
        traffic, traffic_src = (None, None)
        if wname == "S2" and world == 1 and total == WORKLOADS["S2"]["points"]:
            traffic, traffic_src = committed_traffic(kernel_of.get(dom))
        path_bytes = BYTES_PER_POINT * job_points + BYTES_PER_NODE * nodes * (world if mode == "replicas" else 1)
        out = {
            "metric": "NDT grid-build throughput (bin + mean/cov + eigen + labels + ordering)",
            "value": round(value, 3), "unit": "Mpoints/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": "strong" if one_cloud else "weak", "vs_baseline": None,
            "dtype": "f64", "io_dtype": "f32", "data": "synthetic",
            "config": {"workload": W["desc"] + (f", {total} points" if total != W["points"] else ""),
                       "points_total": int(job_points), "points_per_gpu": int(n), "nodes": int(nodes), "columns": int(cols),
                       "slopes": int(slopes), "multi_gpu_mode": mode, "strategy": strat, "max_nodes_hint": int(hint),
                       "scene_generation_s": round(t_gen, 1)},
            "roofline": {"bound": "hbm", "kernel": kernel_of.get(dom),
                         "achieved": round(achieved, 2) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5) if achieved else None, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel_ms_measured_in_timed_region": timed_live, "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel_ms": round(acc_ms, 4) if acc_ms == acc_ms else None},
            "path_roofline": {"bytes": path_bytes, "achieved_GBps": round(path_bytes / (ms_step * 1e-3) / 1e9, 2),
                              "frac": round(path_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "phase_ms": phases,
            "retries_in_timed_region": int(retries_timed),
        }
        if retries_timed:
            out["valid"] = False
        if mode == "shards_without_exchange" and one_cloud:
            # The step-down of last resort changes the COMPUTATION (one map per rank, no global map of the one cloud): the line is
            # kept so that the failure is visible, but it is not a measurement of the configuration.
            out["valid"] = False
            out["invalid_because"] = "multi_gpu_mode shards_without_exchange: each rank built a map of its own shard; the configuration asks for ONE map"
        out.update(multi)
        if global_mode:
            out["exchange"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in exch_timed.items()}
            out["exchange"]["ranks"] = world
            out["exchange"]["backend"] = ("rccl (called from C++ inside libgndt: gndt_build_owned_device — all-to-all of 16-B point records, "
                                          "all-gather of 8 B per column; the map stays sharded by column owner)") if owner_mode else \
                                         "rccl (called from C++ inside libgndt: gndt_build_global_device — statistics all-reduce, whole map on every rank)"
        out.update(extras)
        if not a.no_cpu_baseline and world == 1:
            sample = a.cpu_sample or min(n, 10_000_000)
            out["cpu_baseline"] = cpu_baseline(origin, host_pts, P, min(sample, n))
            if min(sample, n) >= n and not global_mode:      # the whole workload fits the CPU leg: check the whole map, too
                out["parity_full_size"] = full_size_parity(m, torch, origin, host_pts, P, pts, stream, dense=(wname == "S5"))
                parity_failed = not out["parity_full_size"]["ok"]
        elif world == 1:
            out["cpu_baseline"] = None
        if a.check:
            from tests import parity
            from grid_ndt_amd import scenes
            small = scenes.uniform_box(300_001)
            ref = parity.ref_from_cloud(small, P)
            _, o = parity.gpu_from_cloud(small, P, device=local)
            out["check"] = parity.compare(o, ref)["ok"]
        if a.stamps and m.last_strategy() in (2, 3, 4, 6, 7):
            m.enable_stamps(True)
            m.create2DMap("slope", pts, stream)
            cyc, nb = m.debug_bucket_phases()
            print("bucket kernel phase stamps (mean shader cycles per bucket, %d buckets): " % nb +
                  ", ".join(f"{k}={v:.0f}" for k, v in cyc.items() if k != "-"), file=sys.stderr)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()
    if retries_timed:
        sys.exit(3)
    if parity_failed:
        print("bench.py: ERROR the full-size map differs from the oracle's (parity_full_size.fail)", file=sys.stderr)
        sys.exit(4)


if __name__ == "__main__":
    main()
