#!/usr/bin/env python3
"""Randomised parity campaign on the GPU box (test infrastructure: the oracle is the checker).

What the fixed-seed tests do not reach: ONE handle building a SEQUENCE of unrelated clouds (what the handle learnt from the
last cloud — node count, table size, load, the locality answer, two-level failures — is wrong for the next one), mid-size
clouds (10 k .. 3 M points) of mixed structure, every strategy, with and without hints, builds interleaved with update
streams and resets.  Every map is compared with the oracle (tests/parity.py gates).

    python3 tools/fuzz_campaign.py [--seconds 600] [--seed 1] [--max-points 3000000]  -> JSON summary; exit 1 on a failure
"""
import argparse
import json
import os
import sys
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CELLS = [(0.5, 0.1), (0.5, 0.5), (0.2, 0.2), (0.1, 0.05), (1.0 / 3.0, 0.07), (0.25, 0.5), (1.0, 0.1), (0.1, 0.1)]


def make_cloud(rng, n, cells):
    """A body of ~n points of random structure + whether it is lattice-adversarial (points on cell boundaries)."""
    gl, zl = cells
    origin = (rng.random(3) * 40 - 20).astype(np.float32)
    kind = int(rng.integers(0, 8))
    adversarial = False
    if kind == 0:      # uniform box, shuffled
        ext = rng.choice([5.0, 30.0, 120.0])
        body = origin + (rng.random((n, 3)) * 2 - 1) * np.float32([ext, ext, rng.choice([0.5, 2.0, 8.0])])
    elif kind == 1:    # scan-ordered ground raster with walls
        side = int(np.sqrt(n)) + 1
        u, v = np.meshgrid(np.arange(side), np.arange(side))
        step = rng.choice([0.01, 0.03, 0.1])
        x, y = u.ravel()[:n] * step, v.ravel()[:n] * step
        z = 0.3 * np.sin(x / 3.0) + 0.2 * np.cos(y / 2.0) + rng.normal(0, 0.01, n)
        wall = rng.random(n) < 0.1
        z[wall] = rng.random(wall.sum()) * 3.0
        body = origin + np.stack([x, y, z], 1)
    elif kind == 2:    # a few dense clusters (hundreds .. thousands of points per node) + sparse far points
        k = int(rng.integers(1, 6))
        c = origin + (rng.random((k, 3)) * 20 - 10)
        which = rng.integers(0, k, n)
        body = c[which] + rng.normal(0, [0.2, 0.2, 0.05], (n, 3))
        far = rng.random(n) < 0.02
        body[far] = origin + (rng.random((far.sum(), 3)) * 2 - 1) * [800, 800, 20]
    elif kind == 3:    # lattice points and their float neighbours
        ij = rng.integers(-60, 60, (n, 3)).astype(np.float64)
        lat = (origin.astype(np.float64) + ij * [gl, gl, zl]).astype(np.float32)
        pick = rng.integers(0, 3, n)
        body = np.where(pick[:, None] == 0, lat, np.where(pick[:, None] == 1, np.nextafter(lat, np.float32(np.inf)), np.nextafter(lat, np.float32(-np.inf))))
        adversarial = True
    elif kind == 4:    # zero padding runs inside ordinary ground (the converters' pre-allocation)
        body = origin + (rng.random((n, 3)) * 2 - 1) * np.float32([25, 25, 0.3])
        a = int(rng.integers(0, max(n // 2, 1)))
        body[a:a + int(rng.integers(64, max(n // 3, 65)))] = 0.0
    elif kind == 5:    # tall columns: a forest of poles
        k = int(rng.integers(1, 40))
        xy = origin[:2] + (rng.random((k, 2)) * 30 - 15)
        which = rng.integers(0, k, n)
        body = np.concatenate([xy[which] + rng.random((n, 2)) * 0.05, (origin[2] + rng.random(n) * rng.choice([5.0, 60.0]) - 2.0)[:, None]], 1)
    elif kind == 6:    # LiDAR-like rings over rolling ground
        az = rng.random(n) * 2 * np.pi
        ring = rng.integers(0, 64, n)
        r = 2.0 + ring * rng.choice([0.3, 0.8])
        x, y = r * np.cos(az), r * np.sin(az)
        z = 0.5 * np.sin(x / 7.0) * np.cos(y / 5.0) + rng.normal(0, 0.02, n) - 1.5
        order = np.lexsort((az, ring))
        body = origin + np.stack([x, y, z], 1)[order]
    else:              # mixture of two of the above halves, concatenated unshuffled
        a, adv_a = make_cloud(rng, n // 2 + 1, cells)
        b, adv_b = make_cloud(rng, n - n // 2, cells)
        return np.concatenate([a, b[1:]], 0).astype(np.float32), adv_a or adv_b
    body = np.asarray(body, dtype=np.float32)
    if rng.random() < 0.3:
        body = body[rng.permutation(body.shape[0])]
    return np.concatenate([origin[None, :], body], 0).astype(np.float32), adversarial


def run(seconds=600.0, seed=1, max_points=3_000_000, max_handles=0):
    import torch
    import grid_ndt_amd as g
    from tests import parity
    g.build_native()
    a = argparse.Namespace(seconds=seconds, seed=seed, max_points=max_points)
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    stats = {"handles": 0, "builds": 0, "updates": 0, "points": 0, "by_strategy_ran": {}, "failures": []}
    trial = 0
    while time.time() < t_end and len(stats["failures"]) < 5 and (not max_handles or trial < max_handles):
        trial += 1
        cells = CELLS[int(rng.integers(0, len(CELLS)))]
        strategy = int(rng.choice([0, 0, 1, 1, 2, 2, 3, 4, 5]))
        demand = "true" if rng.random() < 0.15 else "slope"
        interval = float(rng.choice([0.08, 0.08, 0.05, 0.2]))
        min_points = int(rng.choice([3, 3, 1, 2, 5]))
        P = dict(grid_len=cells[0], z_len=cells[1], slope_interval=interval, demand=demand, min_points=min_points)
        hint_kind = int(rng.integers(0, 3))          # none / far too low / generous
        hint = [0, 500, 4_000_000][hint_kind]
        m = g.TwoDmap(cells[0], cells[1], max_nodes_hint=hint, strategy=strategy, min_points=min_points)
        m.setInterval(interval)
        stats["handles"] += 1
        desc = None
        try:
            for step in range(int(rng.integers(2, 6))):
                n = int(np.exp(rng.uniform(np.log(200), np.log(a.max_points))))
                cloud, adv = make_cloud(rng, n, cells)
                if rng.random() < 0.15:              # a few points far out, up to ~60 000 cells from the origin (the key range ends at 65 535)
                    k = int(rng.integers(1, 50))
                    far = cloud[0] + (rng.random((k, 3)) * 2 - 1) * np.float32([60000 * cells[0], 60000 * cells[0], 3000 * cells[1]])
                    at = rng.integers(1, cloud.shape[0], k)
                    cloud[at] = far.astype(np.float32)
                layout = int(rng.integers(0, 4))     # device [n,3] / device PointXYZ records [n,4] / host [n,3] / host [n,4]
                desc = dict(trial=trial, step=step, seed=a.seed, cells=cells, strategy=strategy, hint=hint, demand=demand, points=int(cloud.shape[0] - 1),
                            interval=interval, min_points=min_points, layout=layout)
                body = np.ascontiguousarray(cloud[1:])
                if layout in (1, 3):
                    body = np.ascontiguousarray(np.concatenate([body, np.ones((body.shape[0], 1), np.float32)], 1))
                dev = torch.from_numpy(body).cuda() if layout < 2 else body
                streamed = strategy == 1 and hint != 500 and layout < 2 and rng.random() < 0.6 and cloud.shape[0] > 10     # (a stream does not outgrow a hint: documented)
                m.setCloudFirst(cloud[0])
                if streamed:                        # the same cloud as an update stream of uneven frames from an empty map
                    m.reset(demand)
                    cuts = np.unique(np.concatenate([[0, dev.shape[0]], rng.integers(0, dev.shape[0], int(rng.integers(1, 40)))]))
                    check_at = int(rng.integers(1, len(cuts)))              # one checkpoint inside the stream, then the end
                    for fi, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
                        m.change2DMap(demand, dev[int(lo):int(hi)])
                        stats["updates"] += 1
                        if fi + 1 == check_at and fi + 1 < len(cuts) - 1:
                            part = cloud[:int(hi) + 1]
                            rp = parity.compare(m.export(), parity.ref_from_cloud(part, P, mode=2), demand, adversarial=adv, dense=True,
                                                interval=interval, min_points=min_points)
                            stats["checkpoints"] = stats.get("checkpoints", 0) + 1
                            if not rp["ok"]:
                                stats["failures"].append(dict(desc, at_frame=fi + 1, frames=int(len(cuts) - 1), fail=rp["fail"][:5]))
                    desc["streamed_frames"] = int(len(cuts) - 1)
                else:
                    m.create2DMap(demand, dev)
                out = m.export()
                ran = m.STRATEGY_NAMES.get(m.last_strategy(), str(m.last_strategy()))
                stats["by_strategy_ran"][ran] = stats["by_strategy_ran"].get(ran, 0) + 1
                ref = parity.ref_from_cloud(cloud, P, mode=2)
                rep = parity.compare(out, ref, demand, adversarial=adv, dense=True, interval=interval, min_points=min_points)
                stats["builds"] += 1
                stats["points"] += int(cloud.shape[0] - 1)
                for k in ("labels_within_margin",):
                    stats[k] = stats.get(k, 0) + int(rep.get(k, 0))
                stats["labels_on_the_margin"] = stats.get("labels_on_the_margin", 0) + int(rep.get("labels_on_the_margin", 0))
                stats["worst_mean_err_truth"] = max(stats.get("worst_mean_err_truth", 0.0), rep.get("mean_err_truth", 0.0))
                stats["worst_cov_err_truth"] = max(stats.get("worst_cov_err_truth", 0.0), rep.get("cov_err_truth", 0.0))
                stats["nodes"] = stats.get("nodes", 0) + int(rep.get("num_nodes", 0))
                if not rep["ok"]:
                    f = dict(desc, ran=ran, fail=rep["fail"][:5])
                    if "rough_worst_node" in rep:
                        i = rep["rough_worst_node"]
                        f["node"] = dict(i=i, key=[int(ref[k][i]) for k in ("sx", "sy", "sz")], count=int(ref["count"][i]),
                                         cov64=ref["cov64"][i].tolist(), cov_gpu=out["cov"][i].astype(float).tolist(), cov_ref32=ref["cov"][i].astype(float).tolist(),
                                         evals64=ref["evals64"][i].tolist(), rough_gpu=float(out["rough"][i]), rough_ref=float(ref["rough"][i]),
                                         mean64=ref["mean64"][i].tolist(), flags_gpu=int(out["flags"][i]), flags_ref=int(ref["flags"][i]))
                    stats["failures"].append(f)
                    break
                del dev, out, ref
        except Exception as e:            # an error code from the library is a failure of the campaign, too
            stats["failures"].append(dict(desc or {}, error=f"{type(e).__name__}: {e}", trace=traceback.format_exc()[-800:]))
        del m
    stats["seconds"] = round(a.seconds - max(0.0, t_end - time.time()), 1)
    return stats


def run_threads(seconds, seed, max_points, threads):
    """`threads` campaigns at once, each on its own torch stream with its own handles and random sequence: the GPU is shared, so
    anything that is not ordered on the handle's stream (a null-stream fill, a buffer freed under a kernel) gets its chance."""
    import threading
    import torch
    res = [None] * threads

    def body(k):
        torch.cuda.set_device(0)
        with torch.cuda.stream(torch.cuda.Stream()):
            res[k] = run(seconds, seed * 1000 + k, max_points)

    th = [threading.Thread(target=body, args=(k,)) for k in range(threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    total = {"threads": threads, "failures": []}
    for r in res:
        for key, v in (r or {"failures": [{"error": "a campaign thread died"}]}).items():
            if key == "failures":
                total["failures"] += v
            elif isinstance(v, dict):
                d = total.setdefault(key, {})
                for kk, vv in v.items():
                    d[kk] = d.get(kk, 0) + vv
            elif key.startswith("worst") or key == "seconds":
                total[key] = max(total.get(key, 0), v)
            else:
                total[key] = total.get(key, 0) + v
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=600.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-points", type=int, default=3_000_000)
    ap.add_argument("--threads", type=int, default=1, help="independent campaigns at once, sharing the GPU")
    a = ap.parse_args()
    stats = run_threads(a.seconds, a.seed, a.max_points, a.threads) if a.threads > 1 else run(a.seconds, a.seed, a.max_points)
    print(json.dumps(stats, indent=1))
    sys.exit(1 if stats["failures"] else 0)


if __name__ == "__main__":
    main()
