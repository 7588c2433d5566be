"""First build of a FRESH handle (no hint) on the bench scenes: ms, re-runs, strategy; then the steady state (GPU box)."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import grid_ndt_amd as g
from grid_ndt_amd import scenes

CASES = [("S2_10M", lambda: scenes.uniform_box(10_000_000), 0.5, 0.5), ("S2z_10M_z01", lambda: scenes.uniform_box(10_000_000), 0.5, 0.1),
         ("S3_terrain_8M", lambda: scenes.terrain_cloud(8_000_000), 0.2, 0.2), ("S5_site_5M", lambda: scenes.site_two_storey(5_000_000), 0.1, 0.1),
         ("S5_site_20M", lambda: scenes.site_two_storey(20_000_000), 0.1, 0.1)]
out = {}
for name, gen, gl, zl in CASES:
    cloud = gen()
    pts = torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda()
    torch.cuda.synchronize()
    res = []
    for rep in range(2):
        t0 = time.perf_counter()
        m = g.TwoDmap(gl, zl)
        m.setInterval(0.08)
        m.setCloudFirst(cloud[0])
        m.create2DMap("slope", pts)
        nodes, _, _ = m.sync()
        ms = (time.perf_counter() - t0) * 1e3
        r1 = m.retry_count()
        for _ in range(3):
            m.create2DMap("slope", pts); m.sync()
        r2 = m.retry_count()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            m.create2DMap("slope", pts)
        m.sync(); torch.cuda.synchronize()
        res.append({"first_ms": round(ms, 3), "first_re_runs": int(r1), "later_re_runs": int(r2 - r1), "steady_ms": round((time.perf_counter() - t0) / 5 * 1e3, 4),
                    "nodes": int(nodes), "strategy": m.STRATEGY_NAMES[m.last_strategy()]})
        del m
    out[name] = res
    print(name, res, flush=True)
    del pts
print(json.dumps(out))
