import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import grid_ndt_amd as g
from grid_ndt_amd import scenes
from tests import parity
g.TwoDmap.set_debug_option(g.TwoDmap.DEBUG_VERBOSE, 1)
for name, cloud, P in (("uniform 6M", scenes.uniform_box(6_000_001), dict(grid_len=0.5, z_len=0.5, slope_interval=0.08)),
                       ("uniform 6M z0.25", scenes.uniform_box(6_000_001), dict(grid_len=0.5, z_len=0.25, slope_interval=0.08))):
    ref = parity.ref_from_cloud(cloud, P, mode=2, threads=0)
    pts = torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda()
    m = g.TwoDmap(P["grid_len"], P["z_len"])
    m.setInterval(P["slope_interval"]); m.setCloudFirst(cloud[0])
    for k in range(3):
        m.create2DMap("slope", pts)
        out = m.export()
        rep = parity.compare(out, ref)
        print(name, "build", k, m.STRATEGY_NAMES[m.last_strategy()], "ok" if rep["ok"] else rep["fail"][:3], "re-runs", m.retry_count(), flush=True)
    # another cloud of the same size on the same handle: it does not fit the box -> hashed re-run, same map as the oracle's
    other = scenes.uniform_box(6_000_001, seed=77) * np.float32(1.3)
    ref2 = parity.ref_from_cloud(other, P, mode=2, threads=0)
    m.setCloudFirst(other[0]) if False else None
    m2pts = torch.from_numpy(np.ascontiguousarray(other[1:])).cuda()
    m.create2DMap("slope", m2pts)
    out = m.export()
    ref2 = parity.ref_from_cloud(np.concatenate([cloud[:1], other[1:]]), P, mode=2, threads=0)
    rep = parity.compare(out, ref2)
    print(name, "other cloud", m.STRATEGY_NAMES[m.last_strategy()], "ok" if rep["ok"] else rep["fail"][:3], "re-runs", m.retry_count(), flush=True)
