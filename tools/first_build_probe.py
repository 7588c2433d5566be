#!/usr/bin/env python3
"""First build of a fresh handle, verbose: which re-runs it needs and why (stderr lines of libgndt).  GPU box."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import grid_ndt_amd as g
    from grid_ndt_amd import scenes
    g.TwoDmap.set_debug_option(g.TwoDmap.DEBUG_VERBOSE, 1)
    for name, cloud, P in (("campus_120k", scenes.campus_frame(120_001), scenes.CAMPUS_PARAMS),
                           ("campus_200k", scenes.campus_frame(200_001), scenes.CAMPUS_PARAMS),
                           ("bridge_ground", scenes.bridge_ground(), scenes.BRIDGE_PARAMS)):
        pts = torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda()
        m = g.TwoDmap(P["grid_len"], P["z_len"])
        m.setInterval(P["slope_interval"])
        m.setCloudFirst(cloud[0])
        m.create2DMap("slope", pts)
        nodes, cols, slopes = m.sync()
        print(name, "nodes", nodes, "re-runs of the first build:", m.retry_count(), "strategy", m.STRATEGY_NAMES[m.last_strategy()], flush=True)
        m.create2DMap("slope", pts)
        m.sync()
        print(name, "re-runs after the second build:", m.retry_count(), flush=True)


def warmed():
    """gndt_warmup + reserve before the clock: the first build's call and wait, against the second build's."""
    import time
    import torch
    import grid_ndt_amd as g
    from grid_ndt_amd import scenes
    for name, cloud, P in (("campus_200k", scenes.campus_frame(200_001), scenes.CAMPUS_PARAMS), ("bridge_ground", scenes.bridge_ground(), scenes.BRIDGE_PARAMS)):
        pts = torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda()
        n = cloud.shape[0] - 1
        for rep in range(3):
            m = g.TwoDmap(P["grid_len"], P["z_len"], max_points_hint=n)
            m.setInterval(P["slope_interval"])
            m.setCloudFirst(cloud[0])
            t0 = time.perf_counter()
            m.warmup(n)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            m.create2DMap("slope", pts)
            t2 = time.perf_counter()
            m.sync()
            t3 = time.perf_counter()
            m.create2DMap("slope", pts)
            t4 = time.perf_counter()
            m.sync()
            t5 = time.perf_counter()
            print(name, "warmup %.2f ms | first build: call %.3f wait %.3f | second: call %.3f wait %.3f ms | re-runs %d" %
                  ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, m.retry_count()), flush=True)
            del m


if __name__ == "__main__":
    main()
    warmed()
