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


if __name__ == "__main__":
    main()
