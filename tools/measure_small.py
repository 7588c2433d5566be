#!/usr/bin/env python3
"""Small-cloud builds (BASELINE configs[0]: one ~200 k-point frame): eager, back to back and as a replayed hipGraph.
python3 tools/measure_small.py -> JSON (also usable under rocprofv3 --kernel-trace --stats)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import grid_ndt_amd as g
    from grid_ndt_amd import scenes
    out = {}
    for name, cloud, P in (("S1_campus_200k", scenes.campus_frame(200_000), scenes.CAMPUS_PARAMS),
                           ("S1_bridge_ground", scenes.bridge_ground(), scenes.BRIDGE_PARAMS),
                           ("S1_depth_frame", scenes.depth_frame(), scenes.DEPTH_PARAMS)):
        pts = torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda()
        m = g.TwoDmap(P["grid_len"], P["z_len"])
        m.setInterval(P["slope_interval"])
        m.setCloudFirst(cloud[0])
        for _ in range(5):
            m.create2DMap("slope", pts)
            m.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            m.create2DMap("slope", pts)
        nodes = m.sync()[0]
        t_pipe = (time.perf_counter() - t0) / 50 * 1e3
        lat = []
        for _ in range(20):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            m.create2DMap("slope", pts)
            m.sync()
            lat.append((time.perf_counter() - t1) * 1e3)
        # the same build captured once and replayed
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            m.create2DMap("slope", pts, s)
            m.sync()
            graph = torch.cuda.CUDAGraph()
            with g.graph_capture(graph, s):
                m.create2DMap("slope", pts, s)
            for _ in range(3):
                graph.replay()
            s.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                graph.replay()
            s.synchronize()
            t_graph = (time.perf_counter() - t0) / 50 * 1e3
            gl = []
            for _ in range(20):
                s.synchronize()
                t1 = time.perf_counter()
                graph.replay()
                s.synchronize()
                gl.append((time.perf_counter() - t1) * 1e3)
        nodes_g = m.sync()[0]
        out[name] = {"points": int(pts.shape[0]), "nodes": int(nodes), "nodes_after_graph_replay": int(nodes_g),
                     "eager_back_to_back_ms": round(t_pipe, 4), "eager_single_latency_ms": round(float(np.median(lat)), 4),
                     "graph_back_to_back_ms": round(t_graph, 4), "graph_single_latency_ms": round(float(np.median(gl)), 4)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
