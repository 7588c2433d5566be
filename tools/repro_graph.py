#!/usr/bin/env python3
"""Replays what tools/fuzz_graph.py dumped before a replay that killed the process (FUZZ_DUMP=file.npz).
   python3 tools/repro_graph.py file.npz eager|graph|graph_other_first"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import grid_ndt_amd as g

d = np.load(sys.argv[1])
mode = sys.argv[2]
first = np.load(sys.argv[3])["other"] if len(sys.argv) > 3 else None      # (the cloud of the replay before the fatal one)
base, other, cells, strategy = d["base"], d["other"], d["cells"], int(d["strategy"])
n = base.shape[0] - 1
print("points", n, "cells", cells, "strategy", strategy, "replay", int(d["replay"]), flush=True)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    m = g.TwoDmap(float(cells[0]), float(cells[1]), strategy=strategy, max_nodes_hint=0, max_points_hint=n + 1)
    m.setInterval(0.08)
    m.setCloudFirst(base[0])
    if mode == "eager":
        buf = torch.from_numpy(np.ascontiguousarray(other[1:])).cuda()
        for k in range(3):
            m.create2DMap("slope", buf, s)
            print("eager build of `other`", k, m.sync(), m.retry_count(), flush=True)
    else:
        buf = torch.from_numpy(np.ascontiguousarray(base[1:])).cuda()
        for _ in range(2):
            m.create2DMap("slope", buf, s)
            print("eager build of `base`", m.sync(), m.retry_count(), m.STRATEGY_NAMES.get(m.last_strategy()), flush=True)
        gr = torch.cuda.CUDAGraph()
        with g.graph_capture(gr, s):
            m.create2DMap("slope", buf, s)
        print("captured", flush=True)
        seq = [other, other] if mode == "graph" else ([first, other, first, other] if mode == "graph_pair" else [base, other, base, other])
        for k, c in enumerate(seq):
            buf.copy_(torch.from_numpy(np.ascontiguousarray(c[1:])))
            gr.replay()
            s.synchronize()
            print("replayed", k, flush=True)
            try:
                out = m.export()
                print("  export ok", out["num_nodes"], flush=True)
            except Exception as e:
                print("  export:", str(e)[:200], flush=True)
print("done", flush=True)
