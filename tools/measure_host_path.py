#!/usr/bin/env python3
"""The drop-in's host path for one configs[0]-sized frame: host cloud -> gndt_build (H2D + build + sync) -> gndt_export (D2H),
the PCIe-inclusive latency a ROS callback sees (never bench.py's `value`).  python3 tools/measure_host_path.py"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import grid_ndt_amd as g
from grid_ndt_amd import scenes
for name, cloud, P in (("S1_campus_200k", scenes.campus_frame(200_000), scenes.CAMPUS_PARAMS),):
    m = g.TwoDmap(P["grid_len"], P["z_len"]); m.setInterval(P["slope_interval"]); m.setCloudFirst(cloud[0])
    host = np.ascontiguousarray(cloud[1:])
    pinned = torch.from_numpy(host).pin_memory()
    for label, arr in (("pageable numpy", host), ("pinned torch tensor", pinned)):
        for _ in range(5): m.create2DMap("slope", arr)
        lat = []
        for _ in range(30):
            t = time.perf_counter(); m.create2DMap("slope", arr); m.sync(); lat.append((time.perf_counter() - t) * 1e3)
        print(name, label, "host input -> gndt_build (H2D + build + sync) p50 %.3f ms min %.3f" % (np.median(lat), min(lat)))
    dev = torch.from_numpy(host).cuda()
    lat = []
    for _ in range(30):
        torch.cuda.synchronize(); t = time.perf_counter(); m.create2DMap("slope", dev); m.sync(); lat.append((time.perf_counter() - t) * 1e3)
    print(name, "device input p50 %.3f ms" % np.median(lat))
    t = time.perf_counter(); out = m.export(); print("export (D2H of the SoA, %d nodes) %.3f ms" % (out["num_nodes"], (time.perf_counter() - t) * 1e3))
