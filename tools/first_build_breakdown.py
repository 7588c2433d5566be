"""Where the first build of a FRESH handle spends its host time (GPU box): handle creation, the build call (allocations + launches),
the wait.  python3 tools/first_build_breakdown.py"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import grid_ndt_amd as g
from grid_ndt_amd import scenes
for name, cloud, P in (("campus", scenes.campus_frame(200_000), scenes.CAMPUS_PARAMS), ("bridge", scenes.bridge_ground(), scenes.BRIDGE_PARAMS), ("depth", scenes.depth_frame(), scenes.DEPTH_PARAMS)):
    pts = torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda()
    m0 = g.TwoDmap(P["grid_len"], P["z_len"]); m0.setInterval(P["slope_interval"]); m0.setCloudFirst(cloud[0]); m0.create2DMap("slope", pts); m0.sync()
    torch.cuda.synchronize()
    for rep in range(4):
        m = g.TwoDmap(P["grid_len"], P["z_len"])
        m.setInterval(P["slope_interval"]); m.setCloudFirst(cloud[0])
        t1 = time.perf_counter()
        m._ensure("slope")
        t2 = time.perf_counter()
        m.create2DMap("slope", pts)
        t3 = time.perf_counter()
        m.sync()
        t4 = time.perf_counter()
        m.create2DMap("slope", pts)
        t5 = time.perf_counter()
        m.sync()
        t6 = time.perf_counter()
        print(name, "gndt_create %.3f build-call %.3f sync %.3f | second build-call %.3f sync %.3f ms" % tuple(1e3 * x for x in (t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)), flush=True)
        del m
# the allocator's own price
import ctypes as C
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
for bytes_ in (4096, 1 << 20, 64 << 20):
    ps = [C.c_void_p() for _ in range(16)]
    t0 = time.perf_counter()
    for p in ps: hip.hipMalloc(C.byref(p), C.c_size_t(bytes_))
    t1 = time.perf_counter()
    for p in ps: hip.hipFree(p)
    t2 = time.perf_counter()
    print("hipMalloc %d B: %.1f us, hipFree %.1f us" % (bytes_, (t1 - t0) / 16 * 1e6, (t2 - t1) / 16 * 1e6))
ps = [C.c_void_p() for _ in range(8)]
t0 = time.perf_counter()
for p in ps: hip.hipHostMalloc(C.byref(p), C.c_size_t(4096), 0)
t1 = time.perf_counter()
print("hipHostMalloc 4096 B: %.1f us" % ((t1 - t0) / 8 * 1e6))
s = C.c_void_p()
t0 = time.perf_counter(); hip.hipStreamCreate(C.byref(s)); t1 = time.perf_counter()
print("hipStreamCreate: %.1f us" % ((t1 - t0) * 1e6))
e = C.c_void_p()
t0 = time.perf_counter(); hip.hipEventCreate(C.byref(e)); t1 = time.perf_counter()
print("hipEventCreate: %.1f us" % ((t1 - t0) * 1e6))
