#!/usr/bin/env python3
"""What a hipGraph launch costs on this runtime, without libgndt: eight dependent one-workgroup kernels (torch `add_` on a 256-element
tensor) launched eagerly back to back, and the same eight captured once and replayed back to back.  The difference per replay is the
runtime's own (graph launch + the completion signalling between two launches): a library whose recorded call runs the SAME kernels as
its eager call (profiles/r06_s4_graph_vs_eager.txt) cannot replay faster than this allows.  GPU box; JSON to stdout."""
import json
import time

import torch


def main():
    x = torch.zeros(256, device="cuda")
    s = torch.cuda.Stream()
    out = {}
    with torch.cuda.stream(s):
        def body():
            for _ in range(8):
                x.add_(1.0)
        for _ in range(20):
            body()
        s.synchronize()
        for name, reps in (("eager", 2000),):
            t0 = time.perf_counter()
            for _ in range(reps):
                body()
            s.synchronize()
            out["eager_8_kernels_us"] = round((time.perf_counter() - t0) / reps * 1e6, 2)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            body()
        for _ in range(20):
            g.replay()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000):
            g.replay()
        s.synchronize()
        out["graph_replay_8_kernels_us"] = round((time.perf_counter() - t0) / 2000 * 1e6, 2)
        # one awaited launch at a time
        lat_e, lat_g = [], []
        for _ in range(200):
            s.synchronize(); t0 = time.perf_counter(); body(); s.synchronize(); lat_e.append(time.perf_counter() - t0)
            s.synchronize(); t0 = time.perf_counter(); g.replay(); s.synchronize(); lat_g.append(time.perf_counter() - t0)
        lat_e.sort(); lat_g.sort()
        out["eager_awaited_p50_us"] = round(lat_e[100] * 1e6, 2)
        out["graph_awaited_p50_us"] = round(lat_g[100] * 1e6, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
