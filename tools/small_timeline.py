#!/usr/bin/env python3
"""Timeline of small builds from a rocprofv3 kernel trace (tools/trace_small.sh): for every distinct sequence of kernels between two
k_part_clear / k_clear launches, the median duration of each kernel and the median gap to the kernel before it.
    python3 tools/small_timeline.py gpurun_out/<tag>_small_kernel_trace.csv"""
import csv
import statistics
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    builds, cur = [], []
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gndt::", "")
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if name.startswith("k_part_clear") or name.startswith("k_clear_") or name.startswith("k_tile_begin"):
            if cur:
                builds.append(cur)
            cur = []
        cur.append((name, s, e))
    if cur:
        builds.append(cur)
    groups = defaultdict(list)
    for b in builds:
        groups[tuple(k[0] for k in b)].append(b)
    for seq, bs in sorted(groups.items(), key=lambda kv: -len(kv[1]))[:8]:
        if len(bs) < 20:
            continue
        print(f"== {len(bs)} builds of {len(seq)} launches")
        total = statistics.median(b[-1][2] - b[0][1] for b in bs) / 1e3
        pitch = statistics.median(bs[i + 1][0][1] - bs[i][0][1] for i in range(len(bs) - 1)) / 1e3
        for i, name in enumerate(seq):
            dur = statistics.median(b[i][2] - b[i][1] for b in bs) / 1e3
            gap = statistics.median(b[i][1] - b[i - 1][2] for b in bs) / 1e3 if i else 0.0
            print(f"   {name[:44]:44s} {dur:7.2f} us   gap before {gap:6.2f}")
        print(f"   first start -> last end {total:.2f} us; start-to-start of consecutive builds (median) {pitch:.2f} us")


if __name__ == "__main__":
    main()
