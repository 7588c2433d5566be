#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) into
the per-kernel JSON kept under profiles/:  summarise_pmc.py <fetch_dir> <write_dir> > profiles/<name>_pmc.json"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        tot[name] += float(r["Counter_Value"])
        n[name] += 1
    return tot, n


def main():
    fetch, nf = load(sys.argv[1], "FETCH_SIZE")
    write, nw = load(sys.argv[2], "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 3 --warmup 1, S2 10M points; "
                   "KiB per launch as reported; hbm_bytes_corrected applies MI355X_MICROARCH.md's gfx950 correction "
                   "(FETCH_SIZE x2 for wide coalesced reads)", "kernels": {}}
    # identity of the code the counters belong to: bench.py only quotes `traffic` when it matches the loaded sources
    from grid_ndt_amd import _lib
    out["source_hash"] = _lib.source_hash()
    try:
        out["git_head"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        out["git_head"] = None          # the GPU box's snapshot carries no .git: tools/collect_profiles.sh notes HEAD beside it
    total = 0
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("gndt::"):
            continue
        f = fetch.get(k, 0.0) / max(nf.get(k, 1), 1)
        w = write.get(k, 0.0) / max(nw.get(k, 1), 1)
        out["kernels"][k] = {"launches": int(nf.get(k, nw.get(k, 0))), "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                             "hbm_bytes_corrected": int((2 * f + w) * 1024)}
        total += out["kernels"][k]["hbm_bytes_corrected"] * out["kernels"][k]["launches"]
    steps = max((v["launches"] for k, v in out["kernels"].items() if "k_emit_rows" in k), default=1)
    out["hbm_bytes_per_build_all_kernels"] = int(total / max(steps, 1))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
