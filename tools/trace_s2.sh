#!/bin/bash
# Run on the MI355X box: kernel trace of the bench workload -> per-kernel statistics and the timeline of ONE steady-state build
# (start, end and gap to the previous kernel).   tools/trace_s2.sh <tag> [bench args]
set -e
tag=${1:-prof}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_tr -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-configs "$@" > $out/${tag}_tr.log 2>&1
cp $out/${tag}_tr/*/*kernel_stats.csv $out/${tag}_kernel_stats.csv
python3 - "$out/${tag}_tr" > $out/${tag}_timeline.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last complete build: from the last k_part_clear on
idx = [i for i, r in enumerate(rows) if "k_part_clear" in r["Kernel_Name"]]
i0 = idx[-2] if len(idx) > 1 else max(0, len(rows) - 40)       # (no partition build in the trace: the last 40 launches)
i1 = idx[-1] if len(idx) > 1 else len(rows)
prev = None
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-60:]
    print(f"{(s - int(rows[i0]['Start_Timestamp'])) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {((s - prev) / 1e3) if prev else 0:6.1f}  {name}")
    prev = e
PY
rm -rf $out/${tag}_tr
head -16 $out/${tag}_kernel_stats.csv | cut -c1-150
cat $out/${tag}_timeline.txt
