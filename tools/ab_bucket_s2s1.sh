#!/bin/bash
# A/B of compile-time variants of the bucket kernel on the GPU box: S2 and the 200 k-point campus frame, step + phase times + stamps.
#   tools/ab_bucket_s2s1.sh "<flags A>" "<flags B>" ...   (each rebuilt in place; "" = the tree's defaults)
WLS=${WLS:-"S2|S1"}
for F in "$@"; do
  GNDT_EXTRA_CXXFLAGS="$F" python3 -c "import grid_ndt_amd as g; g.build_native(force=True)" > /tmp/ab_build.txt 2>&1 || { echo "[$F] BUILD FAILED"; grep -v "^/opt/rocm/bin/hipcc" /tmp/ab_build.txt | tail -15; continue; }
  echo "$WLS" | tr '|' '\n' | while read W; do
    python3 bench.py --workload $W --steps 20 --no-cpu-baseline --no-extras --no-configs --stamps 2> /tmp/ab_err.txt | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('[$F] $W ms', d['ms_per_step'], 'nodes', d['config']['nodes'], 'retries', d.get('retries_in_timed_region'), {k:round(v,4) for k,v in d['phase_ms'].items() if v>0.003})
"
    grep -h "stamps" /tmp/ab_err.txt | cut -c1-300
  done
done
GNDT_EXTRA_CXXFLAGS="" python3 -c "import grid_ndt_amd as g; g.build_native(force=True)" > /dev/null 2>&1
