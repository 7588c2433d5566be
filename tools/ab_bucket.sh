#!/bin/bash
# A/B of compile-time variants of the bucket kernel on the GPU box: S2 / S3 32 M / S5 step + phase times + stamps per variant.
#   tools/ab_bucket.sh "<flags A>" "<flags B>" ...   (each rebuilt in place; "" = the tree's defaults)
for F in "$@"; do
  GNDT_EXTRA_CXXFLAGS="$F" python3 -c "import grid_ndt_amd as g; g.build_native(force=True)" > /dev/null 2>&1 || { echo "[$F] BUILD FAILED"; continue; }
  for W in "S2" "S3 --points 32000000" "S5"; do
    python3 bench.py --workload $W --steps 10 --no-cpu-baseline --no-extras --no-configs --stamps 2> /tmp/ab_err.txt | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('[$F] $W ms', d['ms_per_step'], 'retries', d.get('retries_in_timed_region'), {k:round(v,4) for k,v in d['phase_ms'].items() if v>0.003})
"
    grep -h "stamps" /tmp/ab_err.txt | cut -c1-300
  done
done
GNDT_EXTRA_CXXFLAGS="" python3 -c "import grid_ndt_amd as g; g.build_native(force=True)" > /dev/null 2>&1
