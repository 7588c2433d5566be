# A/B of compile-time variants on the large workloads (GPU box):  tools/ab_flags_big.sh "<flags A>" "<flags B>" ...
for F in "$@"; do
  GNDT_EXTRA_CXXFLAGS="$F" python3 -c "import grid_ndt_amd as g; g.build_native(force=True)" > /dev/null 2>&1 || { echo "[$F] BUILD FAILED"; continue; }
  for W in "S2" "S3 --points 32000000" "S3" "S5"; do
    python3 bench.py --workload $W --steps 8 --no-cpu-baseline --no-extras --no-configs 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('[$F] $W ms', d['ms_per_step'], 'retries', d['retries_in_timed_region'], {k:v for k,v in d['phase_ms'].items() if v>0.02})
"
  done
done
GNDT_EXTRA_CXXFLAGS="" python3 -c "import grid_ndt_amd as g; g.build_native(force=True)" > /dev/null 2>&1
