# A/B of compile-time variants on the GPU box:  tools/ab_flags.sh "<flags A>" "<flags B>" ... (each rebuilt in place)
for F in "$@"; do
  GNDT_EXTRA_CXXFLAGS="$F" python3 -c "import grid_ndt_amd as g; g.build_native(force=True)" > /dev/null 2>&1
  for W in "S2" "S3 --points 32000000"; do
    python3 bench.py --workload $W --steps 10 --no-cpu-baseline --no-extras --no-configs 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('[$F] $W ms', d['ms_per_step'], {k:v for k,v in d['phase_ms'].items() if v>0.02})
"
  done
done
