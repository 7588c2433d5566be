# A/B of the level-1 / level-2 variants on the GPU box: compile-time tile sizes x run-time knobs
#   tools/ab_l1.sh "<flags A>" "<flags B>" ...    ("" = the tree's defaults); ENVS="A=1|B=2" overrides the run-time sets
ENVS=${ENVS:-"GNDT_L1_INPLACE=0|GNDT_L1_INPLACE=1 GNDT_L2_ORDER=0|GNDT_L1_INPLACE=1 GNDT_L2_ORDER=1"}
WL=${WL:-"S2|S3 --points 32000000|S5"}
for F in "$@"; do
  GNDT_EXTRA_CXXFLAGS="$F" python3 -c "import grid_ndt_amd as g; g.build_native(force=True)" > /dev/null 2>&1 || { echo "[$F] BUILD FAILED"; continue; }
  echo "$ENVS" | tr '|' '\n' | while read E; do
    echo "$WL" | tr '|' '\n' | while read W; do
      env $E python3 bench.py --workload $W --steps 10 --no-cpu-baseline --no-extras --no-configs 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('[$F][$E] $W ms', d['ms_per_step'], 'retries', d['retries_in_timed_region'], {k:v for k,v in d['phase_ms'].items() if v>0.02})
    elif 'WARNING' in l or 'rror' in l: print(l.strip()[:300])
"
    done
  done
done
GNDT_EXTRA_CXXFLAGS="" python3 -c "import grid_ndt_amd as g; g.build_native(force=True)" > /dev/null 2>&1
