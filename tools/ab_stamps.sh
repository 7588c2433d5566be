# A/B of run-time knobs with the bucket kernel's in-kernel phase stamps (GPU box):  tools/ab_stamps.sh "VAR=a" "VAR=b" ...
WL=${WL:-"S2"}
for E in "$@"; do
  echo "$WL" | tr '|' '\n' | while read W; do
    env $E python3 bench.py --workload $W --steps 10 --no-cpu-baseline --no-extras --no-configs --stamps 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('[$E] $W ms', d['ms_per_step'], 'retries', d['retries_in_timed_region'], 'nodes', d['config']['nodes'], {k:v for k,v in d['phase_ms'].items() if v>0.02})
    elif 'WARNING' in l or 'stamps' in l: print('   ', l.strip())
"
  done
done
