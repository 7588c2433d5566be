for S in 256 512; do
  GNDT_BUCKET_SLOTS=$S python3 bench.py --workload S2 --steps 10 --no-cpu-baseline --no-extras --stamps 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('slots $S ms', d['ms_per_step'], d['phase_ms']['bucket_build'])
    elif 'stamps' in l: print(l.strip()[:300])
"
done
