# strategy TILE (5) against AUTO/PARTITION on the scan-ordered scenes:  tools/ab_tile.sh
python3 -m pytest tests -m gpu -x -q > gpurun_out/r02_t.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/r02_t.log | tail -3
for ST in 5 2; do
 for W in "S1" "S3 --points 32000000" "S5"; do
  python3 bench.py --workload $W --strategy $ST --steps 10 --no-cpu-baseline --no-extras 2>gpurun_out/tile.err | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('strategy $ST $W ms', d['ms_per_step'], d['config']['strategy'], {k:v for k,v in d['phase_ms'].items() if v>0.02})
"
 done
done
python3 - <<'PY'
import torch, numpy as np, grid_ndt_amd as g
from grid_ndt_amd import scenes
for name, cloud, P in (("S1 campus", scenes.campus_frame(200000), scenes.CAMPUS_PARAMS), ("bridge", scenes.bridge_ground(), scenes.BRIDGE_PARAMS),
                       ("S3 8M", scenes.terrain_cloud(8_000_000), dict(grid_len=0.2, z_len=0.2)), ("S5 4M", scenes.site_two_storey(4_000_000), dict(grid_len=0.1, z_len=0.1)),
                       ("S2 2M", scenes.uniform_box(2_000_000), dict(grid_len=0.5, z_len=0.5))):
    m = g.TwoDmap(P["grid_len"], P["z_len"]); m.setInterval(0.08); m.setCloudFirst(cloud[0])
    print(name, "points per partial: %.2f" % m.locality_sample(torch.from_numpy(cloud[1:]).cuda()))
PY
