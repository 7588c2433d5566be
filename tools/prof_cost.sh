#!/bin/bash
# Kernel statistics of the cost flood on bridge_ground at its own parameters (tools/host_path), on the GPU box:
#   tools/prof_cost.sh <tag>   -> gpurun_out/<tag>_cost_kernel_stats.csv
. "$(dirname "$0")/_single_process_guard.sh"
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from grid_ndt_amd import scenes
c = scenes.bridge_ground()
np.ascontiguousarray(c, np.float32).tofile('/tmp/bridge.f32')
P = scenes.BRIDGE_PARAMS
open('/tmp/bridge.args', 'w').write(' '.join(['/tmp/bridge.f32', str(c.shape[0]), repr(P['grid_len']), repr(P['z_len']), repr(P['slope_interval']), P.get('demand', 'slope'),
                                              '9.5', '3.0', '1.0', '9.5', '3.0', '3.0', '0.25', '7']))
PY
python3 -c "import bench; print(bench.build_host_path_tool())" > /tmp/hp_exe.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_cost_trace -- $(tail -1 /tmp/hp_exe.txt) $(cat /tmp/bridge.args) > gpurun_out/${tag}_cost_trace.log 2>&1
cp gpurun_out/${tag}_cost_trace/*/*kernel_stats.csv gpurun_out/${tag}_cost_kernel_stats.csv
rm -rf gpurun_out/${tag}_cost_trace
cut -d, -f1-6 gpurun_out/${tag}_cost_kernel_stats.csv | cut -c1-60,200- | head -14
python3 - <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/${tag}_cost_kernel_stats.csv')))
for r in rows:
    if 'k_cost' in r['Name']:
        print(r['Name'][:60], r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3, 2), 'min', round(float(r['MinNs'])/1e3,2), 'max', round(float(r['MaxNs'])/1e3,2))
PY
