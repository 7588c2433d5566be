# The flood with and without the one-workgroup kernel (gndt_debug_set_option GNDT_DEBUG_COST_ONE_WORKGROUP; GNDT_COST_WG is read by the two python entry points below, not by the library): the host path of the three S1 frames (bench.measure_host_path)
# and tools/measure_cost.py.   tools/ab_cost.sh > gpurun_out/ab_cost.txt
for W in 1 0; do
  export GNDT_COST_WG=$W
  echo "== GNDT_COST_WG=$W"
  python -c "
import bench, json, os
import grid_ndt_amd as g
g.TwoDmap.set_debug_option(g.TwoDmap.DEBUG_COST_ONE_WORKGROUP, int(os.environ.get('GNDT_COST_WG', '1')))
d = bench.measure_host_path()
for k, v in d.items():
    if isinstance(v, dict): print(k, v['flood_layers'], v['eager']['compute_cost_ms'], v['lazy']['compute_cost_ms'], v['oracle_as_shipped_ms']['calculate'])
" 2>&1 | tail -3
  python tools/measure_cost.py --points 8000000 2>/dev/null | grep -E "gpu_ms|first_flood|parity_h" | paste - - -
done
