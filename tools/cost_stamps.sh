# Where a layer of the one-workgroup flood kernel spends its cycles (a diagnostic build: -DGNDT_COST_STAMPS, gndt_cost.hpp), on the
# GPU box:   bash tools/cost_stamps.sh > gpurun_out/cost_stamps.txt 2>&1       (rebuilds the library twice: with and without the stamps)
touch grid_ndt_amd/csrc/gndt_cost.hpp
GNDT_EXTRA_CXXFLAGS=-DGNDT_COST_STAMPS python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
python - <<'PY'
import numpy as np, torch, sys
import grid_ndt_amd as g
from grid_ndt_amd import scenes
for name, cloud, P, goal in (("bridge_ground", scenes.bridge_ground(), scenes.BRIDGE_PARAMS, None),
                             ("drivable_site", scenes.drivable_site(), scenes.COST_PARAMS, scenes.DRIVABLE_GOAL)):
    m = g.TwoDmap(P["grid_len"], P["z_len"])
    m.setInterval(P["slope_interval"])
    m.setCloudFirst(cloud[0])
    m.create2DMap(P.get("demand", "slope"), torch.from_numpy(np.ascontiguousarray(cloud[1:])).cuda())
    cells = m.export()
    if goal is None:
        rows = np.nonzero((cells["flags"] & 2) != 0)[0]
        goal = cells["mean"][rows[len(rows) // 3]]
    for _ in range(3):
        st = m.computeCost(goal)
    print(name, st, file=sys.stderr)
PY
touch grid_ndt_amd/csrc/gndt_cost.hpp
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
