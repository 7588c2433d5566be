# A/B of compile-time variants of the flood kernels: tools/ab_cost_flags.sh "<flags>" ...  (tools/measure_cost.py with and without the one-workgroup kernel, kernel statistics on bridge_ground)
for F in "$@"; do
  GNDT_EXTRA_CXXFLAGS="$F" python3 -c "import grid_ndt_amd as g; g.build_native(force=True)" > /dev/null 2>&1 || { echo "[$F] BUILD FAILED"; continue; }
  for W in 1 0; do
  echo "[$F] GNDT_COST_WG=$W"; GNDT_COST_WG=$W python tools/measure_cost.py --points 8000000 2>/dev/null | grep -E "gpu_ms|gpu_us_per_level|parity_h" | paste - - -
  done
  GNDT_COST_WG=0 bash tools/prof_cost.sh r04_abf 2>&1 | grep "^void gndt::k_cost_level\|^gndt::k_cost_flood\|^void gndt::k_cost_flood"
done
GNDT_EXTRA_CXXFLAGS="" python3 -c "import grid_ndt_amd as g; g.build_native(force=True)" > /dev/null 2>&1
