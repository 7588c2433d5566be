# A/B of compile-time variants with stamps on the GPU box:  tools/ab_flags_stamps.sh "<flags A>" "<flags B>" ... (each rebuilt in place)
for F in "$@"; do
  GNDT_EXTRA_CXXFLAGS="$F" python3 -c "import grid_ndt_amd as g; g.build_native(force=True)" > /dev/null 2>&1
  bash tools/ab_stamps.sh "F=$(echo $F | tr ' ' '_')"
done
GNDT_EXTRA_CXXFLAGS="" python3 -c "import grid_ndt_amd as g; g.build_native(force=True)" > /dev/null 2>&1
