"""grid_ndt_amd — MI355X-native NDT voxel-grid build (the uniformDivision -> create2DMap path of
daysun/grid_ndt) behind a C ABI.  See DESIGN.md; the C ABI is include/gndt.h."""
from ._lib import GndtError, build_native, LIB_PATH  # noqa: F401
from .map2d import (TwoDmap, graph_capture, read_pcd, count_morton, morton_to_xy, trans_morton_xyz, device_info,  # noqa: F401
                    FLAG_HAS_STATS, FLAG_SLOPE, FLAG_DOWN)
