"""The synthetic scenes live in the package (grid_ndt_amd/scenes.py) so that bench.py and the tools do not
import the test tree; the tests keep their `from tests import scenes` spelling through this re-export."""
from grid_ndt_amd.scenes import *  # noqa: F401,F403
from grid_ndt_amd.scenes import (BRIDGE_PARAMS, CAMPUS_PARAMS, COST_PARAMS, DRIVABLE_GOAL, FRAME_POINTS, RINGS, AZ,  # noqa: F401
                                 _sweep, _pose_xy)
