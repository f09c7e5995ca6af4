"""MI355X-native stitching hot path behind the reference's VideMosaic API.

The directory name is not a Python identifier; import it with
``importlib.import_module("real-time-video-mosaic_amd")`` (tests and bench.py
alias it as ``rtvm``).
"""
from . import _ffi, synth  # noqa: F401
from ._ffi import Context, RtvmError  # noqa: F401
from .types import KeyPoint, DMatch  # noqa: F401
from .mosaic import VideMosaic, crop_black_areas, scale_to_screen  # noqa: F401
from .navigation import canny_edges, navigation_obstacles, obstacle_boxes, occupancy_grid  # noqa: F401
