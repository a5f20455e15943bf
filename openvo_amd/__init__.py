"""openvo_amd -- MI355X-native stereo visual odometry behind openVO's Python API.

    from openvo_amd import StereoCamera, StereoOdometer, rot2RPY, drawPoseOnImage

mirrors `from openVO import ...` (reference __init__.py:2-5).  The hot path runs in
libvo355.so (hand-written HIP for gfx950, C ABI in include/vo355.h); there is no CPU fallback.
"""
import os as _os

# The look-ahead engines (6) and the ahead-of-time pose steps (3) run on their own HIP streams next to the
# main one; the runtime multiplexes streams onto 4 hardware queues by default, which would serialise
# whichever of them happen to share one.  Must be set before the
# HIP runtime initialises (a process that has already created a HIP context keeps its own setting).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

from .stereo_camera import StereoCamera
from .stereo_odometer import StereoOdometer
from .utils.rot2RPY import rot2RPY
from .utils.drawPoseOnImage import drawPoseOnImage

__all__ = ["StereoCamera", "StereoOdometer", "rot2RPY", "drawPoseOnImage"]
__version__ = "0.1.0"
