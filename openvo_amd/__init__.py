"""openvo_amd -- MI355X-native stereo visual odometry behind openVO's Python API.

    from openvo_amd import StereoCamera, StereoOdometer, rot2RPY, drawPoseOnImage

mirrors `from openVO import ...` (reference __init__.py:2-5).  The hot path runs in
libvo355.so (hand-written HIP for gfx950, C ABI in include/vo355.h); there is no CPU fallback.
"""
from .stereo_camera import StereoCamera
from .stereo_odometer import StereoOdometer
from .utils.rot2RPY import rot2RPY
from .utils.drawPoseOnImage import drawPoseOnImage

__all__ = ["StereoCamera", "StereoOdometer", "rot2RPY", "drawPoseOnImage"]
__version__ = "0.1.0"
