"""Import-name alias so that `from openVO import StereoCamera, StereoOdometer` (the reference's
import line, reference __init__.py:2-5) resolves to the MI355X implementation."""
from openvo_amd import StereoCamera, StereoOdometer, rot2RPY, drawPoseOnImage  # noqa: F401
