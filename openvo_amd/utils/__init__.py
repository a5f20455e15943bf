"""Helpers the reference re-exports at package level (src/openVO/__init__.py:4-5): pose angles and the optional overlay."""
from .rot2RPY import rot2RPY
from .drawPoseOnImage import drawPoseOnImage

__all__ = ["rot2RPY", "drawPoseOnImage"]
