"""drawPoseOnImage(T, img): text overlay of the pose (reference utils/drawPoseOnImage.py:5-38).

Out of the hot path (marked out of scope): the reference rasterises the
text with cv2.putText (Hershey font).  When cv2 is importable the same four lines are drawn at
the same places; otherwise the image is left unchanged and pose_text(T) gives the lines.  Returns None like the reference
(it draws in place).
"""
import numpy as np

from .rot2RPY import rot2RPY


def pose_text(T):
    """The four overlay lines; angles from the Euler solution with the smaller norm, shown in the
    aircraft-like convention of the reference (roll <- camera yaw, pitch <- -pitch, yaw <- roll)."""
    roll, pitch, yaw = rot2RPY(T)
    norms = [np.linalg.norm([roll[k], pitch[k], yaw[k]]) for k in (0, 1)]
    k = 1 if norms[0] > norms[1] else 0
    T = np.asarray(T)
    xyz = [str(np.round(float(T[i, 3]), 1)) for i in range(3)]
    return ["Roll = " + str(np.round(yaw[k], 3)), "Pitch = " + str(np.round(-pitch[k], 3)),
            "Yaw = " + str(np.round(roll[k], 3)), "x,y,z = " + ", ".join(xyz)]


def drawPoseOnImage(T, img):
    try:
        import cv2
    except ImportError:
        return None          # nothing drawn; the reference returns None as well (it draws in place)
    h = img.shape[0]
    for line, dy, scale in zip(pose_text(T), (180, 120, 60, 10), (2.0, 2.0, 2.0, 1.6)):
        cv2.putText(img, text=line, org=(0, h - dy), fontFace=cv2.FONT_HERSHEY_SIMPLEX, fontScale=scale,
                    color=(0, 0, 255), thickness=3)
    return None              # in place, like the reference [utils/drawPoseOnImage.py:5-38]
