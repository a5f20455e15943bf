"""rot2RPY(T): the two (roll, pitch, yaw) Euler solutions of a 4x4 pose, each a (2,1) array.

Display helper of the reference (utils/rot2RPY.py:3-38), restated; pinned by
tests/golden/g6_rot2rpy.npz.  Convention R = Rz(yaw) Ry(pitch) Rx(roll); gimbal branch when
|cos(pitch)| < 1e-4.
"""
import numpy as np


def rot2RPY(T):
    R = np.asarray(T)[0:3, 0:3]
    roll, pitch, yaw = np.zeros((2, 1)), np.zeros((2, 1)), np.zeros((2, 1))
    c = np.sqrt(R[0][0] ** 2 + R[1][0] ** 2)
    if -1e-4 < c < 1e-4:
        # gimbal lock: yaw is not observable, fold it into roll
        pitch[:] = -R[2][0] * (np.pi / 2)
        roll[:] = R[2][0] * np.arctan2(-R[0][1], R[1][1])
        return roll, pitch, yaw
    for k, ck in enumerate((c, -c)):
        pitch[k] = np.arctan2(-R[2][0], ck)
        cp = np.cos(pitch[k])
        roll[k] = np.arctan2(R[2][1] / cp, R[2][2] / cp)
        yaw[k] = np.arctan2(R[1][0] / cp, R[0][0] / cp)
    return roll, pitch, yaw
