"""Diagnostic: VRAM in use after repeated create / stream / destroy cycles of a full camera (look-ahead engines included)."""
import os, subprocess, sys, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvo_amd import StereoCamera, StereoOdometer
from openvo_amd.synth import Corridor
import bench


def used():
    out = subprocess.run(["rocm-smi", "--showmeminfo", "vram"], capture_output=True, text=True).stdout
    m = re.search(r"Used Memory \(B\): (\d+)", out)
    return int(m.group(1)) / 2**30 if m else -1.0


c = Corridor("C2")
frames = c.pairs(0, 24)
print("start %.2f GiB" % used())
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    cam = StereoCamera(c.K(), c.dist(), c.K(), c.dist(), c.rect_params(), c.sgbm_params(), (c.w, c.h), max_keypoints=500)
    odo = StereoOdometer(cam, **bench.ODO_KW)
    for s in cam.stage_pairs(frames):
        odo.update(s, None)
    peak = used()
    del odo
    cam._ctx.close()
    del cam
    print("cycle %d: in use while streaming %.2f GiB, after close %.2f GiB" % (rep, peak, used()))
