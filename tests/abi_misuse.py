"""Run by tests/test_gpu_pipeline.py::test_abi_misuse_returns_a_status_and_never_crashes in a process of its own: every
context-taking entry point of include/vo355.h with a NULL context, then with a fresh context (nothing configured: wrong call
order as well), all pointers NULL and hostile integers.  Prints `survived <calls> <names of calls that returned VO_OK>`."""
import ctypes, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvo_amd import _native

L = _native.lib()
vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
ctx = _native.Context(0, 320, 240, 64, 200)
skip = {"vo_create", "vo_destroy", "vo_last_error", "vo_mgpu_destroy", "vo_mgpu_last_error", "vo_device_count", "vo_mgpu_unique_id",
        "vo_mgpu_create", "vo_mgpu_gather_poses", "vo_mgpu_all_gather_f64", "vo_mgpu_all_reduce_max_f64", "vo_mgpu_info", "vo_rodrigues",
        "vo_ratio_filter", "vo_device_name", "vo_measure_copy"}
src = open(_native.__file__).read()
names = sorted(set(re.findall(r"L\.(vo_[a-z0-9_]+)\.argtypes", src)) - skip)
ints = [-1, 0, 1, 27, 28, 1000, 2**31 - 1, -2**31]
rng = np.random.default_rng(3)
calls, ok = 0, set()
for n in names:
    f = getattr(L, n)
    at = f.argtypes
    if not at or at[0] is not vp:
        continue
    for variant in range(6):
        args = []
        for k, t in enumerate(at):
            if k == 0:
                args.append(None if variant == 0 else ctx._h)
            elif t is ci or t is ctypes.c_int64:
                args.append(int(rng.choice(ints)) if variant > 1 else (0 if variant == 0 else -1))
            elif t is cd or t is ctypes.c_float:
                args.append(float(rng.choice([0.0, -1.0, 0.8, float("nan"), 1e30])))
            elif t is ctypes.c_uint32:
                args.append(int(rng.integers(0, 2**32)))
            else:
                args.append(None)
        calls += 1
        if f(*args) == 0:
            ok.add(n)
            assert variant != 0 or n in ("vo_enable_timing",), n          # nothing succeeds on a NULL context
ctx.close()
print("survived", calls, " ".join(sorted(ok)))
