"""The RCCL binding behind vo_mgpu_* on real hardware.  An 8-GPU node is not available to the tests, but a
one-rank communicator on the one GPU of the test box already goes through everything that can be wrong in the
binding itself: dlopen of librccl.so.1 and its symbols, ncclGetUniqueId, ncclCommInitRank, the collectives on the
library's own HIP stream with device staging buffers, error strings, destroy.  (Several ranks: tests/test_sharding.py
runs the same Group code over its socket transport; RCCL refuses two ranks on one device.)"""
import ctypes

import numpy as np
import pytest

from openvo_amd import _native, sharding

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_round_trip():
    L = _native.lib()
    n = ctypes.c_int(0)
    assert L.vo_device_count(ctypes.byref(n)) == 0 and n.value >= 1
    ident = (ctypes.c_uint8 * 128)()
    assert L.vo_mgpu_unique_id(ident) == 0, L.vo_mgpu_last_error(None)
    assert any(ident)                                                   # a real id, not zeros
    h = ctypes.c_void_p()
    rc = L.vo_mgpu_create(0, 0, 1, ident, ctypes.byref(h))
    assert rc == 0 and h.value, L.vo_mgpu_last_error(None)
    try:
        local = np.arange(5 * 17, dtype=np.float64) * 0.25 - 3.0
        out = np.full(5 * 17, np.nan)
        assert L.vo_mgpu_gather_poses(h, local.ctypes.data, 5, out.ctypes.data) == 0, L.vo_mgpu_last_error(h)
        assert np.array_equal(out, local)
        out2 = np.full(7, np.nan)
        assert L.vo_mgpu_all_gather_f64(h, local.ctypes.data, 7, out2.ctypes.data) == 0
        assert np.array_equal(out2, local[:7])
        v = np.array([1.5, -2.0, 1e300])
        assert L.vo_mgpu_all_reduce_max_f64(h, v.ctypes.data, 3) == 0
        assert np.array_equal(v, [1.5, -2.0, 1e300])
        # bad arguments are refused with a message, not a crash
        assert L.vo_mgpu_gather_poses(h, None, 5, out.ctypes.data) != 0
        assert L.vo_mgpu_last_error(h)
    finally:
        L.vo_mgpu_destroy(h)


def test_group_routes_through_rccl_when_attached():
    """sharding.Group with its RCCL handle attached by hand (world 1): all_gather_f64 / all_reduce_max / gather_relative
    take the vo_mgpu_* path, not the socket one."""
    g = sharding.Group(0, 1, "127.0.0.1")
    L = _native.lib()
    ident = (ctypes.c_uint8 * 128)()
    assert L.vo_mgpu_unique_id(ident) == 0
    h = ctypes.c_void_p()
    assert L.vo_mgpu_create(0, 0, 1, ident, ctypes.byref(h)) == 0
    g._mgpu, g._lib, g.transport = h, L, "rccl"
    try:
        T = np.tile(np.eye(4), (3, 1, 1))
        T[:, 2, 3] = [0.1, 0.2, 0.3]
        ok = np.array([1.0, 0.0, 1.0])
        allT, allok = g.gather_relative(T, ok)
        assert np.array_equal(allT.reshape(-1, 4, 4), T) and allok.tolist() == [True, False, True]
    finally:
        g.close()
