"""The native multi-GPU entry of the C-ABI on a machine WITHOUT a GPU: the partition rule it shards by (against the Python rule the
distributed harness uses), argument checks that come before any device call, and the loud failure without a device."""
import ctypes as C

import numpy as np
import pytest

import jpezy_amd as J
from jpezy_amd import api, sharding


def test_shard_range_of_the_c_entry_is_the_rule_of_the_distributed_harness():
    for n in (0, 1, 2, 7, 8, 9, 63, 64, 65, 4096, 4097, 100003):
        for world in (1, 2, 3, 4, 7, 8, 16):
            spans = [J.shard_range(n, world, k) for k in range(world)]
            assert spans == [sharding.shard_range(n, world, k) for k in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[k][1] == spans[k + 1][0] for k in range(world - 1))          # contiguous, no gaps
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)  # earlier shards take the extra
    lo, n = C.c_long(5), C.c_long(5)
    api.load_library().jpezy_shard_range(10, 4, 9, C.byref(lo), C.byref(n))              # k out of range: an empty shard, no fault
    assert (lo.value, n.value) == (0, 0)


def test_argument_checks_come_before_any_device_call():
    lib = api.load_library()
    px = np.zeros(16 * 16, np.uint8)
    sizes = (C.c_longlong * 1)()
    out = api.MultiOut()
    devs = (C.c_int * 1)(0)

    def call(n_dev=1, W=16, H=16, n_frames=1, o=out, d=devs):
        return lib.jpezy_encode_batch_multi(d, n_dev, api._np_ptr(px), api._np_ptr(px), api._np_ptr(px), W, H, 0, n_frames, 0, b"x", C.byref(o) if o is not None else None)
    assert call(n_dev=0) == -1 and b"devices" in lib.jpezy_hip_last_error()
    assert call(n_dev=65) == -1
    assert call(o=None) == -1
    assert call(W=0) == -1 and call(H=70000) == -1 and call(n_frames=0) == -1
    assert call() == -1 and b"neither" in lib.jpezy_hip_last_error()                      # nothing asked for
    buf = np.zeros(4096, np.uint8)
    out.jpg = buf.ctypes.data
    assert call() == -1 and b"jpg_sizes" in lib.jpezy_hip_last_error()                    # .jpg without sizes / stride
    out.jpg_sizes, out.jpg_stride = sizes, 4096
    rc = call()
    if lib.jpezy_hip_device_count() <= 0:
        assert rc == -2 and b"no CPU fallback" in lib.jpezy_hip_last_error()              # no device: fails loudly, computes nothing
    else:
        assert rc == 0 and sizes[0] > 600


def test_device_index_out_of_range_is_refused():
    lib = api.load_library()
    if lib.jpezy_hip_device_count() <= 0:
        pytest.skip("needs a device to have a range")
    with pytest.raises(J.JpezyError):
        J.encode_batch_multi([lib.jpezy_hip_device_count()], *(np.zeros(256, np.uint8),) * 3, 16, 16, 1)


def test_handle_entry_points_check_their_arguments_before_any_device_call():
    lib = api.load_library()
    devs = (C.c_int * 2)(0, 0)
    assert not lib.jpezy_multi_create(None, 1, 16, 16, 0, 0) and b"devices" in lib.jpezy_hip_last_error()
    assert not lib.jpezy_multi_create(devs, 0, 16, 16, 0, 0)
    assert not lib.jpezy_multi_create(devs, 65, 16, 16, 0, 0)
    assert not lib.jpezy_multi_create(devs, 2, 0, 16, 0, 0) and b"width/height" in lib.jpezy_hip_last_error()
    assert not lib.jpezy_multi_create(devs, 2, 16, 65536, 0, 0)
    px = np.zeros(256, np.uint8)
    out = api.MultiOut()
    assert lib.jpezy_multi_encode(None, api._np_ptr(px), api._np_ptr(px), api._np_ptr(px), 1, b"x", C.byref(out)) == -1
    assert b"null handle" in lib.jpezy_hip_last_error()
    assert lib.jpezy_multi_last_stats(None, None, 0) == 0 and lib.jpezy_multi_chunk_frames(None) == 0
    lib.jpezy_multi_destroy(None)                                                         # a no-op, as free(NULL)
    if lib.jpezy_hip_device_count() <= 0:
        assert not lib.jpezy_multi_create(devs, 2, 16, 16, 0, 0) and b"no CPU fallback" in lib.jpezy_hip_last_error()
        with pytest.raises(J.JpezyError):
            J.MultiEncoder([0], 16, 16)
    else:
        h = lib.jpezy_multi_create(devs, 2, 16, 16, 0, 0)
        assert h
        assert lib.jpezy_multi_encode(h, None, api._np_ptr(px), api._np_ptr(px), 1, b"x", C.byref(out)) == -1
        assert lib.jpezy_multi_encode(h, api._np_ptr(px), api._np_ptr(px), api._np_ptr(px), 1, b"x", None) == -1
        assert lib.jpezy_multi_encode(h, api._np_ptr(px), api._np_ptr(px), api._np_ptr(px), 1, b"x", C.byref(out)) == -1   # nothing asked for
        assert b"neither" in lib.jpezy_hip_last_error()
        sizes = (C.c_longlong * 1)()
        buf = np.zeros(4096, np.uint8)
        out.jpg, out.jpg_sizes, out.jpg_stride = buf.ctypes.data, sizes, 4096
        assert lib.jpezy_multi_encode(h, api._np_ptr(px), api._np_ptr(px), api._np_ptr(px), 0, b"x", C.byref(out)) == -1
        assert lib.jpezy_multi_encode(h, api._np_ptr(px), api._np_ptr(px), api._np_ptr(px), 1, b"x", C.byref(out)) == 0 and sizes[0] > 600
        lib.jpezy_multi_destroy(h)
