"""Walks the host-side layout code of csrc/plan.hip -- plan creation, parameter table, workspace carving, named buffer
views, the 2-D embedding -- for the five BASELINE configs through raw ctypes (no torch, no GPU).  Meant to run against
the AddressSanitizer + UBSan build of the library (tests/test_cpu_sanitized_host.py); prints one line per plan."""
import ctypes as C
import os
import sys

lib = C.CDLL(os.environ["HDF_LIB_PATH"])
lib.hdf_last_error.restype = C.c_char_p
lib.hdf_plan_param_floats.restype = C.c_int64
lib.hdf_plan_num_params.restype = C.c_int64
lib.hdf_plan_workspace_bytes.restype = C.c_int64
lib.hdf_plan_inference_workspace_bytes.restype = C.c_int64
lib.hdf_plan_param_info.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int), C.POINTER(C.c_int64)]
lib.hdf_plan_buffer_info.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.hdf_plan_workspace_bytes.argtypes = [C.c_void_p, C.c_int]
lib.hdf_plan_inference_workspace_bytes.argtypes = [C.c_void_p, C.c_int]
lib.hdf_plan_num_params.argtypes = [C.c_void_p]
lib.hdf_plan_param_floats.argtypes = [C.c_void_p]
lib.hdf_plan_destroy.argtypes = [C.c_void_p]

# BASELINE.json configs: [0] 2-D 4-ch 256^2, [1]/[2] nf 32 4x128^3, [3] 2-modal 144^3 3-class, [4] nf 48 4x160^3
CONFIGS = [("2d", 4, 2, 32, (256, 256), 24), ("3d", 4, 4, 32, (128, 128, 128), 24), ("3d", 2, 3, 32, (144, 144, 144), 24),
           ("3d", 4, 4, 48, (160, 160, 160), 24), ("3d", 2, 3, 16, (32, 32, 32), 8), ("2d", 2, 3, 16, (384, 384), 16)]
BUFFERS = ["xin", "ds0", "ds1", "ds2", "x4", "attnout", "at1", "at2", "at3", "attnall", "cat0", "cat1", "cat2"]
rc = 0
for kind, cin, ncls, nf, size, depth in CONFIGS:
    for dtype in (0, 1, 2):
        h = C.c_void_p()
        if kind == "2d":
            r = lib.hdf_plan_create_2d(cin, ncls, nf, size[0], size[1], depth, dtype, C.byref(h))
        else:
            r = lib.hdf_plan_create(cin, ncls, nf, size[0], size[1], size[2], depth, dtype, C.byref(h))
        if r != 0:
            print("create failed", kind, cin, ncls, nf, size, depth, dtype, lib.hdf_last_error())
            rc = 1
            continue
        n = lib.hdf_plan_num_params(h)
        name = C.create_string_buffer(256)
        off, numel, ndim = C.c_int64(), C.c_int64(), C.c_int()
        shape = (C.c_int64 * 5)()
        end = 0
        for i in range(n):
            assert lib.hdf_plan_param_info(h, i, name, 256, C.byref(off), C.byref(numel), C.byref(ndim), shape) == 0
            assert off.value >= end and off.value % 16 == 0, (name.value, off.value, end)
            end = off.value + numel.value
        assert end <= lib.hdf_plan_param_floats(h)
        # out-of-range queries must be refused, not read past the table
        assert lib.hdf_plan_param_info(h, n, name, 256, C.byref(off), C.byref(numel), C.byref(ndim), shape) != 0
        assert lib.hdf_plan_param_info(h, -1, name, 256, C.byref(off), C.byref(numel), C.byref(ndim), shape) != 0
        tiny = C.create_string_buffer(4)    # a name buffer that is too small must not be overrun
        lib.hdf_plan_param_info(h, 0, tiny, 4, C.byref(off), C.byref(numel), C.byref(ndim), shape)
        sizes = []
        for batch in (1, 2, 3):
            wb, ib = lib.hdf_plan_workspace_bytes(h, batch), lib.hdf_plan_inference_workspace_bytes(h, batch)
            assert 0 < ib <= wb, (batch, ib, wb)
            sizes.append(wb)
            found = 0
            for b in BUFFERS:
                boff, pitch = C.c_int64(), C.c_int64()
                c, d, hh, w = C.c_int(), C.c_int(), C.c_int(), C.c_int()
                if lib.hdf_plan_buffer_info(h, batch, b.encode(), C.byref(boff), C.byref(pitch), C.byref(c), C.byref(d),
                                            C.byref(hh), C.byref(w)) == 0:
                    found += 1
                    esz = 4 if dtype == 0 else 2
                    last = boff.value + ((batch * d.value * hh.value * w.value - 1) * pitch.value + c.value) * esz
                    assert 0 <= boff.value and last <= wb, (b, boff.value, last, wb)
            assert found >= 4
            assert lib.hdf_plan_buffer_info(h, batch, b"no_such_buffer", C.byref(boff), C.byref(pitch), C.byref(c), C.byref(d),
                                            C.byref(hh), C.byref(w)) != 0
        print(kind, cin, ncls, nf, size, depth, "dtype", dtype, "params", n, "floats", lib.hdf_plan_param_floats(h),
              "ws bytes b=1,2,3", sizes)
        lib.hdf_plan_destroy(h)
# refused configurations: error paths of the argument checks
h = C.c_void_p()
for bad in [(0, 4, 32, 128, 128, 128, 24, 1), (4, 9, 32, 128, 128, 128, 24, 1), (4, 4, 40, 128, 128, 128, 24, 1),
            (4, 4, 32, 100, 128, 128, 24, 1), (4, 4, 32, 128, 128, 128, 24, 7)]:
    assert lib.hdf_plan_create(*bad, C.byref(h)) != 0, bad
sys.exit(rc)
