"""SURVEY 5.2 / VERDICT r03 #7c: the host side of csrc/plan.hip -- 1,900 lines of offset arithmetic: parameter table,
workspace carving, named views, the 2-D embedding -- built with AddressSanitizer + UndefinedBehaviorSanitizer (CPU
only: device code is not instrumented, GPU sanitizers are not available on this pool) and walked for the five BASELINE
configs, three storage types and three batch sizes by tools/san_plan_walk.py in a child process that preloads the ASan
runtime.  Any report (heap overflow in a table, signed overflow in an offset, misaligned access) fails the test."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

RT = "/opt/rocm/lib/llvm/lib/clang"


def _asan_runtime():
    for root, _dirs, files in os.walk(RT):
        if "libclang_rt.asan-x86_64.so" in files:
            return os.path.join(root, "libclang_rt.asan-x86_64.so")
    return None


def test_plan_layout_code_under_asan_and_ubsan():
    rt = _asan_runtime()
    if rt is None or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no ROCm clang sanitizer runtime in this image")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "build_sanitized.sh")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    san = os.path.join(ROOT, "h-denseformer_amd", "lib", "libhdf_hip_san.so")
    syms = subprocess.run(["nm", "-D", san], capture_output=True, text=True).stdout
    assert "__asan_" in syms and "__ubsan_handle" in syms, "the build is not instrumented"
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", HDF_LIB_PATH=san,
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    w = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "san_plan_walk.py")], capture_output=True, text=True,
                       env=env, timeout=300)
    out = w.stdout + w.stderr
    assert w.returncode == 0, out[-4000:]
    assert "runtime error" not in out and "AddressSanitizer" not in out, out[-4000:]
    assert out.count("ws bytes") == 18            # 6 configurations x 3 storage types
