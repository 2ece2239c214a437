"""Build libhdf_hip.so for gfx950 with hipcc (cross-compiles without a GPU).  In-tree output:
h-denseformer_amd/lib/libhdf_hip.so (git-ignored, ships to the GPU box with the snapshot)."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
OBJ = os.path.join(HERE, "build")
SOURCES = ["conv_igemm.hip", "unet_ops.hip", "transformer.hip", "transformer_fused.hip", "loss.hip", "plan.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _digest():
    h = hashlib.sha256()
    for f in sorted(os.listdir(SRC)) + ["../../include/hdf.h"]:
        with open(os.path.join(SRC, f), "rb") as fh:
            h.update(f.encode() + fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def lib_path():
    return os.path.join(LIB, "libhdf_hip.so")


def build(force=False, verbose=True):
    os.makedirs(LIB, exist_ok=True)
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(LIB, "libhdf_hip.stamp")
    dig = _digest()
    if not force and os.path.exists(lib_path()) and os.path.exists(stamp) and open(stamp).read() == dig:
        return lib_path()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

    def cc(src):
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(SRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
        return obj

    with ThreadPoolExecutor(max_workers=min(5, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(cc, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path()] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    with open(stamp, "w") as fh:
        fh.write(dig)
    if verbose:
        print("built", lib_path())
    return lib_path()


if __name__ == "__main__":
    build(force="--force" in sys.argv)
