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
SOURCES = ["conv_igemm.hip", "conv_wr.hip", "conv_first.hip", "unet_ops.hip", "transformer.hip", "transformer_fused.hip", "transformer_chain.hip", "loss.hip", "plan.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# per-file additions.  conv_wr.hip: MFMA results in architectural VGPRs -- its AGPR half holds the 216 weight registers
# of a wave (left to its heuristic hipcc puts the accumulators there and spills weights to scratch); its tile phase is one
# fully unrolled block of 216 MFMAs + staging, above the default size limit of `#pragma unroll` with an input transform
# (and no SLP vectorisation there: packed f32 adds beside MFMAs cost more than the scalar pair they replace)
EXTRA = {"conv_wr.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form", "-mllvm", "-pragma-unroll-threshold=1000000"],
         # attention: the score MFMA results feed v_exp_f32 directly (VALU sources cannot be AGPRs)
         "transformer.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _digest():
    h = hashlib.sha256()
    for f in sorted(os.listdir(SRC)) + ["../../include/hdf.h"]:
        with open(os.path.join(SRC, f), "rb") as fh:
            h.update(f.encode() + fh.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA.items())).encode())
    return h.hexdigest()


def lib_path():
    return os.path.join(LIB, "libhdf_hip.so")


def build(force=False, verbose=True, defs=(), name="libhdf_hip"):
    """defs / name: an A/B variant (e.g. defs=["-DHDF_NO_CONV_WR"], name="libhdf_hip_nowr") next to the product library;
    load it with HDF_LIB_PATH.  Variants are always rebuilt and never stamped."""
    os.makedirs(LIB, exist_ok=True)
    variant = name != "libhdf_hip"
    objdir = os.path.join(OBJ, name) if variant else OBJ
    os.makedirs(objdir, exist_ok=True)
    out = os.path.join(LIB, name + ".so")
    stamp = os.path.join(LIB, "libhdf_hip.stamp")
    dig = _digest()
    if not variant and not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == dig:
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + list(defs) + EXTRA.get(src, []) + ["-c", os.path.join(SRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(cc, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    if not variant:
        with open(stamp, "w") as fh:
            fh.write(dig)
    if verbose:
        print("built", out)
    return out


if __name__ == "__main__":
    # python build.py [--force] [--name libhdf_hip_x -DFOO -DBAR ...]
    args = sys.argv[1:]
    nm = args[args.index("--name") + 1] if "--name" in args else "libhdf_hip"
    build(force="--force" in args, defs=[a for a in args if a.startswith("-D")], name=nm)
