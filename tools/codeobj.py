"""Kernel resource records of the shipped library, read from its gfx950 code objects (no GPU needed).

libhdf_hip.so carries one clang offload bundle per translation unit in .hip_fatbin; each holds a gfx950 ELF whose
NT_AMDGPU_METADATA note lists, per kernel: registers, spills, scratch (private segment) and static LDS.
`kernels(so)` returns {demangled name: record}; `disassemble(so, name_substring)` the instruction mnemonics of one kernel.
CLI: python tools/codeobj.py [regex]  -> one line per kernel."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "h-denseformer_amd", "lib", "libhdf_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so=LIB):
    """the gfx950 ELF images inside the library, in file order"""
    data = open(so, "rb").read()
    out = []
    for m in re.finditer(MAGIC, data):
        p = m.start()
        nb = struct.unpack_from("<Q", data, p + 24)[0]
        q = p + 32
        for _ in range(nb):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                out.append(data[p + off:p + off + size])
    return out


def _demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return [n.replace("(anonymous namespace)::", "").replace("void ", "") for n in r.stdout.splitlines()]


def kernels(so=LIB):
    recs = {}
    for img in code_objects(so):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(img)
            f.flush()
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True).stdout
        cur = None
        items = []
        for line in txt.splitlines():
            m = re.match(r"\s+- \.agpr_count:\s+(\d+)", line)
            if m:
                cur = {"agpr_count": int(m.group(1))}
                items.append(cur)
                continue
            m = re.match(r"\s+\.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|"
                         r"group_segment_fixed_size|max_flat_workgroup_size):\s+(\S+)", line)
            if m and cur is not None:
                cur[m.group(1)] = m.group(2) if m.group(1) == "name" else int(m.group(2))
        names = _demangle([k["name"] for k in items])
        for k, n in zip(items, names):
            k["mangled"] = k["name"]
            k["name"] = re.sub(r"\(.*", "", n)
            k["regs"] = k["vgpr_count"]   # (unified file: .vgpr_count already includes the AGPRs)
            alloc = -(-max(k["regs"], 1) // 8) * 8
            k["waves_per_simd"] = min(8, 512 // alloc)
            recs[k["name"]] = k
    return recs


def disassemble(mangled, so=LIB):
    """instruction lines of one kernel (by mangled name)"""
    for img in code_objects(so):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(img)
            f.flush()
            r = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f"--disassemble-symbols={mangled}",
                                f.name], capture_output=True, text=True)
        lines = [l.strip() for l in r.stdout.splitlines() if re.match(r"\s+[a-z_0-9]+ ", l) or re.match(r"\s+[sv]_[a-z_0-9]+", l)]
        if lines:
            return lines
    return []


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    for n, k in sorted(kernels().items()):
        if pat and not re.search(pat, n):
            continue
        print("%-100s V%3d A%3d S%3d spillV%3d scr%4d lds%6d w/simd %d" % (
            n[:100], k["vgpr_count"], k["agpr_count"], k["sgpr_count"], k.get("vgpr_spill_count", 0),
            k.get("private_segment_fixed_size", 0), k.get("group_segment_fixed_size", 0), k["waves_per_simd"]))
