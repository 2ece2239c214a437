"""Turns gpurun_out/prof_<round>/ (tools/profile_round.sh) into the summaries kept under profiles/.

usage: python tools/profile_summarise.py r02 [tag]"""
import csv
import glob
import hashlib
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
tag = sys.argv[2] if len(sys.argv) > 2 else ""
src = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}{tag}")
dst = os.path.join(ROOT, "profiles")
pre = f"{rnd}{tag}"


def one(pattern):
    g = glob.glob(os.path.join(src, pattern), recursive=True)
    return g[0] if g else None


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:72]


def conv_source_digest():
    h = hashlib.sha256()
    for f in ("conv_igemm.hip", "conv_igemm.h", "hdf_common.h"):
        with open(os.path.join(ROOT, "h-denseformer_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def pmc_per_kernel(path):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(path)):
        a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


# 1. kernel stats + one-step totals
for leg in ("bench", "roofline"):
    f = one(f"{leg}/**/*_kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(dst, f"{pre}_{leg}_kernel_stats.csv"))
f = one("bench/**/*_kernel_trace.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    win = rows[adam[-2] + 1: adam[-1] + 1]
    tot = defaultdict(lambda: [0.0, 0])
    for r in win:
        k = short(r["Kernel_Name"])
        tot[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot[k][1] += 1
    # union of the launch intervals (two streams run concurrently in backward: the per-kernel sums overlap)
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in win)
    union, cs, ce = 0, iv[0][0], iv[0][1]
    for a_, b_ in iv[1:]:
        if a_ > ce:
            union += ce - cs
            cs, ce = a_, b_
        else:
            ce = max(ce, b_)
    union += ce - cs
    queues = sorted({r.get("Queue_Id", "?") for r in win})
    with open(os.path.join(dst, f"{pre}_step_kernel_totals.txt"), "w") as fh:
        fh.write(f"one adam-to-adam window of the kernel trace: {union / 1e3:.1f} us with at least one kernel running, "
                 f"{sum(v[0] for v in tot.values()):.1f} us summed over the kernels ({len(queues)} queues: durations of "
                 f"kernels that share the GPU are stretched), {len(win)} launches\n")
        for k, (d, n) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
            fh.write(f"{d:9.1f} us {n:4d}  {k}\n")
for name in ("bench_line_full.json", "roofline_line.json", "clock_probe_64x32.txt", "clock_probe_32x32.txt"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{pre}_{name}"))

# 2. traffic of the dominant kernel
fe, wr = one("pmc_conv_FETCH_SIZE/**/*counter_collection.csv"), one("pmc_conv_WRITE_SIZE/**/*counter_collection.csv")
if fe and wr:
    def per_launch(path, ctr):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
             if r["Counter_Name"] == ctr and "conv_ws2_kernel" in r["Kernel_Name"]]
        return sum(v) / len(v), len(v)
    fk, nf = per_launch(fe, "FETCH_SIZE")
    wk, nw = per_launch(wr, "WRITE_SIZE")
    rec = {
        "kernel": "conv_ws2_kernel<bf16_t, 32, 128, false, 1> (block_1_1_right forward: 64->32 channels @128^3, batch 2)",
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --output-format "
                   "csv -- python3 tools/conv_micro.py --cin 64 --cout 32 --xf 0 --reps 2",
        "launches_averaged": [nf, nw], "FETCH_SIZE_KB_raw": fk, "WRITE_SIZE_KB": wk,
        "correction": "gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM): "
                      "fetch bytes = 2 * FETCH_SIZE * 1024",
        "traffic_bytes_per_launch": int(2 * fk * 1024 + wk * 1024),
        "algorithmic_bytes_per_launch": 2 * 128 ** 3 * (64 + 32) * 2,
        "conv_source_digest": conv_source_digest(),
    }
    json.dump(rec, open(os.path.join(dst, f"{pre}_conv_traffic.json"), "w"), indent=1)
    print("traffic", rec["traffic_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"], "x algorithmic")

# 3. MFMA utilisation
lines = ["MFMA utilisation of the conv kernel families (tools/conv_micro.py, bf16, batch 2; one rocprofv3 --pmc pass per shape:",
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE).  util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024",
         "SIMDs): the share of SIMD-cycles with the matrix pipe busy (both counters are sums over the 8 XCDs; a 32x32x16 bf16",
         "MFMA holds the pipe 32 cycles, so util x 2.5 PF x (clock / 2.4 GHz) is the delivered rate).", ""]
for d in sorted(glob.glob(os.path.join(src, "pmc_mfma_*"))):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    acc = pmc_per_kernel(f[0])
    for k, ctrs in acc.items():
        if "conv" not in k or "pack" in k:
            continue
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in ctrs or "GRBM_GUI_ACTIVE" not in ctrs:
            continue
        n = ctrs["GRBM_GUI_ACTIVE"][0]
        busy = ctrs["SQ_VALU_MFMA_BUSY_CYCLES"][1] / n
        gui = ctrs["GRBM_GUI_ACTIVE"][1] / n
        util = busy / (gui / 8 * 1024) if gui else 0.0
        lines.append(f"{os.path.basename(d)[9:]:22s} {k:58s} launches {n:3d}  MFMA_BUSY {busy:14.0f}  GUI_ACTIVE {gui:12.0f}  "
                     f"cycles/XCD {gui / 8:10.0f}  util {util:.3f}")
open(os.path.join(dst, f"{pre}_mfma_utilisation.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[5:]))

# 4. per-kernel traffic of a step
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = one(f"pmc_step_{c}/**/*counter_collection.csv")
    if not f:
        continue
    acc = pmc_per_kernel(f)
    with open(os.path.join(dst, f"{pre}_pmc_{c}_per_kernel.txt"), "w") as fh:
        fh.write(f"counter {c} (KB; FETCH_SIZE to be doubled per the gfx950 correction), bench.py --steps 1 --warmup 1: 2 steps + roofline leg\n")
        for k, ctrs in sorted(acc.items(), key=lambda kv: -kv[1][c][1]):
            n, v = ctrs[c]
            fh.write(f"{v:16.0f} total {v / n:14.0f} mean {n:5d} calls  {k}\n")
