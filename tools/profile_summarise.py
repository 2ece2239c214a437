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
    for f in ("conv_igemm.hip", "conv_wr.hip", "conv_tile.h", "conv_igemm.h", "hdf_common.h"):
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
for leg in ("bench", "bench1s", "roofline"):
    f = one(f"{leg}/**/*_kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(dst, f"{pre}_{leg}_kernel_stats.csv"))


def step_window(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    return rows[adam[-2] + 1: adam[-1] + 1]


def per_kernel_us(win):
    tot = defaultdict(lambda: [0.0, 0])
    for r in win:
        k = short(r["Kernel_Name"])
        tot[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot[k][1] += 1
    return tot


# 1b. the per-kernel table of a step: clean (one-stream) time, time inside the shipped three-stream step, HBM bytes
#     and matrix-pipe cycles from the PMC passes, and what that is against the two roofs
f1, f3 = one("bench1s/**/*_kernel_trace.csv"), one("bench/**/*_kernel_trace.csv")
if f1 and f3:
    clean, over = per_kernel_us(step_window(f1)), per_kernel_us(step_window(f3))
    pm = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE", "MFMA", "LDS"):
        g = one(f"pmc_step_{c}/**/*counter_collection.csv")
        pm[c] = pmc_per_kernel(g) if g else {}
    steps_in_pmc = 2.0      # bench.py --steps 1 --warmup 1 (the roofline launches are taken out below)
    F32_KERNELS = ("tok_", "attn_", "tf_wgrad", "patch_embed")
    with open(os.path.join(dst, f"{pre}_step_kernel_table.txt"), "w") as fh:
        fh.write("Per-kernel table of one training step (4x128^3, batch 2, bf16).  us_1s: duration with both stream knobs set "
                 "(one stream, nothing shares the GPU);\nus_3s: inside the shipped step (caller's stream + weight-gradient side "
                 "stream + branch stream: durations of kernels that share the GPU are stretched).\nHBM MB per step = "
                 "(2 x FETCH_SIZE + WRITE_SIZE) from separate --pmc passes (gfx950 correction: FETCH_SIZE counts half of "
                 "wide reads); GB/s = MB / us_1s, vs the 8 TB/s HBM roof.\nMFMA TF = SQ_VALU_MFMA_BUSY_CYCLES x (1024 FLOP per "
                 "busy cycle for the bf16 32x32x16 kernels, 64 for the f32 MFMAs of the token / attention / patch kernels) / us_1s, "
                 "vs 2500 (bf16) or 157 (f32) TF: executed matrix FLOPs, padding included.\nLDS/MFMA = SQ_INSTS_LDS per matrix instruction "
                 "(SQ_VALU_MFMA_BUSY_CYCLES / 32 for the bf16 32x32x16 kernels, / 64 and / 32 for the f32 32x32x2 and 16x16x4 forms are "
                 "not told apart: shown for the bf16 kernels only); conf = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.\n\n")
        fh.write(f"{'kernel':60s} {'n':>4s} {'us_1s':>9s} {'us_3s':>9s} {'HBM MB':>9s} {'GB/s':>7s} {'of 8T':>6s} {'MFMA TF':>8s} {'of roof':>7s} {'LDS/MFMA':>8s} {'conf':>5s}\n")
        t1 = t3 = 0.0
        for k, (d, n) in sorted(clean.items(), key=lambda kv: -kv[1][0]):
            def ctr(c, name):
                e = pm[c].get(k, {}).get(name)
                return None if not e else e[1]
            fe, wr, mf = ctr("FETCH_SIZE", "FETCH_SIZE"), ctr("WRITE_SIZE", "WRITE_SIZE"), ctr("MFMA", "SQ_VALU_MFMA_BUSY_CYCLES")
            calls = pm["FETCH_SIZE"].get(k, {}).get("FETCH_SIZE", [0, 0])[0]
            scale = (n / calls) if calls else 0.0        # counters cover 2 steps (+ 120 roofline launches of one kernel)
            mb = (2 * fe + wr) * 1024 / 1e6 * scale if fe is not None and wr is not None else None
            mcalls = pm["MFMA"].get(k, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 0])[0]
            rate = 64.0 if any(t in k for t in F32_KERNELS) else 1024.0
            tf = (mf * (n / mcalls) * rate / (d * 1e-6) / 1e12) if mf and mcalls and d > 0 else None
            roof = 157.0 if rate == 64.0 else 2500.0
            gbs = mb / d * 1e3 / 1e3 if mb is not None and d > 0 else None      # MB / us = TB/s * 1e-... -> GB/s below
            gbs = mb * 1e6 / (d * 1e-6) / 1e9 if mb is not None and d > 0 else None
            o = over.get(k, [0.0, 0])[0]
            t1 += d
            t3 += o
            li, la, lc = ctr("LDS", "SQ_INSTS_LDS"), ctr("LDS", "SQ_LDS_IDX_ACTIVE"), ctr("LDS", "SQ_LDS_BANK_CONFLICT")
            lcalls = pm["LDS"].get(k, {}).get("SQ_INSTS_LDS", [0, 0])[0]
            lds_per = None
            if li and mf and lcalls and mcalls and rate == 1024.0:
                lds_per = (li / lcalls) / ((mf / mcalls) / 32.0)
            conf = (lc / la) if la else None
            fh.write(f"{k:60s} {n:4d} {d:9.1f} {o:9.1f} " + (f"{mb:9.1f} {gbs:7.0f} {gbs / 8000:6.2f} " if mb is not None else f"{'-':>9s} {'-':>7s} {'-':>6s} ") +
                     (f"{tf:8.1f} {tf / roof:7.3f} " if tf is not None and tf > 0.05 else f"{'-':>8s} {'-':>7s} ") +
                     (f"{lds_per:8.2f} " if lds_per is not None else f"{'-':>8s} ") + (f"{conf:5.2f}" if conf is not None else f"{'-':>5s}") + "\n")
        fh.write(f"\nsum of kernel durations: {t1:.1f} us on one stream, {t3:.1f} us summed over the three streams\n")
    print("wrote", f"{pre}_step_kernel_table.txt")
    # 1c. roofline.step of bench.py: algorithmic FLOPs of every matrix-core convolution kernel of a step / their summed
    #     one-stream time / the bf16 MFMA peak, and the same per kernel family (with the family's HBM bytes)
    MFMA_FAMILIES = ("conv_ws2_kernel", "conv_wgrad2_kernel", "conv_igemm_kernel", "conv_wr_kernel", "conv_wgrad_s2_kernel",
                     "conv_wgrad_kernel", "convt_fused_kernel", "convt_ws_kernel", "conv_gather_s2_kernel", "conv_first_kernel",
                     "wgrad_first_kernel")
    fam = defaultdict(lambda: {"us_1s": 0.0, "launches": 0, "hbm_bytes": 0.0})
    for k, (d, n) in clean.items():
        for f_ in MFMA_FAMILIES:
            if k.startswith(f_):
                fe = pm["FETCH_SIZE"].get(k, {}).get("FETCH_SIZE")
                wr_ = pm["WRITE_SIZE"].get(k, {}).get("WRITE_SIZE")
                fam[f_]["us_1s"] += d
                fam[f_]["launches"] += n
                if fe and wr_ and fe[0]:
                    fam[f_]["hbm_bytes"] += (2 * fe[1] + wr_[1]) * 1024 * (n / fe[0])
                break
    # SURVEY.md App. B, per sample forward: 14 BasicConv3d 892 G + 4 UpConv 54.4 G + 3 ConvTranspose3d 50.7 G; x3 for
    # forward + data gradient + weight gradient, minus the first layer's data gradient (14.5 G: it has none), x batch 2
    gflop = ((892.0 + 54.4 + 50.7) * 3 - 14.5) * 2
    us = sum(v["us_1s"] for v in fam.values())
    dom = max(fam.items(), key=lambda kv: kv[1]["us_1s"])[0] if fam else None
    rec = {"what": "all matrix-core convolution kernels of one training step (4x128^3, batch 2, bf16), one-stream kernel trace",
           "algorithmic_gflop": gflop, "mfma_kernels_us_one_stream": us, "achieved": gflop / us * 1e3 if us else None,
           "peak": 2500.0, "unit": "TFLOP/s", "frac": gflop / us * 1e3 / 2500.0 if us else None,
           "dominant_family": dom, "families": {k: v for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["us_1s"])},
           "conv_source_digest": conv_source_digest()}
    json.dump(rec, open(os.path.join(dst, f"{pre}_step_roofline.json"), "w"), indent=1)
    print("roofline.step", rec["frac"], "dominant", dom)

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
             if r["Counter_Name"] == ctr and ("conv_wr_kernel" in r["Kernel_Name"] or "conv_ws2_kernel" in r["Kernel_Name"])]
        return sum(v) / len(v), len(v)
    fk, nf = per_launch(fe, "FETCH_SIZE")
    wk, nw = per_launch(wr, "WRITE_SIZE")
    rec = {
        "kernel": "conv_wr_kernel<bf16_t, 128, 1, 2, 4, false> (block_1_1_right forward: 64->32 channels @128^3, batch 2)",
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
