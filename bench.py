"""Benchmark of the H-DenseFormer 3D training hot path on MI355X.

Metric (BASELINE.json): train samples/sec on 4x128^3 volumes.  One step = forward (4 deep-supervision
outputs) + DeepSuperloss(CEPlusDice) + backward + gradient all-reduce (N>1) + Adam step -- the body of
the reference's inner loop, trainer.py:369-380 -- on synthetic BraTS-shape inputs already resident in HBM.
Workload at every N: HDenseFormer_32(in=4, n_cls=4, 128^3, transformer_depth=24), per-GPU batch 2,
bf16 storage / fp32 accumulate (BASELINE configs[1]; configs[2] is the same at N=8).  Weak scaling.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (plus human-readable notes on stderr)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# the host driver only supports dmabuf IPC: RCCL / cross-process device memory need this before HIP initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# Data-parallel runs (launched through torch.distributed.run): two hardware queues per process.  The step uses four HIP
# streams (caller's, weight gradients, transformer branch, communication); with the runtime's default of four hardware queues
# the communication stream's queue assignment decides whether a collective that holds 16-32 compute units costs the step
# +0.2 or +1.9 ms (bimodal from run to run), with two it is +0.2 (16 workgroups) / +0.9 ms (32), stable, and the step
# without a collective is unchanged (tools/stage_cost.py, profiles/r06_stage_cost_hw_queues_*.json).
if "RANK" in os.environ and "MASTER_ADDR" in os.environ:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

import torch  # noqa: E402

FWD_BWD_GFLOP_PER_SAMPLE = 2998.4      # FlopCounterMode, 4x128^3 nf32 td24
BF16_MFMA_PEAK_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense bf16 MFMA peak
CFG = dict(in_channels=4, n_cls=4, n_filters=32, image_size=(128, 128, 128), transformer_depth=24)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# torch/oneDNN on all 256 host threads of the GPU box is pathologically slow (811 s/step, round 1), so the baseline pins a
# thread count: the best of the sweep 8 / 16 / 32 / 64 on that host (tools/cpu_thread_sweep.sh, profiles/r05_cpu_thread_sweep.txt).
# HDF_BENCH_CPU_THREADS overrides it (the sweep's knob; bench.py is not part of the library).
CPU_THREADS = int(os.environ.get("HDF_BENCH_CPU_THREADS", "16"))
CPU_BUDGET_S = 240


def _cpu_baseline_child():
    """Runs in a child process (no GPU): the oracle's train step -- the functional torch-CPU restatement of
    the reference path, kind 'port' -- on a BOUNDED sample.  First batch 1 of a 4x64^3 crop (1/8 of the voxels of
    the 4x128^3 workload; >99 % of the step's FLOPs are convolutions, linear in voxels), 1 warm-up + 2 timed steps; if
    that predicts a full-size step under 12 s, the real 4x128^3 step at batch 1 and at batch 2 (the benchmarked
    batch), 1 warm-up + 3 timed steps each.  Every stage prints a JSON line; the last one wins."""
    from oracle import hdf_oracle as orc
    nt = min(os.cpu_count() or 1, CPU_THREADS)
    torch.set_num_threads(nt)

    def time_steps(size, batch, n_timed):
        cfg = (CFG["in_channels"], CFG["n_cls"], CFG["n_filters"], (size,) * 3, CFG["transformer_depth"])
        tr = orc.OracleTrainer(orc.det_model(*cfg))
        x = torch.rand(batch, 4, size, size, size)
        lab = torch.randint(0, 4, (batch, size, size, size))
        onehot = torch.nn.functional.one_hot(lab, 4).permute(0, 4, 1, 2, 3).float()
        tr.step(x, onehot, drop_seed=1)
        ts = []
        for i in range(n_timed):
            t0 = time.time()
            tr.step(x, onehot, drop_seed=2 + i)
            ts.append(time.time() - t0)
        return sorted(ts)[len(ts) // 2]

    what = "oracle train step (fwd + DeepSuper CE+Dice + bwd + Adam), fp32"
    t64 = time_steps(64, 1, 2)
    rec = {"value": 1.0 / (8.0 * t64), "unit": "samples/s", "cores": nt, "host_logical_cpus": os.cpu_count(), "kind": "port",
           "sample": f"{what}, batch 1 of a 4x64^3 crop: {t64:.2f} s/step on {nt} threads, scaled x8 voxels to 4x128^3"}
    print(json.dumps(rec), flush=True)
    if 8.0 * t64 < 12.0:
        t1 = time_steps(128, 1, 3)
        rec["value"] = 1.0 / t1
        rec["sample"] = (f"{what}, 4x128^3: batch 1 {t1:.2f} s/step (median of 3 after 1 warm-up) on {nt} threads; "
                         f"4x64^3 crop gave {t64:.2f} s")
        rec["batch1_s_per_step"] = t1
        print(json.dumps(rec), flush=True)
        t2 = time_steps(128, 2, 3)
        rec["value"] = max(1.0 / t1, 2.0 / t2)
        rec["batch2_s_per_step"] = t2
        rec["sample"] = (f"{what}, 4x128^3 on {nt} threads, median of 3 timed steps after 1 warm-up: batch 1 "
                         f"{t1:.2f} s/step ({1.0 / t1:.3f} samples/s), batch 2 {t2:.2f} s/step ({2.0 / t2:.3f} samples/s); "
                         f"value = the better of the two")
        print(json.dumps(rec), flush=True)


def cpu_baseline():
    """Bounded CPU baseline in a subprocess with a hard wall-clock budget; the last JSON line it printed wins."""
    import subprocess
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(CPU_THREADS))
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child"], env=env,
                           capture_output=True, text=True, timeout=CPU_BUDGET_S)
        out = r.stdout
    except subprocess.TimeoutExpired as e:
        out = (e.stdout or b"").decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if not lines:
        return {"value": None, "unit": "samples/s", "cores": CPU_THREADS, "host_logical_cpus": os.cpu_count(), "kind": "port",
                "sample": f"oracle train step did not finish a 4x64^3 crop within {CPU_BUDGET_S} s"}
    return json.loads(lines[-1])


def conv_source_digest():
    """sha256 over the sources the dominant conv kernel is compiled from (what a PMC traffic profile is valid for)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("conv_igemm.hip", "conv_wr.hip", "conv_tile.h", "conv_igemm.h", "hdf_common.h"):
        with open(os.path.join(ROOT, "h-denseformer_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


# The stride-1 convolution launches of one training step that run conv_ws2_kernel (csrc/conv_igemm.hip): forward and data
# gradient of every 3x3x3 layer at 128^3 / 64^3 with <= 64 input channels, minus the first layer (conv_first_kernel) and
# block_1_1_right's forward (conv_wr_kernel).  (cin, cout, size, form): reference layers models/HDenseFormer.py:196-221,
# 190-194.  `form` selects the INSTANTIATION the step runs (round 6, VERDICT r05 #2: the round-5 replay ran every shape
# through the plain form and reported 0.39 where the step's own launches ran at 0.35):
#   "xf"    input transform on the way into LDS (the producer's InstanceNorm + ReLU): conv_ws2_kernel<.., XF=true, ..>
#   "bs"    data gradient with the next InstanceNorm backward's first pass in its epilogue: <.., BS=true> (hdf_op_conv3d_bwd_stats)
#   "split" data gradient of a decoder concat: two dense output halves (hdf_op_conv3d_split)
#   "acc"   data gradient that adds into the skip gradient (the UpConv chain)
#   ""      the plain form
WS2_FAMILY = [
    (32, 32, 128, "xf"), (32, 32, 128, "xf"),                            # block_1_2_left, block_1_2_right forward
    (32, 64, 64, ""), (64, 64, 64, "xf"), (64, 64, 64, "xf"), (64, 32, 64, ""),   # block_2_1_left, block_2_2_left/right, up3 forward
    (32, 32, 128, "bs"), (32, 32, 128, "bs"), (32, 64, 128, "split"),    # data gradients at 128^3 (1_2_right, 1_2_left, 1_1_right)
    (64, 32, 64, ""), (64, 64, 64, ""), (64, 64, 64, ""), (64, 128, 64, "split"), (32, 64, 64, "acc"),   # data gradients at 64^3
]
ROOFLINE_SCHEMA = 3   # 1: one conv launch (rounds 1-4); 2: the family through the plain form (round 5); 3: the family through
                      # the step's own instantiations (this file)


def roofline_conv_family(dev, passes=6):
    """Live HIP-event timing of the kernel family with the largest share of the step (profiles/*_step_kernel_table.txt):
    conv_ws2_kernel, 14 launches per step.  One pass = those 14 launches back to back, each through the operator entry
    that reaches the instantiation the step runs for it (WS2_FAMILY); achieved = their algorithmic FLOPs
    (2*27*Cin*Cout*voxels*batch) / the time of a pass."""
    from hdf_rt._lib import BF16, check, lib, ptr
    n = 2
    st = torch.cuda.current_stream().cuda_stream
    bufs = {}
    for cin, cout, s, form in set(WS2_FAMILY):
        x = torch.randn(n, s, s, s, cin, device=dev).to(torch.bfloat16)
        w = (torch.randn(27 * ((cout + 31) // 32 * 32) * cin, device=dev) * 0.02).to(torch.bfloat16)
        out = torch.zeros(n, s, s, s, cout, device=dev, dtype=torch.bfloat16)
        tiles = lib().hdf_op_conv3d_stat_tiles(BF16, cin, s, s, s)
        part = torch.empty(max(n * tiles, 1024) * ((cout + 31) // 32 * 32) * 2, device=dev)
        extra = None
        if form == "xf":
            extra = (torch.rand(n * cin, device=dev) + 0.5, torch.randn(n * cin, device=dev) * 0.1)
        elif form == "bs":
            extra = (torch.randn(n, s, s, s, cout, device=dev).to(torch.bfloat16), torch.rand(n * cout, device=dev) + 0.5,
                     torch.randn(n * cout, device=dev) * 0.1, torch.randn(n * cout, device=dev) * 0.1,
                     torch.rand(n * cout, device=dev) + 0.5)
        bufs[(cin, cout, s, form)] = (x, w, out, part, extra)

    def one_pass():
        for key in WS2_FAMILY:
            cin, cout, s, form = key
            x, w, out, part, extra = bufs[key]
            if form == "xf":
                check(lib().hdf_op_conv3d(BF16, 0, ptr(x), cin, cin, n, s, s, s, ptr(w), None, ptr(extra[0]), ptr(extra[1]), 1,
                                          ptr(out), cout, cout, ptr(part), 0, st), "conv xf")
            elif form == "bs":
                y, sc, sh, mu, rs = extra
                check(lib().hdf_op_conv3d_bwd_stats(BF16, ptr(x), cin, cin, n, s, s, s, ptr(w), ptr(out), cout, cout, ptr(y),
                                                    cout, ptr(sc), ptr(sh), ptr(mu), ptr(rs), ptr(part), st), "conv bs")
            elif form == "split":
                half = cout // 2
                o2 = out.view(-1)[out.numel() // 2:]
                check(lib().hdf_op_conv3d_split(BF16, ptr(x), cin, cin, n, s, s, s, ptr(w), ptr(out), ptr(o2), half, cout,
                                                half, ptr(part), None, 0, st), "conv split")
            else:
                check(lib().hdf_op_conv3d(BF16, 0, ptr(x), cin, cin, n, s, s, s, ptr(w), None, None, None, 0, ptr(out), cout,
                                          cout, ptr(part), 1 if form == "acc" else 0, st), "conv")
    for _ in range(3):
        one_pass()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(passes):
        one_pass()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / passes
    flops = sum(2.0 * 27 * cin * cout * (s ** 3) * n for cin, cout, s, _f in WS2_FAMILY)
    byt = sum(2.0 * n * s ** 3 * (cin + cout) for cin, cout, s, _f in WS2_FAMILY)
    achieved = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": achieved, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / BF16_MFMA_PEAK_TFLOPS, "traffic": None, "schema": ROOFLINE_SCHEMA,
            "frac_definition": "algorithmic FLOPs of the 14 conv_ws2_kernel launches of a step / the HIP-event time of one "
                               "back-to-back pass over them, each launch in the instantiation the step runs (input "
                               "transform, statistics epilogue, split / accumulating outputs) / 2.5 PF; rounds 1-4 reported "
                               "the single conv_wr launch (now roofline.wr), round 5 the plain-form replay",
            "kernel": "conv_ws2_kernel family: the 14 stride-1 3x3x3 conv launches (forward + data gradient, 128^3 / 64^3, "
                      "batch 2) of one training step -- the family with the largest share of the step's kernel time",
            "launches_per_pass": len(WS2_FAMILY), "pass_ms": ms, "avg_launch_ms": ms / len(WS2_FAMILY),
            "flops_per_pass": flops, "algorithmic_bytes_per_pass": byt}


def step_roofline_from_profile():
    """roofline.step: algorithmic FLOPs of all matrix-core conv kernels of a step / their summed one-stream time / peak, from
    the newest committed profile set (tools/profile_summarise.py writes profiles/*_step_roofline.json); not measurable
    live (it needs a one-stream kernel trace).  Emitted only when that profile was taken on THIS build's conv kernels
    (conv_source_digest); otherwise null plus the reason (a stale fraction next to live numbers)."""
    best = None
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if name.endswith("_step_roofline.json"):
            best = name
            break
    if best is None:
        return None, "no profiles/*_step_roofline.json"
    rec = json.load(open(os.path.join(ROOT, "profiles", best)))
    if rec.get("conv_source_digest") != conv_source_digest():
        return None, ("profiles/%s was taken on other conv kernel sources (digest %s, this build %s): not reported"
                      % (best, str(rec.get("conv_source_digest"))[:12], conv_source_digest()[:12]))
    rec["source"] = "profiles/" + best
    rec["from_committed_profile"] = True
    return rec, None


def roofline_dominant_kernel(dev):
    """Live HIP-event timing of the dominant kernel class of the step: the bf16 implicit-GEMM conv on its largest layer,
    block_1_1_right: 64->32 channels at 128^3, batch 2 (round 4: conv_wr_kernel, csrc/conv_wr.hip -- the launch goes
    through hdf_op_conv3d, i.e. through the plan's own routing rule).
    Algorithmic FLOPs per launch = 2*27*Cin*Cout*voxels*batch."""
    from hdf_rt._lib import BF16, check, lib, ptr
    n, cin, cout, s = 2, 64, 32, 128
    x = torch.randn(n, s, s, s, cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(27 * 32 * cin, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.empty(n, s, s, s, cout, device=dev, dtype=torch.bfloat16)
    tiles = lib().hdf_op_conv3d_stat_tiles(BF16, cin, s, s, s)
    part = torch.empty(n * tiles * 32 * 2, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def launch():
        check(lib().hdf_op_conv3d(BF16, 0, ptr(x), cin, cin, n, s, s, s, ptr(w), None, None, None, 0, ptr(out), cout,
                                  cout, ptr(part), 0, st), "conv")
    for _ in range(20):      # the package sits at its 1400 W cap under this kernel: let clock and power settle
        launch()
    reps = 100
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * 27 * cin * cout * (s ** 3) * n
    achieved = flops / (ms * 1e-3) / 1e12
    # HBM bytes per launch of exactly this kernel/shape from the committed PMC passes (FETCH_SIZE x2 correction +
    # WRITE_SIZE, collected in separate rocprofv3 --pmc runs by tools/conv_traffic.py); not measurable live.  The
    # profile records the digest of the conv kernel sources it was taken on: a profile of another build is refused.
    traffic, traffic_src = None, None
    digest = conv_source_digest()
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if name.endswith("_conv_traffic.json"):
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))
            if rec.get("conv_source_digest") == digest:
                traffic, traffic_src = rec.get("traffic_bytes_per_launch"), "profiles/" + name
                break
    if traffic is None:
        log("roofline.traffic: no profiles/*_conv_traffic.json taken on this build's conv kernels "
            f"(digest {digest[:12]}); reporting null")
    return {"bound": "mfma", "achieved": achieved, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / BF16_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
            "kernel": "conv_wr_kernel<bf16_t,128,1,2,4,false> (block_1_1_right fwd, 64->32 @128^3, batch 2)",
            "avg_launch_ms": ms, "flops_per_launch": flops, "algorithmic_bytes_per_launch": 2.0 * n * s ** 3 * (cin + cout)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2, help="per-GPU batch")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true",
                    help="skip the 120 launches of the roofline leg (per-kernel counter passes of tools/profile_round.sh)")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only the roofline leg (the dominant kernel's timed launches): the command profiled for "
                         "profiles/*_roofline_kernel_stats.csv, whose per-kernel average must match avg_launch_ms")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.cpu_baseline_child:
        _cpu_baseline_child()
        return

    if a.roofline_only:
        torch.cuda.set_device(0)
        from hdf_rt import _lib
        _lib.lib()
        dev0 = torch.device("cuda", 0)
        r = roofline_conv_family(dev0)
        r["wr"] = roofline_dominant_kernel(dev0)
        r["step"], why = step_roofline_from_profile()
        if why:
            r["step_unavailable"] = why
        print(json.dumps({"roofline": r}), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world == 1 and a.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    # HDF_BENCH_ONE_DEVICE=1 (diagnostic): every rank on cuda:0 over gloo, to exercise the multi-rank path on a
    # single-GPU box; the measured run is one rank per GPU over RCCL ("nccl")
    one_dev = os.environ.get("HDF_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local = 0
        # two processes on one device cannot both hold every compute unit: the persistent transformer kernels of the two
        # ranks can each be left with half a grid resident and give up at a barrier (include/hdf.h: HDF_ERR_CHAIN_TIMEOUT;
        # the in-process serialisation does not reach across processes) -- this diagnostic mode runs the launch chain
        os.environ["HDF_NO_TF_CHAIN"] = "1"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # Launched through torch.distributed.run (RANK / MASTER_ADDR in the environment) the process group, GradSync and
    # the fences are ALWAYS set up, also for one rank: `--nproc-per-node 1` is the pre-flight of the RCCL path that a
    # single-GPU box allows (RCCL refuses two ranks on one device).  A plain `python bench.py` stays collective-free.
    dist_on = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if dist_on:
        import torch.distributed as dist
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from hdf_rt import _lib
    _lib.lib()                                       # no fallback: fail here if the HIP extension is missing
    from hdf_rt.optim import FlatAdam
    from loss.combine_loss import CEPlusDice, DeepSuperloss
    from models.HDenseFormer import HDenseFormer

    torch.manual_seed(0)
    net = HDenseFormer(CFG["in_channels"], CFG["n_cls"], CFG["n_filters"], image_size=CFG["image_size"],
                       transformer_depth=CFG["transformer_depth"]).to(dev)
    net.train()
    net.compute_dtype = a.dtype
    crit = DeepSuperloss(criterion=CEPlusDice(weight=None, ignore_index=0))
    opt = FlatAdam(net, lr=1e-3, weight_decay=1e-4)
    sync = None
    if dist_on:
        from hdf_rt.parallel import GradSync
        sync = GradSync(net)
        net.grad_hook = sync

    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x = torch.rand(a.batch, 4, 128, 128, 128, generator=g).to(dev)
    lab = torch.randint(0, 4, (a.batch, 128, 128, 128), generator=g)
    target = torch.nn.functional.one_hot(lab, 4).permute(0, 4, 1, 2, 3).float().contiguous().to(dev)

    def step():
        opt.zero_grad()
        outs = net(x)
        loss = crit(outs, target)
        loss.backward()
        if sync is not None:
            sync.wait()
        opt.step()
        return loss

    def fence():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        loss = step()
    fence()
    # wall clock around EXACTLY K steps between two barrier + synchronize fences (the contract's `value`), plus a HIP
    # event after every step on the stream the kernels run on: their median is the per-step device time
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    # the dominant conv launch INSIDE the timed steps: the library records one event pair per step around it
    # (hdf_plan_set_probe); read back after the closing fence
    probes, plan_h = [], None
    if world == 1 and not a.no_roofline:
        from hdf_rt._lib import check as _check, lib as _lib
        plan_h = net._last_rt.plan.h
        for _ in range(a.steps):
            pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            pair[0].record(), pair[1].record()          # torch creates the HIP event at the first record
            probes.append(pair)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs[0].record()
    for k in range(a.steps):
        if probes:
            _check(_lib().hdf_plan_set_probe(plan_h, probes[k][0].cuda_event, probes[k][1].cuda_event), "set_probe")
        loss = step()
        evs[k + 1].record()
    t_enqueued = time.perf_counter() - t0          # host side only: all K steps issued (the GPU is still working)
    fence()
    dt = time.perf_counter() - t0
    in_step_ms = None
    if probes:
        _check(_lib().hdf_plan_set_probe(plan_h, None, None), "set_probe")
        in_step_ms = sorted(p0.elapsed_time(p1) for p0, p1 in probes)
    raw_steps = [evs[k].elapsed_time(evs[k + 1]) for k in range(a.steps)]
    per_step = sorted(raw_steps)
    median_ms = per_step[len(per_step) // 2]
    if dist_on:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    last_loss = float(loss.item())

    if rank == 0:
        global_batch = a.batch * world
        sps = global_batch * a.steps / dt
        rec = {
            "metric": "train samples/sec on 4x128^3 volumes", "value": sps, "unit": "samples/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "HDenseFormer_32 3D train step (fwd + DeepSuper CE+Dice + bwd + Adam), "
                                   "in=4 n_cls=4 128^3 transformer_depth=24 (BASELINE configs[1])",
                       "global_batch": global_batch, "per_gpu_batch": a.batch, "parallelism": f"dp{world}",
                       "dropout": "on (train mode)", "loss": last_loss,
                       "collective": (torch.distributed.get_backend() if dist_on else None)},
            "step_tflops": FWD_BWD_GFLOP_PER_SAMPLE * sps / 1e3,
            "hip_event_ms_per_step": {"median": median_ms, "min": per_step[0], "max": per_step[-1],
                                      "slowest_step": raw_steps.index(per_step[-1]),
                                      "all": [round(v, 3) for v in raw_steps]},
            "host_issue_ms_per_step": t_enqueued / a.steps * 1e3,
        }
        if world == 1 and not a.no_roofline:
            wr = roofline_dominant_kernel(dev)
            rec["roofline"] = roofline_conv_family(dev)
            rec["roofline"]["step"], why = step_roofline_from_profile()
            if why:
                rec["roofline"]["step_unavailable"] = why
                log("roofline.step:", why)
            rec["roofline"]["wr"] = wr      # the single conv_wr_kernel launch of a step (rounds 1-4 reported this one)
            # HBM bytes of the family's 14 launches from the committed PMC passes of the same build (else null), and the
            # family's fraction INSIDE the step (one-stream kernel trace of the same build) next to the live replay
            stp = rec["roofline"]["step"]
            if stp:
                fam = stp.get("families", {}).get("conv_ws2_kernel", {})
                rec["roofline"]["traffic"] = fam.get("hbm_bytes") or None
                rec["roofline"]["traffic_source"] = stp["source"]
                if fam.get("us_1s"):
                    rec["roofline"]["frac_in_step_profile"] = (rec["roofline"]["flops_per_pass"] / (fam["us_1s"] * 1e-6) / 1e12
                                                               / BF16_MFMA_PEAK_TFLOPS)
            if in_step_ms:
                # the same launch as it runs inside the timed steps (HIP events recorded by the library around it, one
                # pair per step): `frac` above stays the back-to-back figure of the earlier rounds, this is what the
                # kernel does in the workload (the chip is not at its power cap between two launches of a step)
                ms_in = sum(in_step_ms) / len(in_step_ms)
                fl = wr["flops_per_launch"]
                wr["in_step"] = {
                    "avg_launch_ms": ms_in, "median_ms": in_step_ms[len(in_step_ms) // 2], "min_ms": in_step_ms[0],
                    "max_ms": in_step_ms[-1], "launches": len(in_step_ms),
                    "achieved": fl / (ms_in * 1e-3) / 1e12, "frac": fl / (ms_in * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS}
        if world == 1:
            if not a.no_cpu_baseline:
                del net, opt
                torch.cuda.empty_cache()
                rec["cpu_baseline"] = cpu_baseline()
                if rec["cpu_baseline"]["value"]:
                    rec["gpu_over_cpu"] = sps / rec["cpu_baseline"]["value"]
        print(json.dumps(rec), flush=True)
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
