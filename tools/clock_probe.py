"""Sample sclk / package power with rocm-smi while the roofline conv kernel runs back to back (diagnostic)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
import torch

from hdf_rt._lib import BF16, check, lib, ptr

cin, cout, s, n = int(sys.argv[1]), int(sys.argv[2]), 128, 2
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(n, s, s, s, cin, device=dev).to(torch.bfloat16)
w = (torch.randn(27 * 32 * ((cout + 31) // 32) * cin, device=dev) * 0.02).to(torch.bfloat16)
out = torch.empty(n, s, s, s, cout, device=dev, dtype=torch.bfloat16)
tiles = lib().hdf_op_conv3d_stat_tiles(BF16, cin, s, s, s)
part = torch.empty(n * tiles * 64 * 2, device=dev)


def launch():
    check(lib().hdf_op_conv3d(BF16, 0, ptr(x), cin, cin, n, s, s, s, ptr(w), None, None, None, 1, ptr(out), cout, cout,
                              ptr(part), 0, st), "conv")


launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
reps = 20000
for _ in range(reps):
    launch()
e1.record()
for k in range(4):
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True)
    print([l.split(":")[-1].strip() for l in r.stdout.splitlines() if "sclk" in l or "Power (W)" in l], flush=True)
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"{cin}->{cout}: {ms*1e3:.1f} us  {2.0*27*cin*cout*s**3*n/ms/1e9:.1f} TFLOP/s")
