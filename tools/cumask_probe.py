"""CU-masked streams on MI355X (hipExtStreamCreateWithCUMask): (1) which (XCC, SE, CU) a mask bit selects,
(2) what a persistent conv launch costs on 224 of 256 CUs, (3) a latency-bound kernel chain (the transformer's
attention op) on the remaining 32 CUs, alone and next to the conv.  Uses tools/probe/libcumask_probe.so
(hipcc --offload-arch=gfx950 -shared -fPIC tools/probe/cumask_probe.hip) and the C ABI of libhdf_hip.so."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "h-denseformer_amd")):
    sys.path.insert(0, p)
import torch

from hdf_rt._lib import BF16, check, lib, ptr

dev = "cuda:0"
torch.cuda.set_device(0)
P = C.CDLL(os.path.join(ROOT, "tools", "probe", "libcumask_probe.so"))
P.probe_create_stream.argtypes = [C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_void_p)]
P.probe_where.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong]
P.probe_get_mask.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int]


def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = P.probe_create_stream(words, 8, C.byref(s))
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask rc={rc}")
    return s.value


def where(stream, blocks=4096, spin=40000):
    out = torch.zeros(2 * blocks, dtype=torch.int32, device=dev)
    rc = P.probe_where(stream, out.data_ptr(), blocks, 64, spin)
    assert rc == 0, rc
    torch.cuda.synchronize()
    o = out.cpu().view(blocks, 2)
    xcc = (o[:, 0] & 0xF).tolist()
    hw = o[:, 1].tolist()
    cus = {}
    for x, h in zip(xcc, hw):
        key = (x, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15)   # xcc, se, sh, cu
        cus[key] = cus.get(key, 0) + 1
    return cus


def show(name, cus):
    per_xcc = {}
    for (x, se, sh, cu), n in cus.items():
        per_xcc.setdefault(x, []).append((se, sh, cu))
    print(f"{name}: {len(cus)} distinct CUs; per XCC:", {x: len(v) for x, v in sorted(per_xcc.items())})
    return per_xcc


full = where(0)
show("null stream", full)
for name, bits in [("bits 0-7", range(8)), ("bits 0-31", range(32)), ("bits 0,8,16,24", [0, 8, 16, 24]),
                   ("bits 32-63", range(32, 64)), ("bits 224-255", range(224, 256))]:
    s = masked_stream(list(bits))
    px = show(name, where(s))
    if len(bits) <= 8:
        print("    ", {x: sorted(v) for x, v in px.items()})

small = masked_stream(list(range(224, 256)))
big = masked_stream(list(range(0, 224)))
s_small, s_big = torch.cuda.ExternalStream(small), torch.cuda.ExternalStream(big)
show("small (224-255)", where(small))
show("big (0-223)", where(big))

# ---- conv 64->32 @128^3 on the full chip vs on 224 CUs
n, cin, cout, s = 2, 64, 32, 128
x = torch.randn(n, s, s, s, cin, device=dev).to(torch.bfloat16)
w = (torch.randn(27 * 32 * cin, device=dev) * 0.02).to(torch.bfloat16)
out = torch.empty(n, s, s, s, cout, device=dev, dtype=torch.bfloat16)
tiles = lib().hdf_op_conv3d_stat_tiles(BF16, cin, s, s, s)
part = torch.empty(n * tiles * 32 * 2, device=dev)


def conv(stream):
    check(lib().hdf_op_conv3d(BF16, 0, ptr(x), cin, cin, n, s, s, s, ptr(w), None, None, None, 0, ptr(out), cout, cout,
                              ptr(part), 0, stream), "conv")


# ---- latency-bound chain: the attention op at the bench geometry (N = 512 tokens, 4 modalities x 2 samples)
ntok, groups = 512, 8
qkv = torch.randn(groups * ntok, 96, device=dev)
o = torch.empty(groups * ntok, 32, device=dev)
lse = torch.empty(groups * 8 * ntok, device=dev)


def attn(stream):
    check(lib().hdf_op_attention_fwd(ptr(qkv), groups, ntok, ptr(o), ptr(lse), stream), "attn")


def timed(fn, stream_obj, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream_obj):
        for _ in range(5):
            fn(stream_obj.cuda_stream)
        e0.record()
        for _ in range(reps):
            fn(stream_obj.cuda_stream)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cur = torch.cuda.current_stream()
print("conv full chip, grid 256: %.1f us" % timed(conv, cur, 30))
print("attn full chip: %.2f us per launch" % timed(attn, cur, 200))
check(lib().hdf_set_cu_budget(224), "budget")
print("conv full chip, grid 224: %.1f us" % timed(conv, cur, 30))
print("conv on 224-CU stream, grid 224: %.1f us" % timed(conv, s_big, 30))
check(lib().hdf_set_cu_budget(256), "budget")
print("conv on 224-CU stream, grid 256: %.1f us" % timed(conv, s_big, 30))
print("attn on 32-CU stream: %.2f us per launch" % timed(attn, s_small, 200))

# together: conv back to back on the big stream while the attention chain runs on the small one
check(lib().hdf_set_cu_budget(224), "budget")
ec0, ec1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ea0, ea1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
with torch.cuda.stream(s_big):
    ec0.record()
    for _ in range(30):
        conv(big)
    ec1.record()
with torch.cuda.stream(s_small):
    ea0.record()
    for _ in range(400):
        attn(small)
    ea1.record()
torch.cuda.synchronize()
print("together: conv %.1f us per launch (224 CUs), attn %.2f us per launch (32 CUs)" %
      (ec0.elapsed_time(ec1) / 30 * 1e3, ea0.elapsed_time(ea1) / 400 * 1e3))
# the same pairing without masks: two plain streams
check(lib().hdf_set_cu_budget(256), "budget")
p1, p2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
with torch.cuda.stream(p1):
    ec0.record()
    for _ in range(30):
        conv(p1.cuda_stream)
    ec1.record()
with torch.cuda.stream(p2):
    ea0.record()
    for _ in range(400):
        attn(p2.cuda_stream)
    ea1.record()
torch.cuda.synchronize()
print("together, unmasked streams: conv %.1f us per launch, attn %.2f us per launch" %
      (ec0.elapsed_time(ec1) / 30 * 1e3, ea0.elapsed_time(ea1) / 400 * 1e3))
