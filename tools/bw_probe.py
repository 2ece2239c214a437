import torch, time
x = torch.empty(512*1024*1024, dtype=torch.bfloat16, device="cuda")  # 1 GiB
y = torch.empty_like(x)
for f, name, nbytes in [(lambda: y.copy_(x), "copy 1GiB->1GiB", 2*x.numel()*2), (lambda: x.add_(1), "add_ inplace", 2*x.numel()*2), (lambda: torch.add(x, y, out=y), "a+b->b", 3*x.numel()*2), (lambda: x.sum(), "sum", x.numel()*2)]:
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print(f"{name}: {ms*1e3:.0f} us  {nbytes/ms/1e9:.2f} TB/s")
