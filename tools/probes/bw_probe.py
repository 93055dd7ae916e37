"""HBM bandwidth reference points with plain torch kernels: fill (write only), copy (read+write), sum (read only)."""
import torch
dev = torch.device("cuda:0")
n = 1 << 29  # 512 Mi bf16 = 1 GiB
x = torch.randn(n, device=dev, dtype=torch.float32).to(torch.bfloat16)
y = torch.empty_like(x)


def t(fn, byts, name, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:10s} {ms * 1e3:9.1f} us  {byts / ms / 1e9:7.2f} TB/s")


t(lambda: y.fill_(1.0), n * 2, "fill")
t(lambda: y.copy_(x), n * 4, "copy")
t(lambda: x.sum(), n * 2, "sum")
t(lambda: torch.add(x, x, out=y), n * 4, "add")
for mb in (64, 256, 512):
    m = mb * 1024 * 1024 // 2
    t(lambda: y[:m].copy_(x[:m]), m * 4, f"copy{mb}MB")
