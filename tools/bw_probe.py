"""Streaming yardsticks on this GPU (torch ops): read-only, write-only, copy.  Not part of the product."""
import torch
dev = torch.device("cuda")
n = 640 * 1024 * 1024          # 2.5 GiB of fp32
x = torch.randn(n, device=dev); y = torch.empty_like(x)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
gb = n * 4 / 1e9
ms = t(lambda: x.sum()); print(f"read  {gb/ms:.2f} TB/s ({ms:.3f} ms)")
ms = t(lambda: y.fill_(1.0)); print(f"write {gb/ms:.2f} TB/s ({ms:.3f} ms)")
ms = t(lambda: y.copy_(x)); print(f"copy  {2*gb/ms:.2f} TB/s r+w ({ms:.3f} ms)")
ms = t(lambda: torch.add(x, 1.0, out=y)); print(f"add   {2*gb/ms:.2f} TB/s r+w ({ms:.3f} ms)")
