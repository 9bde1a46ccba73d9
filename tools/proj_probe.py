"""Times of the projection GEMMs (h = x W^T, d_x, d_W) at the hidden-layer shapes of the
3-layer network, from torch.profiler.  Usage: python tools/proj_probe.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from gat_amd import ops

dev = torch.device("cuda:0")
for n, f, h, c in [(169343, 256, 8, 32), (169343, 128, 8, 32), (100000, 240, 8, 30), (2449029, 256, 4, 64)]:
    x = torch.randn(n, f, device=dev)
    w = torch.randn(h * c, f, device=dev) * 0.1
    ad = torch.randn(h, c, device=dev)
    cp = ops.padded_channels(c)
    dhp = torch.randn(n, h * cp, device=dev)
    dad = torch.randn(n, h, device=dev)
    for _ in range(2):
        ops.project_fwd(x, w, ad, h, c)
        ops.project_bwd(x, w, dhp, h, c, True, dad, ad)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            ops.project_fwd(x, w, ad, h, c)
            ops.project_bwd(x, w, dhp, h, c, True, dad, ad)
        torch.cuda.synchronize()
    flops = 2.0 * n * f * h * cp
    print(f"N={n} F={f} {h}x{c}: ideal {flops / 157e12 * 1e3:.3f} ms per GEMM at 157 TFLOP/s")
    for r in sorted(prof.key_averages(), key=lambda r: -r.device_time_total)[:6]:
        print(f"   {r.device_time_total / 5e3:8.3f} ms {r.count // 5}x {r.key[:100]}")
