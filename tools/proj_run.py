"""The projection GEMMs of tools/proj_probe.py without a profiler inside, for rocprofv3 passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gat_amd import ops

dev = torch.device("cuda:0")
for n, f, h, c in [(169343, 256, 8, 32), (2449029, 256, 4, 64)]:
    x = torch.randn(n, f, device=dev)
    w = torch.randn(h * c, f, device=dev) * 0.1
    ad = torch.randn(h, c, device=dev)
    dhp = torch.randn(n, h * ops.padded_channels(c), device=dev)
    dad = torch.randn(n, h, device=dev)
    for _ in range(4):
        ops.project_fwd(x, w, ad, h, c)
        ops.project_bwd(x, w, dhp, h, c, True, dad, ad)
    torch.cuda.synchronize()
