"""A/B of the multi-edge kernels for 5..8 heads (GATMI_EDGE8) on layers and degrees the named
workloads do not cover.  Usage: python tools/edge8_ab.py   (on a GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gat_amd import GATConv, _lib, synth

dev = torch.device("cuda:0")
CASES = [  # nodes, edges, max degree, F, H, C
    (100000, 500000, 0, 25, 8, 30),
    (200000, 5000000, 3000, 25, 8, 30),
    (200000, 5000000, 3000, 60, 8, 16),
    (200000, 5000000, 3000, 128, 8, 32),
    (169343, 1166243, 13000, 128, 8, 32),
    (200000, 5000000, 3000, 100, 8, 30),
    (200000, 1400000, 3000, 100, 8, 30),
    (200000, 5000000, 3000, 100, 6, 64),
]
for n, e, dmax, f, h, c in CASES:
    ei = synth.powerlaw_graph(n, e, dmax, seed=1, device=dev)
    x = torch.randn(n, f, device=dev)
    go = torch.randn(n, h * c, device=dev)
    res = {}
    for mode in (1, 0, "rows"):
        _lib.set_option("GATMI_EDGE8", -1 if mode == "rows" else mode)
        conv = GATConv(f, c, heads=h, dropout=0.425, input_aggregation=mode != "rows",
                       stream=False).to(dev).train()
        for it in range(13):
            if it == 3:
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            conv.zero_grad()
            conv(x, ei).backward(go)
        b.record()
        torch.cuda.synchronize()
        res[mode] = a.elapsed_time(b) / 10
    _lib.set_option("GATMI_EDGE8", -1)
    print(f"n={n} e={e} dmax={dmax} {f}->{h}x{c}: multi-edge {res[1]:.3f} ms, one-edge {res[0]:.3f} ms, "
          f"projected rows {res['rows']:.3f} ms", flush=True)
