"""One GAT layer of any shape on a named synthetic graph: fwd+bwd ms/step and the library's per-kernel
averages (HIP events).  python tools/layer_bench.py --graph products --f 100 --heads 8 --channels 32 [--dx]
GATMI_LIB / GATMI_* switches select builds and kernels as everywhere else."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gat_amd import GATConv, synth, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--graph", default="products")
ap.add_argument("--f", type=int, default=100)
ap.add_argument("--heads", type=int, default=8)
ap.add_argument("--channels", type=int, default=32)
ap.add_argument("--dx", action="store_true", help="x carries a gradient (hidden layer)")
ap.add_argument("--dropout", type=float, default=0.425)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--bf16", action="store_true", help="feature_dtype=torch.bfloat16 (bf16 rows for the gathers where covered)")
a = ap.parse_args()
dev = torch.device("cuda:0")
ei, n, _, _, _ = synth.workload_graph(a.graph, 0, dev, a.scale)
torch.manual_seed(0)
conv = GATConv(a.f, a.channels, heads=a.heads, dropout=a.dropout,
               feature_dtype=torch.bfloat16 if a.bf16 else None).to(dev).train()
x = torch.randn(n, a.f, device=dev, requires_grad=a.dx)
go = torch.randn(n, a.heads * a.channels, device=dev)

def step():
    conv.zero_grad(set_to_none=True)
    x.grad = None
    conv(x, ei).backward(go)

for _ in range(2):
    step()
torch.cuda.synchronize()
_lib.prof_read()
_lib.prof_enable(True)
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / a.steps * 1e3
_lib.prof_enable(False)
per = {}
for tag, v in _lib.prof_read():
    per.setdefault(tag, []).append(v)
print(f"{a.graph} x{a.scale}: {a.f} -> {a.heads}x{a.channels}{' +d_x' if a.dx else ''}{' bf16' if a.bf16 else ''}: {ms:.3f} ms/step",
      {k: round(sum(v) / len(v), 3) for k, v in per.items()})
