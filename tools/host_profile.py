"""Where the host time of a small-graph step goes (cProfile over 300 steps of one layer)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gat_amd import GATConv, synth
wl = sys.argv[1] if len(sys.argv) > 1 else "cora"
dev = torch.device("cuda:0")
ei, n, f, h, c = synth.workload_graph(wl, 0, dev)
x = torch.randn(n, f, device=dev)
go = torch.randn(n, h * c, device=dev)
conv = GATConv(f, c, heads=h, dropout=0.425).to(dev).train()
def step():
    conv.zero_grad(set_to_none=True)
    conv(x, ei).backward(go)
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): step()
torch.cuda.synchronize()
print(f"{wl}: {(time.perf_counter() - t0) / 300 * 1e3:.4f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
