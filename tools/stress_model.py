"""Run-only sweep of the reference's network over the widths / class counts / sizes its searches visit
(hidden_dim 24..40 at 8 heads, 2..10 classes), f32 and bf16 tables, eager and HIP-graph replay: three
training epochs + one evaluation each, loss finite.  Finds shape-dependent host / launch errors that the
parity tests' small graphs do not reach (round 3: SIGFPE of the split GEMM launcher on conv3 from 4096 rows)."""
import itertools, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gat_amd import synth
from gat_amd.data import Data
from gat_amd.model import GAT

dev = torch.device("cuda:0")
ok = bad = 0
for n, hidden, classes, bf16, graph in itertools.product((5000, 40000), (24, 28, 30, 33, 36, 40), (2, 3, 10), (False, True), (False, True)):
    if graph and (hidden not in (30, 36) or classes == 3):
        continue
    ei = synth.out_degree_k_graph(n, 5, seed=n, device=dev)
    g = torch.Generator(device=dev).manual_seed(1)
    data = Data(torch.randn(n, 25, generator=g, device=dev), ei, torch.randint(0, classes, (n,), generator=g, device=dev))
    tag = f"n={n} hidden={hidden} classes={classes} bf16={bf16} hipgraph={graph}"
    try:
        torch.manual_seed(0)
        model = GAT(torch.optim.AdamW, 25, classes, dropout=0.425, hidden_dim=hidden, lr=0.01, device=dev, verbose=False,
                    hipgraph=graph, feature_dtype=torch.bfloat16 if bf16 else None)
        for _ in range(3):
            r = model.train_epoch(data)
        v = model.validate_epoch(data)
        torch.cuda.synchronize()
        fin = all(torch.isfinite(p).all().item() for p in model.parameters())
        assert fin and r[0] == r[0] and v[0] == v[0], "non-finite parameters or loss"
        ok += 1
        print("ok  ", tag, f"train loss {r[0]:.4f} val loss {v[0]:.4f}", flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL", tag, type(e).__name__, str(e)[:300], flush=True)
print(f"{ok} ok, {bad} failed")
sys.exit(1 if bad else 0)
