"""Epoch time of the reference's 3-layer network (gat_amd.model.GAT) on a synthetic graph,
with a per-kernel breakdown from torch.profiler: how much of an epoch is the GAT layer
(libgat_mi355 kernels) and how much the stock glue around it."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gat_amd import synth  # noqa: E402
from gat_amd.data import Data  # noqa: E402
from gat_amd.model import GAT  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="arxiv")
    ap.add_argument("--hidden", type=int, default=32)
    ap.add_argument("--classes", type=int, default=40)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--hipgraph", action="store_true", help="replay the step as a captured HIP graph")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--bf16", action="store_true", help="bf16 feature tables / projection where covered")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ei, n, f, _, _ = synth.workload_graph(args.workload, 0, dev, args.scale)
    g = torch.Generator(device=dev).manual_seed(0)
    data = Data(torch.randn(n, f, generator=g, device=dev), ei,
                torch.randint(0, args.classes, (n,), generator=g, device=dev))
    model = GAT(torch.optim.AdamW, f, args.classes, dropout=0.425, hidden_dim=args.hidden, lr=0.01,
                device=dev, verbose=False, hipgraph=args.hipgraph,
                feature_dtype=torch.bfloat16 if args.bf16 else None)
    for _ in range(3):
        model.train_epoch(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.epochs):
        model.train_epoch(data)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.epochs * 1e3
    print(f"{args.workload} x{args.scale}: N={n} E={ei.size(1)} hidden=8x{args.hidden}"
          f"{' hipgraph' if args.hipgraph else ''}{' bf16' if args.bf16 else ''}: train epoch {ms:.3f} ms", flush=True)
    if args.no_profile:
        return
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            model.train_epoch(data)
        torch.cuda.synchronize()
    rows = sorted(prof.key_averages(), key=lambda r: -r.device_time_total)
    tot = sum(r.device_time_total for r in rows)
    ours = sum(r.device_time_total for r in rows if "gatmi" in r.key or "rocprim" in r.key)
    print(f"device time/epoch {tot / 3e3:.3f} ms, libgat_mi355 kernels {100 * ours / tot:.1f} %")
    for r in rows[:int(os.environ.get("GAT_PROF_ROWS", "26"))]:
        print(f"  {r.device_time_total / 3e3:8.3f} ms  {r.count // 3:4d}x  {r.key[:90]}")


if __name__ == "__main__":
    main()
