"""A/B of the f32 projection GEMMs: native f32 MFMA (GATMI_SPLIT3=0) against the three-way bf16
split on the bf16 matrix cores (bits 1 fwd, 2 d_x, 4 d_W): error against an fp64 product on a
sample of rows, and time per call at the full row count.

    python tools/split_probe.py [--rows 2449029] [--f 256] [--heads 8] [--channels 32]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gat_amd import _lib, ops  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def rel(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2449029)
    ap.add_argument("--f", type=int, default=256)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--channels", type=int, default=32)
    ap.add_argument("--sample", type=int, default=200000)
    ap.add_argument("--masks", default="0,1,3,7")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    n, f, h, c = a.rows, a.f, a.heads, a.channels
    x = torch.randn(n, f, device=dev, generator=g)
    w = torch.randn(h * c, f, device=dev, generator=g) / f ** 0.5
    att = torch.randn(h * c, device=dev, generator=g)
    d_hp = torch.randn(n, h * c, device=dev, generator=g)
    d_a = torch.randn(n, h, device=dev, generator=g)
    m = min(n, a.sample)
    xs, ds, das = x[:m].contiguous(), d_hp[:m].contiguous(), d_a[:m].contiguous()
    ref_h = xs.double() @ w.double().t()
    dfull = ds.double() + (das.double().repeat_interleave(c, 1) * att.double())
    ref_dx = dfull @ w.double()
    ref_dw = dfull.t() @ xs.double()
    out = {}
    for mask in [int(v) for v in a.masks.split(",")]:
        _lib.set_option("GATMI_SPLIT3", mask)
        hp, _ = ops.project_fwd(xs, w, att, h, c)
        d_w, d_x = ops.project_bwd(xs, w, ds, h, c, True, das, att)
        row = {"err_h": rel(hp, ref_h), "err_dx": rel(d_x, ref_dx), "err_dw": rel(d_w, ref_dw)}
        row["ms_fwd"] = timed(lambda: ops.project_fwd(x, w, att, h, c))
        row["ms_bwd"] = timed(lambda: ops.project_bwd(x, w, d_hp, h, c, True, d_a, att))
        row["ms_bwd_w_only"] = timed(lambda: ops.project_bwd(x, w, d_hp, h, c, False, d_a, att))
        out[mask] = row
        print(mask, json.dumps(row), flush=True)
    _lib.set_option("GATMI_SPLIT3", -1)


if __name__ == "__main__":
    main()
