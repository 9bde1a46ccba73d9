"""attention kernels on f32 against bf16 feature tables, per-kernel times (products-shaped graph)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gat_amd import _lib, ops, synth
from gat_amd.graph import build_graph
wl = sys.argv[1] if len(sys.argv) > 1 else "products_hidden"
dev = torch.device("cuda:0")
ei, n, f, h, c = synth.workload_graph(wl, 0, dev)
layout = build_graph(ei, n)
g = torch.Generator(device=dev).manual_seed(0)
hp = torch.randn(n, h * c, generator=g, device=dev) * 0.3
go = torch.randn(n, h * c, generator=g, device=dev)
a_dst = torch.randn(n, h, generator=g, device=dev) * 0.3
att = torch.randn(h * c, generator=g, device=dev) * 0.1
b = torch.zeros(h * c, device=dev)
for bf in (False, True):
    cfg = ops.AttnConfig(h, c, 0.2, 0.425, _lib.MASK_HASH, seed=5)
    cfg.feat_bf16 = bf
    tab = hp.to(torch.bfloat16) if bf else hp
    for it in range(4):
        if it == 1:
            _lib.prof_read(); _lib.prof_enable(True)
        out, lse = ops.attn_fwd(layout, cfg, f, tab, a_dst, att, att, b, True)
        r = ops.attn_bwd(layout, cfg, f, tab, a_dst, att, att, b, out, lse, go, defer_dst_term=True)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    per = {}
    for t, ms in _lib.prof_read():
        per.setdefault(t, []).append(ms)
    print("bf16" if bf else "f32 ", {k: round(sum(v) / len(v), 3) for k, v in per.items()}, flush=True)
