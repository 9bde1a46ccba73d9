"""What ONE rank of a P-way destination partition computes per step of exchange "h", measured on one
GPU: the rank's projection, its per-piece attention forward + combine, its per-piece backward, for
P = 2, 4, 8 and K = 1, 2, 4 pieces -- the kernel side of DESIGN.md section 7's overlap estimate (the
collectives are not run: the table is filled locally).
usage: python tools/piece_bench.py [--workload products_hidden] [--ranks 0,3] [--dtype f32|bf16]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="products_hidden")
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--pieces", default="1,2,4")
    ap.add_argument("--ranks", default="0")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    from gat_amd import _lib, ops, synth
    from gat_amd.dist import HipBackend, chunk_edge_lists, make_partition, _piece_cfg
    dev = torch.device("cuda:0")
    ei, n, f, h, c = synth.workload_graph(args.workload, 0, dev)
    be = HipBackend()
    g = torch.Generator(device=dev).manual_seed(0)
    w = torch.randn(h * c, f, generator=g, device=dev) / f ** 0.5
    att_s = torch.randn(h * c, generator=g, device=dev) * 0.1
    att_d = torch.randn(h * c, generator=g, device=dev) * 0.1
    bias = torch.zeros(h * c, device=dev)
    x_all = torch.randn(n, f, generator=g, device=dev)
    for world in [int(v) for v in args.worlds.split(",")]:
        for rank in [int(v) for v in args.ranks.split(",")]:
            if rank >= world:
                continue
            part = make_partition(ei, n, world, rank, "edges")
            x = x_all[part.lo:part.hi].contiguous()
            go = torch.randn(part.n_local, h * c, generator=g, device=dev)
            for k in [int(v) for v in args.pieces.split(",")]:
                cfg = ops.AttnConfig(h, c, dropout=0.425, mask_mode=_lib.MASK_HASH, seed=7)
                cfg.feat_bf16 = args.dtype == "bf16"
                tm = part.table(k)
                lists = chunk_edge_lists(ei, part, tm)
                layouts = [be.build_chunk_layout(e_, tm.rows[s], part.n_local) for s, (e_, _) in enumerate(lists)]
                del lists
                hp_own, a_dst = be.project_fwd(x, w, att_d, cfg)
                table = torch.randn(tm.total, hp_own.size(1), generator=g, device=dev).to(hp_own.dtype) * 0.3
                piece = lambda t, s: t[tm.base[s]: tm.base[s] + tm.rows[s]]

                def step(timers):
                    ev = lambda: torch.cuda.Event(enable_timing=True)
                    t = [ev() for _ in range(5)]
                    t[0].record()
                    hp_l, a_d = be.project_fwd(x, w, att_d, cfg)
                    t[1].record()
                    triples = [be.attn_fwd_partial(layouts[s], _piece_cfg(cfg, s), f, piece(table, s), a_d, att_s)
                               for s in range(k)]
                    out, lse = be.attn_fwd_combine(triples, cfg, bias, True)
                    t[2].record()
                    cb = be.chunked_backward(cfg, f, a_d, att_s, att_d, bias, out, lse, go, k)
                    d = [cb.chunk(s, layouts[s], _piece_cfg(cfg, s), piece(table, s)) for s in range(k)]
                    d_as, d_ad, d_b, d_a_dst = cb.end(hp_l)
                    t[3].record()
                    # stand-in for the reduce-scatter's result: this rank's rows of d_h
                    d_hp = torch.zeros(part.n_local, d[0].size(1), device=dev)
                    take = min(part.n_local, d[0].size(0))
                    d_hp[:take] = d[0][:take]
                    be.add_dst_term(d_hp, d_a_dst, att_d, cfg)
                    be.project_bwd(x, w, d_hp, cfg, True)
                    t[4].record()
                    timers.append(t)

                for _ in range(2):
                    step([])
                _lib.prof_read()
                torch.cuda.synchronize()
                _lib.prof_enable(True)
                timers = []
                for _ in range(args.steps):
                    step(timers)
                torch.cuda.synchronize()
                _lib.prof_enable(False)
                per = {}
                for tag, ms in _lib.prof_read():
                    per[tag] = per.get(tag, 0.0) + ms / args.steps
                phase = [sum(t[i].elapsed_time(t[i + 1]) for t in timers) / args.steps for i in range(4)]
                dp = hp_own.size(1)
                wire = (tm.total - sum(tm.cap[s][rank] for s in range(k))) * dp * hp_own.element_size()
                print(json.dumps({
                    "workload": args.workload, "dtype": args.dtype, "world": world, "rank": rank, "pieces": k,
                    "rows_local": part.n_local, "edges_local": sum(l.num_edges for l in layouts),
                    "project_fwd_ms": round(phase[0], 3), "attn_fwd_ms": round(phase[1], 3),
                    "attn_bwd_ms": round(phase[2], 3), "project_bwd_ms": round(phase[3], 3),
                    "step_ms": round(sum(phase), 3),
                    "recv_bytes_fwd": wire, "send_recv_bytes_bwd_f32": (tm.total - sum(tm.cap[s][rank] for s in range(k))) * dp * 4,
                    "kernels": {kk: round(v, 3) for kk, v in sorted(per.items())}}), flush=True)
                del layouts, table
                torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
