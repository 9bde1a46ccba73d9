"""bench.py -- GAT layer fwd+bwd edges/sec on MI355X (BASELINE.json's metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
                  [--workload products|arxiv|cora|runpy100k|products_uniform|arxiv_uniform|cora_uniform]

A step is one forward + backward of one GAT layer (projection, fused attention /
softmax / aggregation, all gradients) over the whole synthetic graph, inputs resident
in HBM, graph layout built beforehand (one-off, reported separately).  N > 1 is launched
by torch.distributed.run, one rank per GPU, destination-partitioned (gat_amd/dist.py).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


from gat_amd.synth import kernel_bytes  # noqa: E402  (algorithmic bytes per launch)


def measured_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the rocprofv3 --pmc passes committed under profiles/
    (bench.py cannot collect counters itself: this is a RECORDED figure of the same command, not
    one measured in this run) -> (bytes or None, where it came from)."""
    for rnd in ("r3", "r2", "r1"):
        path = os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")
        try:
            with open(path) as f:
                t = json.load(f)
            if t.get("workload") == workload and t["kernels"].get(kernel) is not None:
                return t["kernels"][kernel], (f"profiles/{rnd}/pmc_traffic.json: recorded rocprofv3 --pmc "
                                              "FETCH_SIZE/WRITE_SIZE passes of this command, not measured in this run")
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def cpu_baseline(name, seed, budget_s=25.0):
    """The oracle (CPU restatement of the reference's path) timed on this host's cores on
    a bounded sample of the same workload: the same generator at reduced scale."""
    from gat_amd import synth
    from oracle import gat_oracle as O
    n_full, e_full = synth.WORKLOADS[name][:2]
    scale = min(1.0, 1.5e6 / e_full)
    ei, n, f, h, c = synth.workload_graph(name, seed, "cpu", scale)
    cores = os.cpu_count() or 1
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    torch.set_num_threads(cores)
    w, a_s, a_d, b = O.init_params(f, h, c, seed)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, f, generator=g)
    go = torch.randn(n, h * c, generator=g)
    ep = ei.size(1) - int((ei[0] == ei[1]).sum()) + n
    times = []
    t_all = time.time()
    for it in range(11):            # 1 warm-up + up to 10 timed runs, inside the budget
        t0 = time.time()
        O.gatconv_grads_autograd(x, ei, w, a_s, a_d, b, h, c, go)
        times.append(time.time() - t0)
        if time.time() - t_all > budget_s:
            break
    best = sorted(times[1:] or times)[len(times[1:] or times) // 2]
    torch.set_num_threads(1)        # "also report a 1-thread figure" (one run)
    t0 = time.time()
    O.gatconv_grads_autograd(x, ei, w, a_s, a_d, b, h, c, go)
    single = time.time() - t0
    torch.set_num_threads(cores)
    return {"value": ep / best, "unit": "edges/s", "cores": cores, "kind": "port",
            "single_thread_value": ep / single,
            "sample": f"{name}-shaped generator at scale {scale:.4f} ({n} nodes, {ep} edges after "
                      f"self-loop normalisation), oracle/gat_oracle.py fwd+bwd incl. self-loop "
                      f"normalisation, median of {max(len(times) - 1, 1)} timed runs after 1 warm-up, "
                      f"torch CPU threads={cores}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="products")
    ap.add_argument("--dropout", type=float, default=0.425)  # run.py:21, attention dropout on
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--input-aggregation", default="auto", choices=["auto", "on", "off"],
                    help="N=1: aggregate x rows per head before projecting (gat_amd/csrc/inagg.hip); "
                         "auto = what GATConv picks (on here: x carries no gradient, F < H*C)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "h", "x", "xs", "as"],
                    help="N>1: what the destination partition all-gathers (gat_amd/dist.py)")
    ap.add_argument("--partition", default="edges", choices=["edges", "nodes"],
                    help="N>1: destination ranges of equal in-edge count (SURVEY.md 8e) or equal node count")
    ap.add_argument("--chunks", type=int, default=0,
                    help="N>1, exchange h: pieces the table travels in, attention over piece k overlapping the "
                         "collective of piece k+1 (0: 4 when N > 1, else 1)")
    ap.add_argument("--padding", default="auto", choices=["auto", "pad", "exact"],
                    help="N>1: gathered tables padded to the longest range (equal-shard collectives) or exact "
                         "(per-owner collectives); auto: exact when padding exceeds 1.25 rows per node")
    ap.add_argument("--no-exchange-h", action="store_true",
                    help="N>1: skip the second timed region (exchange=h, the north_star's all-gather of "
                         "projected features) that is reported next to --exchange auto")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="bf16: projected rows kept as a bf16 table, projection on the bf16 matrix cores "
                         "(layers on the projected-row path only; tolerance: tests/test_gpu_bf16.py)")
    ap.add_argument("--hipgraph", action="store_true",
                    help="N=1: also time the same step replayed as ONE captured HIP graph (dropout seed on "
                         "the device, GATMI_MASK_HASH_DEV) -> `hipgraph` in the line; for launch-bound graphs")
    ap.add_argument("--no-extra-layers", action="store_true",
                    help="N=1: skip the hidden-layer and model-sum measurements (products only)")
    args = ap.parse_args()
    if os.environ.get("GAT_HANG_DUMP"):      # debugging aid: dump every thread's stack and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["GAT_HANG_DUMP"]), exit=True)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 "
                             "--nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N")
    local = local % max(torch.cuda.device_count(), 1)   # >1 rank per GPU only in gloo rehearsals
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from gat_amd import synth

    # GAT_FORCE_PARTITIONED=1: the N>1 path on a one-rank process group (RCCL rehearsal on one GPU)
    if world > 1 or os.environ.get("GAT_FORCE_PARTITIONED"):
        import torch.distributed as dist
        backend = os.environ.get("GAT_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        from gat_amd.dist import bench_partitioned
        return bench_partitioned(args, rank, world, dev)

    head = measure_layer(args.workload, args, dev)
    n, ep, f, h, c, avg, kb = (head[k] for k in ("nodes", "edges", "f", "h", "c", "avg", "kb"))
    dom = max(kb, key=lambda k: avg.get(k, 0.0))
    achieved = kb[dom] / (avg[dom] * 1e-3) / 1e9
    traffic, traffic_src = measured_traffic(args.workload, dom)
    s8d = survey_8d_bytes(n, ep, h, c)
    path = "input-side aggregation" if ("inagg_fwd" in avg or "inagg_fwd_fused" in avg) else "projected rows"
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "avg_ms": round(avg[dom], 4), "algorithmic_bytes": kb[dom],
                "algorithmic_bytes_model": "DESIGN.md section 4: what THIS kernel has to move per launch "
                                           "(fp32 values, int32 indices)",
                "survey_8d_bytes": s8d,
                "survey_8d_note": "SURVEY.md 8d prices one layer fwd+bwd of the projected-row algorithm "
                                  "(every edge gathers a 4*H*C-byte projected row, twice in the backward); "
                                  + ("this step gathers the 4*Fs-byte input rows instead and never moves "
                                     "projected rows, so whole-step survey bytes / step time = "
                                     f"{s8d['total'] / (head['ms_per_step'] * 1e-3) / 1e9:.0f} GB/s is not an HBM rate"
                                     if path == "input-side aggregation" else
                                     "this step is that algorithm with the backward's second gather removed: "
                                     f"whole-step survey bytes / step time = "
                                     f"{s8d['total'] / (head['ms_per_step'] * 1e-3) / 1e9:.0f} GB/s")}
    res = {"metric": "GAT layer fwd+bwd edges/sec", "value": ep / (head["ms_per_step"] * 1e-3),
           "unit": "edges/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(head["ms_per_step"], 4), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": head["dtype"], "data": "synthetic",
           "config": {"workload": head["text"], "nodes": n, "edges": ep, "in_features": f, "heads": h,
                      "channels": c, "parallelism": "single GPU", "path": path,
                      "graph_layout_build_s": round(head["layout_s"], 4)},
           "roofline": roofline, "kernels": kernel_table(avg, kb)}
    if head["graph_ms"] is not None:
        res["hipgraph"] = {"ms_per_step": round(head["graph_ms"], 4), "value": ep / (head["graph_ms"] * 1e-3),
                           "unit": "edges/s", "kernel_ms_sum": round(sum(avg.values()), 4),
                           "note": "the same forward + backward replayed as one captured HIP graph; "
                                   "`value` above is the eager step (one host launch per kernel)"}

    # "report per layer and summed over the model's layers".  The headline is the
    # network's INPUT layer (x is data: no d_x, input-side aggregation).  The layers behind it take
    # the projected-row path and produce d_x; the 3-layer network of SURVEY.md 8 on this graph is
    # input layer + two same-width hidden layers.
    hidden_name = args.workload + "_hidden"
    if not args.no_extra_layers and hidden_name in synth.WORKLOADS:
        hid = measure_layer(hidden_name, args, dev)
        hdom = max(hid["kb"], key=lambda k: hid["avg"].get(k, 0.0))
        hach = hid["kb"][hdom] / (hid["avg"][hdom] * 1e-3) / 1e9
        htraffic, htsrc = measured_traffic(args.workload, hdom + ("_bf16" if hid["dtype"] == "bf16" else ""))
        res["hidden_layer"] = {
            "workload": hid["text"], "ms_per_step": round(hid["ms_per_step"], 4),
            "value": ep / (hid["ms_per_step"] * 1e-3), "unit": "edges/s", "path": "projected rows",
            "roofline": {"bound": "hbm", "kernel": hdom, "achieved": round(hach, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(hach / HBM_PEAK_GBS, 4),
                         "avg_ms": round(hid["avg"][hdom], 4), "algorithmic_bytes": hid["kb"][hdom],
                         "traffic": htraffic, "traffic_source": htsrc,
                         "survey_8d_bytes": survey_8d_bytes(n, ep, hid["h"], hid["c"])},
            "kernels": kernel_table(hid["avg"], hid["kb"])}
        if args.dtype == "f32":
            # the same hidden layer with bf16 feature rows and the bf16 projection: reported NEXT TO
            # the f32 figures, never in their place (tolerance stated in tests/test_gpu_bf16.py)
            h16 = measure_layer(hidden_name, args, dev, dtype="bf16")
            if h16["dtype"] == "bf16":
                d16 = max(h16["kb"], key=lambda k: h16["avg"].get(k, 0.0))
                a16 = h16["kb"][d16] / (h16["avg"][d16] * 1e-3) / 1e9
                t16, t16src = measured_traffic(args.workload, d16 + "_bf16")
                res["hidden_layer_bf16"] = {
                    "dtype": "bf16", "workload": h16["text"], "ms_per_step": round(h16["ms_per_step"], 4),
                    "value": ep / (h16["ms_per_step"] * 1e-3), "unit": "edges/s",
                    "roofline": {"bound": "hbm", "kernel": d16, "achieved": round(a16, 1),
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a16 / HBM_PEAK_GBS, 4),
                                 "avg_ms": round(h16["avg"][d16], 4), "algorithmic_bytes": h16["kb"][d16],
                                 "traffic": t16, "traffic_source": t16src},
                    "kernels": kernel_table(h16["avg"], h16["kb"]),
                    "tolerance": "out 1e-2 max-norm, d_x / d_W 5e-2 relative L2 against the f32 oracle "
                                 "(tests/test_gpu_bf16.py)"}
                res["model_sum_bf16_hidden"] = {
                    "ms_per_step": round(head["ms_per_step"] + 2 * h16["ms_per_step"], 4),
                    "edges_per_s_per_layer": 3 * ep / ((head["ms_per_step"] + 2 * h16["ms_per_step"]) * 1e-3),
                    "note": "f32 input layer + two bf16 hidden layers"}
            # and the input layer itself on bf16 source rows (gatmi_inagg_pack16): same rule
            i16 = measure_layer(args.workload, args, dev, dtype="bf16")
            if i16["dtype"] == "bf16" and "inagg_pack16" in i16["avg"]:
                di = max(i16["kb"], key=lambda k: i16["avg"].get(k, 0.0))
                ai = i16["kb"][di] / (i16["avg"][di] * 1e-3) / 1e9
                ti, tisrc = measured_traffic(args.workload, di + "_bf16")
                res["input_layer_bf16"] = {
                    "dtype": "bf16", "workload": i16["text"], "ms_per_step": round(i16["ms_per_step"], 4),
                    "value": ep / (i16["ms_per_step"] * 1e-3), "unit": "edges/s",
                    "path": "input-side aggregation on bf16 copies of the packed x rows (256-byte rows: two "
                            "128-byte lines per edge instead of four); logits, softmax, Z, q and outputs f32",
                    "roofline": {"bound": "hbm", "kernel": di, "achieved": round(ai, 1), "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": round(ai / HBM_PEAK_GBS, 4),
                                 "avg_ms": round(i16["avg"][di], 4), "algorithmic_bytes": i16["kb"][di],
                                 "traffic": ti, "traffic_source": tisrc},
                    "kernels": kernel_table(i16["avg"], i16["kb"]),
                    "tolerance": "out 1e-2 max-norm, d_W 5e-2 relative L2 against the f32 oracle; the f32 bar "
                                 "when x is bf16-exact (tests/test_gpu_bf16.py)"}
                if "hidden_layer_bf16" in res:
                    tot16 = i16["ms_per_step"] + 2 * res["hidden_layer_bf16"]["ms_per_step"]
                    res["model_sum_bf16"] = {"ms_per_step": round(tot16, 4),
                                             "edges_per_s_per_layer": 3 * ep / (tot16 * 1e-3),
                                             "note": "bf16 input layer + two bf16 hidden layers"}
        total_ms = head["ms_per_step"] + 2 * hid["ms_per_step"]
        res["model_sum"] = {
            "layers": [f"{f}->{h}x{c} (input, no d_x)", f"{hid['f']}->{hid['h']}x{hid['c']} (hidden, d_x)",
                       f"{hid['f']}->{hid['h']}x{hid['c']} (hidden, d_x; timed once, counted twice)"],
            "ms_per_step": round(total_ms, 4), "edges_per_s_per_layer": 3 * ep / (total_ms * 1e-3),
            "note": "sum of the per-layer fwd+bwd times above; value = 3 E' / that sum"}
    if not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(args.workload, args.seed)
    print(json.dumps(res))


def kernel_table(avg, kb):
    return {k: {"avg_ms": round(v, 4),
                **({"alg_GBps": round(kb[k] / (v * 1e-3) / 1e9, 1)} if k in kb else {})}
            for k, v in sorted(avg.items())}


def survey_8d_bytes(n, ep, h, c):
    """SURVEY.md 8d's B_fwd / B_bwd / B_alg for one layer (projected-row algorithm, D = H*C)."""
    d = h * c
    fwd = ep * (4 * d + 4 * h + 4) + n * (4 * d + 12 * h + 4)
    bwd = ep * (8 * d + 16 * h + 12) + n * (8 * d + 16 * h)
    return {"fwd": fwd, "bwd": bwd, "total": fwd + bwd}


def measure_layer(workload, args, dev, dtype=None):
    """Time `args.steps` forward+backward steps of one GAT layer on the named synthetic graph
    (inputs resident in HBM, layout built beforehand) and collect the library's per-kernel HIP-event
    timings over the same steps."""
    import gc

    from gat_amd import GATConv, _lib, synth
    from gat_amd.graph import cached_graph
    ei, n, f, h, c = synth.workload_graph(workload, args.seed, dev)
    d = h * c
    g = torch.Generator(device=dev).manual_seed(args.seed)
    x = torch.randn(n, f, generator=g, device=dev)
    hidden = workload.endswith("_hidden")   # a hidden layer: x has a gradient, d_x is part of the step
    x.requires_grad_(hidden)
    torch.manual_seed(args.seed)
    dtype = dtype or args.dtype
    conv = GATConv(f, c, heads=h, dropout=args.dropout,
                   input_aggregation={"auto": None, "on": True, "off": False}[args.input_aggregation],
                   feature_dtype=torch.bfloat16 if dtype == "bf16" else None).to(dev).train()

    torch.cuda.synchronize()
    t0 = time.time()
    layout = cached_graph(ei, n)
    torch.cuda.synchronize()
    t_layout = time.time() - t0
    ep = layout.num_edges
    go = torch.randn(n, d, generator=g, device=dev)

    def step():
        conv.zero_grad(set_to_none=True)
        x.grad = None
        out = conv(x, ei)
        out.backward(go)

    for _ in range(args.warmup):
        step()
    _lib.prof_read()
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _lib.prof_enable(False)
    per = {}
    for tag, ms in _lib.prof_read():
        per.setdefault(tag, []).append(ms)
    avg = {k: sum(v) / len(v) for k, v in per.items()}
    graph_ms = None
    if getattr(args, "hipgraph", False):
        conv.device_seed = torch.tensor([args.seed + 1], dtype=torch.int64, device=dev)
        conv.zero_grad(set_to_none=True)
        x.grad = None
        for _ in range(2):
            step()
        conv.zero_grad(set_to_none=True)
        x.grad = None
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = conv(x, ei)
            out.backward(go)
        for _ in range(args.warmup):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            graph.replay()
        torch.cuda.synchronize()
        graph_ms = (time.perf_counter() - t0) / args.steps * 1e3
        del graph, out
    text = (f"{workload}-shaped synthetic graph: {n} nodes, {ei.size(1)} edges ({ep} after self-loop "
            f"normalisation), one GAT layer {f}->{h}x{c}{' (hidden layer: d_x computed)' if hidden else ''}, "
            f"attention dropout {args.dropout} (train mode)")
    ran_bf16 = "proj16_fwd_gemm" in avg or "inagg_pack16" in avg or "inagg_table16" in avg
    kb = kernel_bytes(n, ep, h, d, f, feat_bytes=2 if ran_bf16 else 4)
    edge_after = kb.pop("inagg_bwd_edge_after_products")
    if "inagg_q_dw_fused" in avg and h > 4 and ep <= 12 * n and not ran_bf16:
        # the row-per-group edge kernel (5..8 heads, low degree) takes S from the products pass and
        # leaves the Z rows alone (gat_amd/csrc/inagg.hip: gatmi_inagg_bwd)
        kb["inagg_bwd_edge"] = edge_after
    res = dict(nodes=n, edges=ep, f=f, h=h, c=c, avg=avg, dtype="bf16" if ran_bf16 else "f32",
               kb=kb,
               ms_per_step=elapsed / args.steps * 1e3, layout_s=t_layout, text=text, graph_ms=graph_ms)
    del conv, x, go, layout, ei
    from gat_amd.graph import clear_graph_cache
    clear_graph_cache()
    gc.collect()
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    main()
