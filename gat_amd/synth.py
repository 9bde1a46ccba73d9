"""Synthetic graphs of the shapes BASELINE.json names: power-law
in-degree, uniform sources, no datasets needed.  Generated on whatever device the
generator lives on, deterministically from the seed."""
from __future__ import annotations

import math

import torch

# name -> (nodes, edges, in_features, heads, channels, max in-degree aimed for)
WORKLOADS = {
    "cora": (2708, 10556, 1433, 8, 8, 170),
    "arxiv": (169343, 1166243, 128, 4, 64, 13000),
    "products": (2449029, 62000000, 100, 4, 64, 17000),
    "runpy100k": (100000, 500000, 25, 8, 30, 0),
    "runpy10k": (10000, 50000, 25, 8, 30, 0),
    "runpy1m": (1000000, 5000000, 25, 8, 30, 0),
    # hidden layers of the 3-layer networks of SURVEY.md 8 (x carries a gradient: bench.py also
    # times d_x): 256 -> 4x64 on the arxiv- and products-shaped graphs
    "arxiv_hidden": (169343, 1166243, 256, 4, 64, 13000),
    "products_hidden": (2449029, 62000000, 256, 4, 64, 17000),
    # uniform-degree variants ("to separate skew effects"): same sizes, dst uniform
    "products_uniform": (2449029, 62000000, 100, 4, 64, 0),
    "arxiv_uniform": (169343, 1166243, 128, 4, 64, 0),
    "cora_uniform": (2708, 10556, 1433, 8, 8, 0),
}


def powerlaw_graph(num_nodes: int, num_edges: int, max_degree: int, seed: int = 0,
                   device="cpu") -> torch.Tensor:
    """edge_index int64 [2, E]: src uniform; dst = perm[floor(N * u**gamma)], gamma chosen
    so the heaviest destination collects about `max_degree` edges (0: uniform dst)."""
    g = torch.Generator(device=device).manual_seed(seed)
    src = torch.randint(0, num_nodes, (num_edges,), generator=g, device=device)
    if max_degree and max_degree * num_nodes > num_edges:
        inv_gamma = math.log(num_edges / max_degree) / math.log(num_nodes)
        gamma = 1.0 / max(inv_gamma, 1e-3)
        u = torch.rand(num_edges, generator=g, device=device, dtype=torch.float64)
        idx = (u.pow(gamma) * num_nodes).long().clamp_(max=num_nodes - 1)
        perm = torch.randperm(num_nodes, generator=g, device=device)
        dst = perm[idx]
    else:
        dst = torch.randint(0, num_nodes, (num_edges,), generator=g, device=device)
    return torch.stack([src, dst])


def out_degree_k_graph(num_nodes: int, k: int = 5, seed: int = 0, device="cpu") -> torch.Tensor:
    """The shape /root/reference/gat/converter.py:7-15 emits: every node sends to k
    others (first neighbour = itself, as include_self=True usually yields)."""
    g = torch.Generator(device=device).manual_seed(seed)
    src = torch.arange(num_nodes, device=device).repeat_interleave(k)
    dst = torch.randint(0, num_nodes, (num_nodes, k), generator=g, device=device)
    dst[:, 0] = torch.arange(num_nodes, device=device)
    return torch.stack([src, dst.reshape(-1)])


def workload_graph(name: str, seed: int = 0, device="cpu", scale: float = 1.0):
    n, e, f, h, c, dmax = WORKLOADS[name]
    n, e = max(int(n * scale), 8), max(int(e * scale), 8)
    if name.startswith("runpy"):
        return out_degree_k_graph(n, 5, seed, device), n, f, h, c
    return powerlaw_graph(n, e, int(dmax * max(scale, 0.05)) if dmax else 0, seed, device), n, f, h, c


def kernel_bytes(n, e, h, d, f, feat_bytes=4):
    """Algorithmic bytes per launch of the two heavy edge kernels (fp32 values, int32
    indices).  Forward is SURVEY.md 8d's B_fwd.  The backward gathers
    features once instead of SURVEY 8d's twice: one pass over the transposed CSR reads, per
    edge, the grad_out row (4D), the destination's 16H-byte record, col_t and pos_t, and
    writes d_u (4H); per node it reads h, writes d_h and d_a_src."""
    fp = (f + 3) // 4 * 4
    fs = fp + (h + 3) // 4 * 4        # packed source row: x | a_src
    rowb = feat_bytes * fp + 4 * (fs - fp)
    return {
        # feat_bytes = 2: bf16 feature tables -- the gathered rows (h in the forward, the bf16 copy
        # of grad_out in the backward) and the row-local h are half as wide; out, d_h, a_src stay f32
        "attn_fwd": e * (feat_bytes * d + 4 * h + 4) + n * (4 * d + 12 * h + 4),
        "attn_bwd_edge": e * (feat_bytes * d + 16 * h + 8 + 4 * h) + n * ((4 + feat_bytes) * d + 4 * h),
        # input-side aggregation (gat_amd/csrc/inagg.hip): an edge moves a packed source row
        # x | a_src (4 Fs bytes) and its col index; a node reads rowptr and a_dst, writes Z [H, Fp]
        # and lse (forward); reads its own packed row, q and Z [H, Fp], a_dst and lse (backward;
        # nothing is written per edge or per node)
        # feat_bytes = 2: the gathered row is the bf16 copy (2 Fp bytes of x + the f32 logits); the
        # row-local reads and Z, q stay f32
        "inagg_fwd": e * (rowb + 4) + n * (4 + 4 * h + 4 * h * fp + 4 * h),
        "inagg_bwd_edge": e * (rowb + 4) + n * (4 + 4 * fs + 8 * h * fp + 8 * h),
        # the fused tiles of the row-per-group layout (run.py's conv1): the forward also
        # writes the out row (4 D) -- the separate out = Z W^T product and its re-read of Z are gone --;
        # the backward reads the grad_out row (4 D) and Z once, q never exists in HBM
        "inagg_fwd_fused": e * (rowb + 4) + n * (4 + 4 * h + 4 * h * fp + 4 * h + 4 * d),
        "inagg_bwd_fused": e * (rowb + 4) + n * (4 + 4 * fs + 4 * d + 4 * h * fp + 8 * h),
        # beyond 200 k rows the backward's products run as one pass over grad_out: reads the grad_out row
        # and Z, writes q and S = <q, Z> per head; the edge kernel after it reads q and S, not Z
        "inagg_q_dw_fused": n * (4 * d + 8 * h * fp + 4 * h),
        "inagg_bwd_edge_after_products": e * (rowb + 4) + n * (4 + 4 * fs + 4 * h * fp + 12 * h),
    }
