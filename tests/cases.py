"""Seeded inputs shared by the oracle tests, the golden-vector generator and the GPU
parity tests: the graph families SURVEY.md 8c lists (kNN with duplicated rows, random
multigraph with pre-existing self-loops and an isolated node, star, chain) and the layer
shapes the reference instantiates (model.py:26,28,30: 8 heads x hidden, 1 x classes)."""
from __future__ import annotations

import numpy as np
import torch


def knn_graph(n=64, f=5, k=5, seed=0, duplicates=4):
    """edge_index exactly as /root/reference/gat/converter.py:7-15 builds it."""
    from sklearn.neighbors import kneighbors_graph
    rng = np.random.default_rng(seed)
    x = rng.random((n, f)).astype(np.float32)
    for d in range(duplicates):           # duplicated rows: the self hit may be a twin
        x[n - 1 - d] = x[d]
    a = kneighbors_graph(x, n_neighbors=k, mode="connectivity", include_self=True, n_jobs=1).tocoo()
    ei = np.vstack([a.row.astype(np.int64), a.col.astype(np.int64)])
    return torch.tensor(ei, dtype=torch.long), n


def random_multigraph(n=50, e=300, seed=0):
    """parallel edges, pre-existing self-loops, node n-1 isolated (no in/out edges)."""
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n - 1, (2, e), generator=g)
    ei[:, :10] = ei[0, :10]                # 10 self-loops
    ei[:, 10:20] = ei[:, 20:30]            # 10 duplicated edges
    return ei, n


def star_graph(n=201, seed=0):
    """every node sends to node 0 (in-degree n-1), node 0 sends to a few."""
    src = torch.arange(1, n)
    dst = torch.zeros(n - 1, dtype=torch.long)
    back = torch.stack([torch.zeros(5, dtype=torch.long), torch.arange(1, 6)])
    return torch.cat([torch.stack([src, dst]), back], dim=1), n


def hub_source_graph(n=300, seed=0):
    """node 0 sends to everyone (out-degree n-1): long rows in the transposed CSR."""
    ei, _ = star_graph(n, seed)
    return ei.flip(0).contiguous(), n


def chain_graph(n=40):
    src = torch.arange(0, n - 1)
    return torch.stack([src, src + 1]), n


def empty_graph(n=7):
    return torch.zeros(2, 0, dtype=torch.long), n


GRAPHS = {
    "knn": knn_graph,
    "multigraph": random_multigraph,
    "star": star_graph,
    "hub_source": hub_source_graph,
    "chain": chain_graph,
    "empty": empty_graph,
}

# (in_features F, heads H, channels C)
SHAPES = {
    "conv1_h30": (25, 8, 30),     # run.py default hidden_dim=30 (run.py:20)
    "conv3_cls2": (240, 1, 2),    # model.py:30, 2 classes
    "cora_l1": (33, 8, 8),
    "cora_l2": (64, 1, 7),
    "arxiv_h64": (16, 4, 64),
    "odd_c40": (12, 4, 40),
    "wide_t2": (9, 8, 40),
    "three_heads": (6, 3, 5),
    "hidden_8x32": (40, 8, 32),   # model.py:28 at hidden_dim 32: full-wave rows of eight heads
    # input layers narrower than their output (the input-side aggregation path)
    "products_l1": (100, 4, 64),  # BASELINE.json config 5
    "arxiv_l1": (128, 4, 64),     # config 3, layer 1
    "in_t4": (200, 4, 64),
    "in_one_head": (5, 1, 16),
    "in_odd": (7, 5, 9),
}


def layer_inputs(n, f, h, c, seed=0):
    """x ~ N(0,1); lin.weight / att glorot (GATConv.reset_parameters); bias random so that
    the bias path is exercised; upstream gradient ~ N(0,1)."""
    import math
    g = torch.Generator().manual_seed(1000 + seed)
    x = torch.randn(n, f, generator=g)
    bw = math.sqrt(6.0 / (h * c + f))
    ba = math.sqrt(6.0 / (h + c))
    w = (torch.rand(h * c, f, generator=g) * 2 - 1) * bw
    a_s = (torch.rand(1, h, c, generator=g) * 2 - 1) * ba
    a_d = (torch.rand(1, h, c, generator=g) * 2 - 1) * ba
    b = torch.randn(h * c, generator=g) * 0.1
    go = torch.randn(n, h * c, generator=g)
    return x, w, a_s, a_d, b, go


def keep_mask(num_edges, h, p, seed=0):
    g = torch.Generator().manual_seed(2000 + seed)
    return torch.rand(num_edges, h, generator=g) >= p
