"""Graph container and construction on the host, mirroring what the reference feeds its
model: Data(x, edge_index, y) from /root/reference/gat/converter.py:18-24 with the kNN
graph of :7-15 (k=5, include_self=True): on the GPU by gat_amd/csrc/knn.hip (SURVEY.md 8f
rank 2), or by the reference's own scikit-learn call on the host."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch


class Data:
    """x float32 [N, F], edge_index int64 [2, E] (row 0 = source, row 1 = destination),
    y int64 [N]; num_features == x.size(1) as torch_geometric.data.Data reports it."""

    def __init__(self, x, edge_index, y=None):
        self.x, self.edge_index, self.y = x, edge_index, y

    @property
    def num_features(self) -> int:
        return int(self.x.size(1))

    @property
    def num_nodes(self) -> int:
        return int(self.x.size(0))

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device)
        return Data(mv(self.x), mv(self.edge_index), mv(self.y))


def _values(X) -> np.ndarray:
    return X.values if hasattr(X, "values") else np.asarray(X)


def knn_graph_hip(X, k: int = 5, device=None, return_ties: bool = False):
    """create_knn_graph on the GPU (gat_amd/csrc/knn.hip): exact fp64 brute force, same
    edge_index layout as the scikit-learn call of /root/reference/gat/converter.py:8-15.
    Returns an int64 [2, k*N] tensor on `device` (default: the current device); with
    `return_ties` also a bool [N] tensor marking the queries whose neighbour order is not
    strict (equal or nearly equal distances among the k + 1 nearest: their order in
    scikit-learn's result is implementation-defined, here the lower index comes first)."""
    from . import _lib
    if not torch.cuda.is_available():
        raise RuntimeError("knn_graph_hip needs a GPU")
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if isinstance(X, torch.Tensor):
        x = X.to(device=dev, dtype=torch.float64).contiguous()
    else:
        x = torch.as_tensor(np.ascontiguousarray(_values(X), dtype=np.float64)).to(dev)
    if x.dim() != 2:
        raise ValueError(f"X must be [N, F], got {tuple(x.shape)}")
    n, f = x.shape
    if k > n:   # the message scikit-learn gives
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {k}, n_samples_fit = {n}")
    lib = _lib.lib()
    with torch.cuda.device(dev):
        ei = torch.empty(2, k * n, dtype=torch.int64, device=dev)
        ties = torch.zeros(n, dtype=torch.uint8, device=dev) if return_ties else None
        ws = torch.empty(int(lib.gatmi_knn_workspace_bytes(n)), dtype=torch.uint8, device=dev)
        _lib.check(lib.gatmi_knn_graph_ties(x.data_ptr(), n, f, k, ei.data_ptr(),
                                            None if ties is None else ties.data_ptr(), ws.data_ptr(),
                                            ws.numel(), torch.cuda.current_stream(dev).cuda_stream),
                   "gatmi_knn_graph_ties")
    return (ei, ties.bool()) if return_ties else ei


def _knn_graph_sklearn(values, k: int) -> torch.Tensor:
    from sklearn.neighbors import kneighbors_graph
    adj = kneighbors_graph(values, n_neighbors=k, mode="connectivity", include_self=True,
                           n_jobs=-1).tocoo()
    pairs = np.vstack([adj.row.astype(np.int64), adj.col.astype(np.int64)])
    return torch.tensor(pairs, dtype=torch.long)


def create_knn_graph(X, k: int = 5, backend: str = "auto", device=None) -> torch.Tensor:
    """/root/reference/gat/converter.py:7-15; returns a CPU int64 [2, k*N] tensor like the
    reference.  backend "sklearn": the reference's own scikit-learn call on the host.  "hip":
    computed on the GPU; equal distances resolve towards the lower index (scikit-learn's order
    among them is implementation-defined).  "auto": the GPU result when a GPU is present, the
    shape is covered (F <= 64, k <= 16) AND no query has tied neighbours -- then it is bit for bit
    what scikit-learn returns; with ties (duplicate rows) the whole graph comes from the
    scikit-learn call, so "auto" always equals the reference's output."""
    values = _values(X)
    if backend == "auto":
        if not (torch.cuda.is_available() and values.ndim == 2 and values.shape[1] <= 64 and k <= 16):
            backend = "sklearn"
        else:
            ei, ties = knn_graph_hip(values, k, device, return_ties=True)
            tied = int(ties.sum())
            if tied == 0:
                return ei.cpu()
            print(f"[gat_amd] kNN graph: {tied} of {ties.numel()} rows have neighbours at equal distance "
                  "(duplicate rows); taking scikit-learn's result so that tie order matches the reference")
            backend = "sklearn"
    if backend == "hip":
        return knn_graph_hip(values, k, device).cpu()
    if backend != "sklearn":
        raise ValueError(f"unknown backend {backend!r}")
    return _knn_graph_sklearn(values, k)


def convert_to_graph(X, y) -> Data:
    xv = X.values if hasattr(X, "values") else np.asarray(X)
    yv = y.values if hasattr(y, "values") else np.asarray(y)
    return Data(x=torch.tensor(xv, dtype=torch.float), edge_index=create_knn_graph(X),
                y=torch.tensor(yv, dtype=torch.long))


@dataclass
class Metrics:
    """Per-epoch curves train_model returns (/root/reference/gat/data_models.py:5-17)."""
    epoch_values: List[int] = field(default_factory=list)
    train_loss: List[float] = field(default_factory=list)
    val_loss: List[float] = field(default_factory=list)
    train_accuracy: List[float] = field(default_factory=list)
    val_accuracy: List[float] = field(default_factory=list)
    train_precision: List[float] = field(default_factory=list)
    train_recall: List[float] = field(default_factory=list)
    train_f1: List[float] = field(default_factory=list)
    val_precision: List[float] = field(default_factory=list)
    val_recall: List[float] = field(default_factory=list)
    val_f1: List[float] = field(default_factory=list)


def synthetic_traces(num_rows: int = 20000, num_features: int = 25, seed: int = 0):
    """Stand-in for the absent data/traces.csv (/root/reference/.gitignore:33): features in
    [0, 1] like the reference's min-max-normalised columns (encoder.py:19,37,49), binary
    label correlated with a few of them so that training has something to learn."""
    import pandas as pd
    rng = np.random.default_rng(seed)
    x = rng.random((num_rows, num_features)).astype(np.float32)
    score = 2.5 * x[:, 0] - 1.5 * x[:, 3] + x[:, 7] * x[:, 11] + 0.3 * rng.standard_normal(num_rows)
    y = (score > np.quantile(score, 0.8)).astype(np.int64)
    cols = [f"f{i}" for i in range(num_features)]
    return pd.DataFrame(x, columns=cols), pd.Series(y, name="is_anomaly")
