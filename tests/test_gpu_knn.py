"""GPU parity of the kNN graph construction (gat_amd/csrc/knn.hip, C ABI gatmi_knn_graph) with the
reference's call, sklearn.neighbors.kneighbors_graph(X, k, mode="connectivity",
include_self=True) -> COO -> edge_index (/root/reference/gat/converter.py:7-15).  scikit-learn is
the library the reference calls and is present in this image, so the expected output is the
real call's.  Integer output: bit exact.  Exact duplicates are the documented exception
(scikit-learn's order among equal distances follows its heap, ours the lower index): there the
neighbour DISTANCES must agree, and the sets wherever no tie straddles the k-th place."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def sklearn_edge_index(x, k):
    from sklearn.neighbors import kneighbors_graph
    a = kneighbors_graph(x, n_neighbors=k, mode="connectivity", include_self=True, n_jobs=4).tocoo()
    return np.vstack([a.row.astype(np.int64), a.col.astype(np.int64)])


@pytest.mark.parametrize("n,f,k", [(64, 25, 5), (1000, 25, 5), (5000, 25, 5), (20000, 25, 5), (777, 6, 5),
                                   (300, 40, 3), (513, 64, 8), (257, 1, 2), (100, 25, 16), (5, 25, 5)])
def test_knn_graph_equals_sklearn(cuda, n, f, k):
    from gat_amd.data import knn_graph_hip
    rng = np.random.default_rng(n * 7 + f)
    x = rng.random((n, f))                   # features are min-max normalised to [0, 1] (encoder.py)
    want = sklearn_edge_index(x, k)
    got = knn_graph_hip(x, k, device=cuda)
    assert got.dtype == torch.int64 and tuple(got.shape) == (2, k * n) and got.is_cuda
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_knn_graph_float32_features_and_dataframe(cuda):
    """X.values of a float32 frame: scikit-learn ranks float32 input in its own precision path;
    the neighbour SETS must agree (order too, away from near-ties, which random data has none of)."""
    import pandas as pd
    from gat_amd.data import create_knn_graph
    rng = np.random.default_rng(3)
    x = pd.DataFrame(rng.random((1500, 25)).astype(np.float32))
    want = sklearn_edge_index(x.values, 5)
    got = create_knn_graph(x, backend="hip")
    assert not got.is_cuda                   # the reference returns a CPU tensor
    np.testing.assert_array_equal(got.numpy(), want)


def test_knn_graph_with_duplicate_rows(cuda):
    from gat_amd.data import knn_graph_hip
    rng = np.random.default_rng(11)
    x = rng.random((400, 25))
    x[50:60] = x[40:50]                       # ten exact duplicates
    x[200] = x[7]
    k = 5
    want = sklearn_edge_index(x, k)
    got = knn_graph_hip(x, k, device=cuda).cpu().numpy()
    np.testing.assert_array_equal(got[0], want[0])
    dist = lambda ei: np.linalg.norm(x[ei[0]] - x[ei[1]], axis=1).reshape(-1, k)
    np.testing.assert_allclose(dist(got), dist(want), rtol=0, atol=1e-12)
    # rows without a duplicate among their neighbours are identical
    same = (np.sort(got[1].reshape(-1, k), 1) == np.sort(want[1].reshape(-1, k), 1)).all(1)
    assert same.mean() > 0.9
    # every point is its own neighbour or has a duplicate standing in at distance 0
    assert (dist(got)[:, 0] == 0).all()


def test_auto_backend_is_bit_exact_on_duplicate_rows_too(cuda):
    """Integer output, bar = bit exact: the default backend flags the queries whose neighbour
    order is not strict and, when there are any, returns scikit-learn's own graph -- the whole
    edge_index equals the reference's call, duplicates or not.  Without ties it is the GPU result."""
    from gat_amd import _lib
    from gat_amd.data import create_knn_graph, knn_graph_hip
    rng = np.random.default_rng(11)
    x = rng.random((400, 25))
    clean = x.copy()
    x[50:60] = x[40:50]                       # ten exact duplicates
    x[200] = x[7]
    for k in (5, 8, 16):
        _lib.prof_read(); _lib.prof_enable(True)
        got = create_knn_graph(x, k)          # backend="auto"
        _lib.prof_enable(False)
        assert "knn_graph" in {t for t, _ in _lib.prof_read()}
        np.testing.assert_array_equal(got.numpy(), sklearn_edge_index(x, k))
    # which rows are flagged: exactly those with a tie among their k + 1 nearest
    ei, ties = knn_graph_hip(x, 5, device=cuda, return_ties=True)
    d = np.sqrt(((x[:, None, :] - x[None, :, :]) ** 2).sum(-1))
    near = np.sort(d, axis=1)[:, :6]
    want = (np.diff(near, axis=1) <= 1e-12).any(1)
    assert want.sum() >= 22                    # the 22 duplicated points at least
    np.testing.assert_array_equal(ties.cpu().numpy(), want)
    # no duplicates: nothing flagged, and auto IS the GPU result
    ei, ties = knn_graph_hip(clean, 5, device=cuda, return_ties=True)
    assert not bool(ties.any())
    np.testing.assert_array_equal(create_knn_graph(clean, 5).numpy(), ei.cpu().numpy())
    np.testing.assert_array_equal(ei.cpu().numpy(), sklearn_edge_index(clean, 5))
    # a near-tie (two candidates 1e-14 apart) is flagged as well
    y = clean.copy()
    y[300] = y[20] * (1 + 1e-15) + 1e-16
    _, ties = knn_graph_hip(y, 5, device=cuda, return_ties=True)
    assert bool(ties[20]) and bool(ties[300])


def test_knn_graph_rejects_bad_arguments(cuda):
    from gat_amd.data import knn_graph_hip
    x = np.random.default_rng(0).random((4, 3))
    with pytest.raises(ValueError):
        knn_graph_hip(x, 5, device=cuda)
    with pytest.raises(RuntimeError):
        knn_graph_hip(np.zeros((10, 65)), 2, device=cuda)
    with pytest.raises(RuntimeError):
        knn_graph_hip(np.zeros((100, 4)), 17, device=cuda)


def test_knn_graph_feeds_the_layer(cuda):
    """The edge_index built on the GPU goes straight into GATConv (no host round trip)."""
    from gat_amd import GATConv
    from gat_amd.data import knn_graph_hip
    rng = np.random.default_rng(5)
    x = rng.random((2000, 25))
    ei = knn_graph_hip(x, 5, device=cuda)
    conv = GATConv(25, 30, heads=8).to(cuda)
    out = conv(torch.tensor(x, dtype=torch.float32, device=cuda), ei)
    assert tuple(out.shape) == (2000, 240) and bool(torch.isfinite(out).all())
