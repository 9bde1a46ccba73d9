"""CPU tests of the oracle itself: the two statements of the layer agree, the numpy and
plain-C graph layouts agree bit for bit, and the committed golden vectors are what the
oracle produces today."""
import glob
import os

import numpy as np
import pytest
import torch

import cases
from oracle import gat_oracle as O
from oracle.csr_ref import build_csr_c

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
CSR_FIELDS = ["rowptr", "col", "perm", "rowptr_t", "col_t", "pos_t"]


@pytest.mark.parametrize("gname", list(cases.GRAPHS))
@pytest.mark.parametrize("asl", [True, False])
def test_csr_numpy_equals_c(gname, asl):
    ei, n = cases.GRAPHS[gname]()
    a = O.build_csr(ei, n, add_self_loops=asl)
    b, norm = build_csr_c(ei.numpy(), n, add_self_loops=asl)
    assert a.num_edges == b.num_edges
    for f in CSR_FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    expect = O.normalize_edges(ei, n).numpy() if asl else ei.numpy()
    assert np.array_equal(norm, expect)


def test_csr_semantics():
    """kept edges keep their order, loops are appended last, nothing is coalesced."""
    ei = torch.tensor([[2, 0, 1, 1, 2, 2], [1, 0, 1, 2, 1, 1]])
    norm = O.normalize_edges(ei, 3)
    assert norm.tolist() == [[2, 1, 2, 2, 0, 1, 2], [1, 2, 1, 1, 0, 1, 2]]
    csr = O.build_csr(ei, 3)
    assert csr.rowptr.tolist() == [0, 1, 5, 7]
    assert csr.col.tolist() == [0, 2, 2, 2, 1, 1, 2]      # row 1: three 2->1 edges, then the loop
    assert csr.perm.tolist() == [4, 0, 2, 3, 5, 1, 6]
    # transpose round trip
    assert np.array_equal(csr.col[csr.pos_t], np.repeat(np.arange(3), np.diff(csr.rowptr_t)))


def test_out_of_range_rejected():
    with pytest.raises(ValueError):
        build_csr_c(np.array([[0, 5], [1, 0]]), 3)


@pytest.mark.parametrize("gname,sname,p", [("knn", "conv1_h30", 0.425), ("multigraph", "three_heads", 0.3),
                                           ("star", "cora_l1", 0.0), ("chain", "cora_l2", 0.5)])
def test_autograd_equals_closed_form(gname, sname, p):
    ei, n = cases.GRAPHS[gname]()
    f, h, c = cases.SHAPES[sname]
    x, w, a_s, a_d, b, go = [t.double() for t in cases.layer_inputs(n, f, h, c)]
    csr = O.build_csr(ei, n)
    mask = cases.keep_mask(csr.num_edges, h, p) if p > 0 else None
    _, ga = O.gatconv_grads_autograd(x, ei, w, a_s, a_d, b, h, c, go, dropout=p, mask=mask,
                                     training=p > 0)
    gc = O.gatconv_backward_closed_form(x, ei, w, a_s, a_d, h, c, go, dropout=p, mask=mask)
    for k in ga:
        torch.testing.assert_close(gc[k].view_as(ga[k]), ga[k], rtol=1e-10, atol=1e-11)


def test_softmax_rows_sum_to_one_and_isolated_node():
    ei, n = cases.random_multigraph()
    f, h, c = cases.SHAPES["three_heads"]
    x, w, a_s, a_d, b, _ = cases.layer_inputs(n, f, h, c)
    out, inter = O.gatconv_forward(x, ei, w, a_s, a_d, b, h, c, return_intermediates=True)
    dst = inter["edge_index"][1]
    sums = torch.zeros(n, h).index_add_(0, dst, inter["alpha"])
    torch.testing.assert_close(sums, torch.ones(n, h), rtol=1e-6, atol=1e-6)
    # the isolated node only sees its own self-loop: out = h_self + bias
    hh = (x @ w.t())[n - 1]
    torch.testing.assert_close(out[n - 1], hh + b, rtol=1e-6, atol=1e-6)


def test_eval_mode_ignores_dropout():
    ei, n = cases.chain_graph()
    f, h, c = cases.SHAPES["three_heads"]
    x, w, a_s, a_d, b, _ = cases.layer_inputs(n, f, h, c)
    a = O.gatconv_forward(x, ei, w, a_s, a_d, b, h, c, dropout=0.6, training=False)
    bb = O.gatconv_forward(x, ei, w, a_s, a_d, b, h, c, dropout=0.0)
    assert torch.equal(a, bb)


def test_module_state_dict_keys():
    m = O.OracleGATConv(5, 3, heads=2)
    assert sorted(m.state_dict()) == ["att_dst", "att_src", "bias", "lin.weight"]


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_golden_is_current(path):
    z = np.load(path)
    ei = torch.tensor(z["edge_index"])
    n, h, c, p = int(z["num_nodes"]), int(z["heads"]), int(z["channels"]), float(z["dropout"])
    t = lambda k: torch.tensor(z[k])
    mask = t("mask") if p > 0 else None
    torch.set_num_threads(1)
    out, g = O.gatconv_grads_autograd(t("x"), ei, t("weight"), t("att_src"), t("att_dst"),
                                      t("bias"), h, c, t("grad_out"), dropout=p, mask=mask,
                                      training=p > 0)
    torch.testing.assert_close(out, t("out"), rtol=1e-6, atol=1e-6)
    for k in ("x", "weight", "att_src", "att_dst", "bias"):
        torch.testing.assert_close(g[k], t("d_" + k), rtol=1e-5, atol=1e-5)
    csr, _ = build_csr_c(z["edge_index"], n)
    for f in CSR_FIELDS:
        assert np.array_equal(getattr(csr, f), z[f]), f


def test_golden_present():
    assert len(GOLDEN) >= 6
