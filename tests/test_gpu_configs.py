"""BASELINE.json's configurations at their full shapes, against the oracle.

* config 2, Cora-shaped: 2,708 nodes / 10,556 edges / 1,433 features, layers 1433 -> 8x8 and
  64 -> 1x7 (/root/reference/gat/model.py:40,43,46 with those widths): every layer's output and
  every gradient, with and without attention dropout, and the two layers chained under a loss.
* config 5 at 1/10 scale (the largest the fp64 oracle does in well under a minute): BOTH paths of
  the layer -- projected rows and input-side aggregation -- measured against the fp64 oracle next
  to the fp32 oracle's own distance from it, plus how many LeakyReLU slopes each fp32 computation
  takes differently from fp64 (logits within rounding of zero).
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

import cases
from oracle import gat_oracle as O
from test_gpu_parity import close, run_layer

pytestmark = pytest.mark.gpu


def cora_graph():
    from gat_amd import synth
    ei, n, f, h, c = synth.workload_graph("cora", 0, "cpu")
    assert (n, ei.size(1), f, h, c) == (2708, 10556, 1433, 8, 8)
    return ei, n


@pytest.mark.parametrize("p", [0.0, 0.6], ids=["eval", "dropout"])
@pytest.mark.parametrize("f,h,c", [(1433, 8, 8), (64, 1, 7)], ids=["layer1_1433_to_8x8", "layer2_64_to_1x7"])
def test_cora_shaped_layers(cuda, f, h, c, p):
    ei, n = cora_graph()
    inputs = cases.layer_inputs(n, f, h, c, seed=11)
    x, w, a_s, a_d, b, go = inputs
    ne = O.build_csr(ei, n).num_edges
    mask = cases.keep_mask(ne, h, p, seed=3) if p > 0 else None
    ref_out, ref_g = O.gatconv_grads_autograd(x, ei, w, a_s, a_d, b, h, c, go, dropout=p, mask=mask,
                                              training=p > 0)
    out, g = run_layer(cuda, ei, n, f, h, c, p, mask, 512, inputs)
    close(out, ref_out, "out")
    summand = float(go.abs().max() * (x @ w.t()).abs().max())
    for k in ref_g:
        close(g[k], ref_g[k], "d_" + k, floor=summand if k.startswith("att") else 0.0)
    # the way the reference calls its first layer (model.py:40): x is data, no gradient
    if f == 1433:
        from gat_amd import GATConv
        conv = GATConv(f, c, heads=h, dropout=p).to(cuda).train(p > 0)
        with torch.no_grad():
            conv.lin.weight.copy_(w); conv.att_src.copy_(a_s); conv.att_dst.copy_(a_d); conv.bias.copy_(b)
        conv.injected_mask = mask
        o2 = conv(x.to(cuda), ei.to(cuda))
        o2.backward(go.to(cuda))
        close(o2, ref_out, "out (x without gradient)")
        close(conv.lin.weight.grad, ref_g["weight"], "d_weight (x without gradient)")


def test_cora_shaped_two_layer_network(cuda):
    """conv(1433 -> 8x8) -> ELU -> conv(64 -> 1x7) -> log_softmax -> nll_loss, attention dropout
    masks injected into both layers: loss and the gradient of every parameter of both layers."""
    from gat_amd import GATConv
    ei, n = cora_graph()
    p = 0.6
    ne = O.build_csr(ei, n).num_edges
    m1, m2 = cases.keep_mask(ne, 8, p, seed=5), cases.keep_mask(ne, 1, p, seed=6)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(n, 1433, generator=g)
    y = torch.randint(0, 7, (n,), generator=g)
    torch.manual_seed(4)
    r1, r2 = O.OracleGATConv(1433, 8, heads=8, dropout=p), O.OracleGATConv(64, 7, heads=1, dropout=p)
    with torch.no_grad():
        r1.bias.normal_(0, 0.1); r2.bias.normal_(0, 0.1)
    c1, c2 = GATConv(1433, 8, heads=8, dropout=p), GATConv(64, 7, heads=1, dropout=p)
    c1.load_state_dict(r1.state_dict()); c2.load_state_dict(r2.state_dict())
    c1, c2 = c1.to(cuda).train(), c2.to(cuda).train()
    for mod, m in ((r1, m1), (r2, m2), (c1, m1), (c2, m2)):
        mod.injected_mask = m
    r1.train(); r2.train()
    ref_logp = F.log_softmax(r2(F.elu(r1(x, ei)), ei), dim=1)
    ref_loss = F.nll_loss(ref_logp, y)
    ref_loss.backward()
    eid = ei.to(cuda)
    logp = F.log_softmax(c2(F.elu(c1(x.to(cuda), eid)), eid), dim=1)
    loss = F.nll_loss(logp, y.to(cuda))
    loss.backward()
    close(logp, ref_logp, "log-probabilities")
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 1e-5 * abs(float(ref_loss.detach()))
    for (name, pr), pg in zip(list(r1.named_parameters()) + list(r2.named_parameters()),
                              list(c1.parameters()) + list(c2.parameters())):
        # d_att_*: sums of cancelling terms, measured against one summand (as everywhere)
        floor = float(pr.grad.abs().max()) * 10 if "att" in name else 0.0
        close(pg.grad, pr.grad, "d_" + name, floor=floor)


# ------------------------------------------------------------------ config 5 at 1/10 scale
SCALE = 0.1


def _oracle_per_head(x, ei, w, a_s, a_d, b, h, c, go, dtype):
    """The oracle one head at a time (heads are independent: out is their concatenation, d_x
    their sum) -- bounds the [E', C] temporaries of the fp64 run to a few GB."""
    outs, gs, us = [], [], []
    xd = x.to(dtype)
    for hh in range(h):
        sl = slice(hh * c, (hh + 1) * c)
        o, g = O.gatconv_grads_autograd(xd, ei, w[sl].to(dtype), a_s[:, hh:hh + 1].to(dtype),
                                        a_d[:, hh:hh + 1].to(dtype), b[sl].to(dtype), 1, c,
                                        go[:, sl].to(dtype).contiguous())
        outs.append(o); gs.append(g)
    grads = dict(x=sum(g["x"] for g in gs), weight=torch.cat([g["weight"] for g in gs]),
                 att_src=torch.cat([g["att_src"] for g in gs], dim=1),
                 att_dst=torch.cat([g["att_dst"] for g in gs], dim=1),
                 bias=torch.cat([g["bias"] for g in gs]))
    return torch.cat(outs, dim=1), grads


def _logits(a_src, a_dst, src, dst):
    return a_src[src] + a_dst[dst]


def test_products_tenth_scale_both_paths_against_fp64(cuda, record_property):
    from gat_amd import _lib, ops, synth
    ei, n, f, h, c = synth.workload_graph("products", 0, "cpu", SCALE)
    x, w, a_s, a_d, b, go = cases.layer_inputs(n, f, h, c, seed=5)
    torch.set_num_threads(16)
    o64, g64 = _oracle_per_head(x, ei, w, a_s, a_d, b, h, c, go, torch.float64)
    o32, g32 = _oracle_per_head(x, ei, w, a_s, a_d, b, h, c, go, torch.float32)

    def err(a, ref):
        ref = ref.double()
        return float((a.detach().double().cpu().view_as(ref) - ref).abs().max() / ref.abs().max())

    inputs = (x, w, a_s, a_d, b, go)
    out_p, g_p = run_layer(cuda, ei, n, f, h, c, 0.0, None, 512, inputs)               # projected rows
    out_i, g_i = run_layer(cuda, ei, n, f, h, c, 0.0, None, 512, inputs, inagg=True)   # input-side

    # LeakyReLU slopes taken differently from the fp64 computation, per fp32 computation: the
    # logits u = a_src[j] + a_dst[i] of every (edge, head) from each computation's own tables
    nei = O.normalize_edges(ei, n)
    src, dst = nei[0], nei[1]
    h64 = (x.double() @ w.double().t()).view(n, h, c)
    u64 = _logits((h64 * a_s.double()).sum(-1), (h64 * a_d.double()).sum(-1), src, dst)
    h32 = (x @ w.t()).view(n, h, c)
    u_o32 = _logits((h32 * a_s).sum(-1), (h32 * a_d).sum(-1), src, dst)
    xd, wd = x.to(cuda), w.to(cuda)
    asd, add_ = a_s.view(-1).to(cuda), a_d.view(-1).to(cuda)
    hp, adst_p = ops.project_fwd(xd, wd, add_, h, c)
    asrc_p = (hp.view(n, h, -1)[:, :, :c] * a_s.to(cuda)).sum(-1)
    u_p = _logits(asrc_p.cpu(), adst_p.cpu(), src, dst)
    xs, adst_i = ops.inagg_pack(xd, wd, asd, add_, h, c)
    fp = (f + 3) // 4 * 4
    u_i = _logits(xs[:, fp:fp + h].cpu(), adst_i.cpu(), src, dst)
    flips = {k: int(((u > 0) != (u64 > 0)).sum()) for k, u in
             (("oracle_fp32", u_o32), ("projected_rows", u_p), ("input_side", u_i))}
    pairs = u64.numel()

    report = {"pairs": pairs, "slope_flips": flips}
    for label, out, g in (("projected_rows", out_p, g_p), ("input_side", out_i, g_i)):
        e_out, e_ref = err(out, o64), err(o32, o64)
        report[label] = {"out": (e_out, e_ref)}
        assert e_out <= 4 * e_ref + 1e-7, (label, "out", e_out, e_ref)
        for k in g:
            e, e_ref = err(g[k], g64[k]), err(g32[k], g64[k])
            report[label]["d_" + k] = (e, e_ref)
            # a flipped slope moves a gradient by ~1e-4 of its largest element whichever fp32
            # computation flips it, the oracle's included: the bound is the larger of 4x the fp32
            # oracle's own distance and one flip's worth per flip this path took
            assert e <= max(4 * e_ref, 2e-4 * max(flips[label], 1)) + 1e-7, (label, k, e, e_ref, flips)
    print("\n1/10-scale products layer vs fp64 (max-norm relative error: path, fp32 oracle):", report)
    record_property("products_tenth_scale", str(report))
    # no fp32 computation flips more than a handful of the 2.6e7 pairs
    assert max(flips.values()) <= 50, flips
