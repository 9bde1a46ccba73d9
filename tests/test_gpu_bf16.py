"""bf16 feature tables (gatmi_attn.feat_bf16, gatmi_project_*_bf16): tolerance STATED SEPARATELY
from the f32 path's, as BASELINE.json's north_star asks.

Two bars:
 (1) kernels: against the oracle with the SAME rounding points (projected rows stored as bf16, the
     gradient arriving at the aggregate rounded to bf16, operands of the projection rounded to bf16
     where the bf16 MFMA runs) the results agree to the f32 bar (max-norm 1e-5; 2e-5 where the
     rounded operands' products are re-associated) -- everything but the roundings is f32;
 (2) end to end, against the plain f32 oracle -- THE bf16 TOLERANCE: out within 1e-2 of its largest
     magnitude (bf16 keeps 8 bits: 4e-3 per rounding); gradients within 5e-2 in relative L2 norm
     for d_x and d_weight, 2.5e-1 for d_att_src / d_att_dst, and d_bias at the f32 bar (it never
     sees a rounded value).  Gradients are looser than the 4e-3 of a rounding because the logits
     move by ~1e-3 and the ones that land on the other side of zero take the other LeakyReLU slope
     (0.2 against 1): about one (edge, head) pair in a thousand changes its contribution five-fold
     -- measured on these cases with the oracle alone: 2.0e-2 / 2.2e-2 / 3.6e-2 / 1.1e-1 (L2), a
     single d_x element up to 1.8e-1 of the largest.  d_att_* are sums of cancelling terms.
"""
import ctypes as C

import pytest
import torch

import cases
from oracle import gat_oracle as O
from test_gpu_parity import close

pytestmark = pytest.mark.gpu

BF16_OUT_MAXNORM = 1e-2
BF16_GRAD_L2 = {"x": 5e-2, "weight": 5e-2, "att_src": 2.5e-1, "att_dst": 2.5e-1}


def rel(got, ref):
    ref = ref.detach().cpu().double()
    return float((got.detach().cpu().double().view_as(ref) - ref).abs().max() / ref.abs().max().clamp(min=1e-30))


def rel_l2(got, ref):
    ref = ref.detach().cpu().double()
    return float((got.detach().cpu().double().view_as(ref) - ref).norm() / ref.norm().clamp(min=1e-30))


def check_bf16_tolerance(out, grads, f32_out, f32_g):
    """bar (2) of the module docstring"""
    assert rel(out, f32_out) <= BF16_OUT_MAXNORM
    for k, tol in BF16_GRAD_L2.items():
        if k in grads:
            assert rel_l2(grads[k], f32_g[k]) <= tol, (k, rel_l2(grads[k], f32_g[k]))
    assert rel(grads["bias"], f32_g["bias"]) <= 1e-5


def graph(kind, n, seed):
    from gat_amd import synth
    if kind == "powerlaw":
        return synth.powerlaw_graph(n, 9 * n, max(n // 4, 8), seed=seed)
    if kind == "knn5":
        return synth.out_degree_k_graph(n, 5, seed=seed)
    return torch.stack([torch.arange(n), torch.zeros(n, dtype=torch.long)])       # star: one long row


@pytest.mark.parametrize("kind", ["powerlaw", "knn5", "star"])
@pytest.mark.parametrize("n,f,h,c,p", [(3001, 48, 4, 64, 0.0), (3001, 48, 4, 64, 0.4), (2049, 40, 8, 30, 0.3),
                                       (1500, 24, 8, 32, 0.0), (777, 16, 5, 40, 0.25), (900, 12, 2, 128, 0.0),
                                       (600, 12, 4, 200, 0.0)])
def test_attention_kernels_on_a_bf16_table(cuda, kind, n, f, h, c, p):
    """gatmi_attn_fwd / gatmi_attn_bwd through the C ABI on a bf16 table (here: the f32 projection
    rounded by torch), against the oracle with the same rounding points."""
    from gat_amd import _lib, ops
    from gat_amd.graph import build_graph
    assert ops.attn_bf16_supported(h, c)
    ei = graph(kind, n, seed=n + h)
    x, w, a_s, a_d, b, go = cases.layer_inputs(n, f, h, c, seed=n)
    ne = O.build_csr(ei, n).num_edges
    mask = cases.keep_mask(ne, h, p, seed=1) if p > 0 else None
    ref_out, ref_g = O.gatconv_grads_autograd(x, ei, w, a_s, a_d, b, h, c, go, dropout=p, mask=mask,
                                              training=p > 0, bf16_tables=True)
    layout = build_graph(ei.to(cuda), n, seg_len=64 if kind == "star" else 512)
    cp = (c + 3) // 4 * 4
    hfull = (x @ w.t()).to(torch.bfloat16)                       # the table, rounded as the oracle rounds it
    hp16 = torch.zeros(n, h, cp, dtype=torch.bfloat16)
    hp16[:, :, :c] = hfull.view(n, h, c)
    hp16 = hp16.view(n, h * cp).to(cuda)
    a_dst = (hfull.float().view(n, h, c) * a_d.view(1, h, c)).sum(-1).to(cuda)
    cfg = ops.AttnConfig(h, c, 0.2, p, _lib.MASK_BYTES if p > 0 else _lib.MASK_NONE,
                         mask=layout.edge_mask_to_csr(mask) if p > 0 else None)
    cfg.feat_bf16 = True
    asd, add_, bd = a_s.view(-1).to(cuda), a_d.view(-1).to(cuda), b.to(cuda)
    out, lse = ops.attn_fwd(layout, cfg, f, hp16, a_dst, asd, add_, bd, True)
    close(out, ref_out, "out")
    d_hp, d_as, d_ad, d_b, _ = ops.attn_bwd(layout, cfg, f, hp16, a_dst, asd, add_, bd, out, lse, go.to(cuda))
    d_h = d_hp.view(n, h, cp)[:, :, :c].reshape(n, h * c).cpu()
    summand = float(go.abs().max() * (x @ w.t()).abs().max())
    close(d_h @ w, ref_g["x"], "d_x")                            # the projection's backward, done here in torch
    close(d_h.t() @ x, ref_g["weight"], "d_weight")
    close(d_as, ref_g["att_src"], "d_att_src", floor=summand)
    close(d_ad, ref_g["att_dst"], "d_att_dst", floor=summand)
    close(d_b, ref_g["bias"], "d_bias")
    assert (d_hp.view(n, h, cp)[:, :, c:] == 0).all()
    # end to end against the plain f32 oracle: the separately stated bf16 tolerance
    f32_out, f32_g = O.gatconv_grads_autograd(x, ei, w, a_s, a_d, b, h, c, go, dropout=p, mask=mask,
                                              training=p > 0)
    check_bf16_tolerance(out, dict(x=d_h @ w, weight=d_h.t() @ x, att_src=d_as, att_dst=d_ad, bias=d_b),
                         f32_out, f32_g)


# the last five: the reference's search widths hidden_dim 28, 34, 36, 40 at 8 heads
# (/root/reference/bayesian_search.py:121, evolutionary_computing_search.py:213: heads on float4 but not
# 8-element boundaries, 288 and 320 columns in two column passes) and the 24-wide one
@pytest.mark.parametrize("n,f,h,c", [(4099, 256, 4, 64), (3000, 240, 8, 32), (1000, 64, 8, 8 * 4), (5001, 24, 4, 40),
                                     (33, 8, 8, 32), (2500, 136, 2, 128), (1, 16, 4, 64),
                                     (3001, 224, 8, 28), (2100, 272, 8, 34), (2049, 288, 8, 36),
                                     (3100, 320, 8, 40), (1500, 192, 8, 24)])
@pytest.mark.parametrize("dw_rows", [-1, 0])
def test_bf16_projection_gemms(cuda, n, f, h, c, dw_rows):
    """gatmi_project_fwd_bf16 / gatmi_project_bwd_bf16 through the C ABI against f64 products of
    the bf16-ROUNDED operands (what the MFMA multiplies), and, looser, of the unrounded ones.
    dw_rows = 0: d_W on the first bf16 kernel instead of the row-streaming one (GATMI_DW16_ROWS)."""
    from gat_amd import _lib
    _lib.set_option("GATMI_DW16_ROWS", dw_rows)
    try:
        _bf16_projection_gemms(cuda, n, f, h, c)
    finally:
        _lib.set_option("GATMI_DW16_ROWS", -1)


def _bf16_projection_gemms(cuda, n, f, h, c):
    from gat_amd import ops
    assert ops.project_bf16_supported(f, h, c)
    g = torch.Generator().manual_seed(n + f)
    x = torch.randn(n, f, generator=g)
    w = torch.randn(h * c, f, generator=g) / f ** 0.5
    att = torch.randn(h * c, generator=g)
    r = lambda t: t.to(torch.bfloat16).double()
    xd, wd, ad = x.to(cuda), w.to(cuda), att.to(cuda)
    cp = (c + 3) // 4 * 4           # head-padded rows: C = 34 lives in 36 columns per head
    hp16, a_dst = ops.project_fwd_bf16(xd, wd, ad, h, c)
    assert hp16.dtype == torch.bfloat16 and tuple(hp16.shape) == (n, h * cp)
    href = r(x) @ r(w).t()
    # one bf16 rounding of an f32-accumulated sum: half an ulp = 2^-9 of the value, plus the sum's own noise
    padded = hp16.double().cpu().view(n, h, cp)
    assert (padded[:, :, c:] == 0).all()
    got = padded[:, :, :c].reshape(n, h * c)
    assert float(((got - href).abs() - href.abs() * 2.0 ** -8).max()) <= 1e-5 * float(href.abs().max())
    close(a_dst, (got.view(n, h, c) * att.double().view(1, h, c)).sum(-1).float(), "a_dst")
    dh = torch.randn(n, h * c, generator=g)
    da = torch.randn(n, h, generator=g)
    dh_p = torch.zeros(n, h, cp)
    dh_p[:, :, :c] = dh.view(n, h, c)
    dh_p = dh_p.view(n, h * cp)
    for with_term in (False, True):
        full = dh.double() + (da.double().unsqueeze(-1) * att.double().view(1, h, c)).reshape(n, h * c) \
            if with_term else dh.double()
        dw, dx = ops.project_bwd_bf16(xd, wd, dh_p.to(cuda), h, c, True,
                                      da.to(cuda) if with_term else None, ad if with_term else None)
        fr = full.float().to(torch.bfloat16).double()             # the kernel rounds d_h + term, once
        close(dx, (fr @ r(w)).float(), "d_x" + (" + term" if with_term else ""))
        close(dw, (fr.t() @ r(x)).float(), "d_weight" + (" + term" if with_term else ""))
        assert rel_l2(dx, full @ w.double()) <= 1e-2 and rel_l2(dw, full.t() @ x.double()) <= 1e-2
    dw, none = ops.project_bwd_bf16(xd, wd, dh_p.to(cuda), h, c, False)
    assert none is None
    close(dw, (r(dh).t() @ r(x)).float(), "d_weight (no d_x)")


@pytest.mark.parametrize("kind", ["powerlaw", "knn5"])
@pytest.mark.parametrize("n,f,h,c,p", [(3001, 256, 4, 64, 0.0), (3001, 256, 4, 64, 0.4), (2049, 240, 8, 32, 0.3),
                                       (1200, 64, 2, 128, 0.0)])
def test_layer_in_bf16_mode(cuda, kind, n, f, h, c, p):
    """GATConv(feature_dtype=torch.bfloat16) with x carrying a gradient (a hidden layer): bf16
    projection, bf16 table, f32 everything else.  Bar (1): the oracle with the same rounding points
    (operands of the projection, the stored rows, the gradient at the aggregate); the backward
    GEMMs round d_h once more, which the oracle does not model -- hence 1e-2 in L2 for the
    gradients there and the f32 bar for the forward.  Bar (2): the stated bf16 tolerance."""
    from gat_amd import GATConv, _lib
    ei = graph(kind, n, seed=n + h)
    x, w, a_s, a_d, b, go = cases.layer_inputs(n, f, h, c, seed=n)
    ne = O.build_csr(ei, n).num_edges
    mask = cases.keep_mask(ne, h, p, seed=1) if p > 0 else None
    kw = dict(dropout=p, mask=mask, training=p > 0)
    ref_out, ref_g = O.gatconv_grads_autograd(x, ei, w, a_s, a_d, b, h, c, go, bf16_tables=True,
                                              bf16_matmul=True, **kw)
    f32_out, f32_g = O.gatconv_grads_autograd(x, ei, w, a_s, a_d, b, h, c, go, **kw)
    conv = GATConv(f, c, heads=h, dropout=p, feature_dtype=torch.bfloat16).to(cuda).train(p > 0)
    with torch.no_grad():
        conv.lin.weight.copy_(w); conv.att_src.copy_(a_s); conv.att_dst.copy_(a_d); conv.bias.copy_(b)
    conv.injected_mask = mask
    xd = x.to(cuda).requires_grad_(True)
    _lib.prof_read(); _lib.prof_enable(True)
    out = conv(xd, ei.to(cuda))
    out.backward(go.to(cuda))
    torch.cuda.synchronize(); _lib.prof_enable(False)
    tags = {t for t, _ in _lib.prof_read()}
    assert {"proj16_fwd_gemm", "proj16_bwd_w_gemm", "proj16_bwd_x_gemm", "attn_fwd"} <= tags, tags
    grads = dict(x=xd.grad, weight=conv.lin.weight.grad, att_src=conv.att_src.grad,
                 att_dst=conv.att_dst.grad, bias=conv.bias.grad)
    assert rel(out, ref_out) <= 2e-3          # a table entry within an ulp of a rounding boundary may round the other way
    assert rel_l2(out, ref_out) <= 2e-4
    for k in ("x", "weight"):
        assert rel_l2(grads[k], ref_g[k]) <= 1e-2, (k, rel_l2(grads[k], ref_g[k]))
    close(grads["bias"], ref_g["bias"], "d_bias")
    check_bf16_tolerance(out, grads, f32_out, f32_g)


@pytest.mark.parametrize("world,balance", [(2, "edges"), (3, "nodes")])
def test_partitioned_layer_with_bf16_exchange(cuda, world, balance):
    """exchange "h" with a bf16 table (what PartitionedGATConv(feature_dtype=torch.bfloat16) runs per
    rank): every rank projects its rows into bf16, the table is their concatenation (the all-gather,
    half the bytes), d_h comes back in f32 -- one process plays all ranks."""
    from gat_amd import ops, synth
    from gat_amd.dist import HipBackend, local_edges, make_partition
    n, f, h, c = 2500, 64, 4, 64
    ei = synth.powerlaw_graph(n, 8 * n, 500, seed=5)
    x, w, a_s, a_d, b, go = cases.layer_inputs(n, f, h, c, seed=6)
    ref_out, ref_g = O.gatconv_grads_autograd(x, ei, w, a_s, a_d, b, h, c, go, bf16_tables=True, bf16_matmul=True)
    be = HipBackend()
    cfg = ops.AttnConfig(h, c)
    cfg.feat_bf16 = True
    xd, wd, bd, god = x.to(cuda), w.to(cuda), b.to(cuda), go.to(cuda)
    asd, add_ = a_s.to(cuda).view(-1), a_d.to(cuda).view(-1)
    parts = [make_partition(ei, n, world, r, balance, padding="pad") for r in range(world)]   # the tables below are padded by hand
    shards = []
    for p in parts:
        hp, a_dst = be.project_fwd(xd[p.lo:p.hi].contiguous(), wd, add_, cfg)
        assert hp.dtype == torch.bfloat16
        pad = hp.new_zeros(p.block, hp.size(1)); pad[:p.n_local] = hp
        shards.append((pad, a_dst))
    hp_full = torch.cat([s_[0] for s_ in shards])
    d_hp_sum, outs, d_adst = None, [], []
    for p, (_, a_dst) in zip(parts, shards):
        lay = be.build_layout(local_edges(ei.to(cuda), p), p)
        out, lse = be.attn_fwd(lay, cfg, f, hp_full, a_dst, asd, add_, bd, True)
        outs.append(out)
        r_ = be.attn_bwd(lay, cfg, f, hp_full, a_dst, asd, add_, bd, out, lse, god[p.lo:p.hi].contiguous())
        d_hp_sum = r_[0] if d_hp_sum is None else d_hp_sum + r_[0]
        d_adst.append(r_[4])
    out = torch.cat(outs)
    assert rel(out, ref_out) <= 2e-3 and rel_l2(out, ref_out) <= 2e-4
    d_w, d_x = 0, []
    for p, dd in zip(parts, d_adst):
        blk = d_hp_sum[p.plo:p.phi].contiguous()
        be.add_dst_term(blk, dd, add_, cfg)
        dw, dx = be.project_bwd(xd[p.lo:p.hi].contiguous(), wd, blk, cfg, True)
        d_w = d_w + dw; d_x.append(dx)
    assert rel_l2(torch.cat(d_x), ref_g["x"]) <= 1e-2 and rel_l2(d_w, ref_g["weight"]) <= 1e-2


def test_bf16_mode_falls_back_to_f32_where_not_covered(cuda):
    from gat_amd import GATConv, _lib, ops
    assert not ops.project_bf16_supported(1433, 8, 8) and not ops.project_bf16_supported(25, 8, 30)
    ei, n = cases.knn_graph(n=300, f=6, seed=2)
    conv = GATConv(25, 30, heads=8, feature_dtype=torch.bfloat16).to(cuda).eval()
    x = torch.randn(n, 25, device=cuda, requires_grad=True)
    _lib.prof_read(); _lib.prof_enable(True)
    conv(x, ei.to(cuda)).sum().backward()
    torch.cuda.synchronize(); _lib.prof_enable(False)
    tags = {t for t, _ in _lib.prof_read()}
    assert "proj_fwd_gemm" in tags and "proj16_fwd_gemm" not in tags


def test_bf16_tables_are_refused_for_narrow_rows(cuda):
    from gat_amd import _lib, ops
    from gat_amd.graph import build_graph
    assert not ops.attn_bf16_supported(1, 7) and not ops.attn_bf16_supported(8, 8)
    ei, n = cases.chain_graph()
    layout = build_graph(ei.to(cuda), n)
    cfg = ops.AttnConfig(8, 8)
    cfg.feat_bf16 = True
    z = torch.zeros(n, 64, dtype=torch.bfloat16, device=cuda)
    v = torch.zeros(64, device=cuda)
    with pytest.raises(RuntimeError, match="bf16 feature tables need"):
        ops.attn_fwd(layout, cfg, 4, z, torch.zeros(n, 8, device=cuda), v, v, v, False)


@pytest.mark.parametrize("kind", ["powerlaw", "knn5", "star"])
@pytest.mark.parametrize("n,f,h,c,p", [(3001, 100, 4, 64, 0.0), (3001, 100, 4, 64, 0.4), (2049, 112, 2, 128, 0.3),
                                       (1500, 40, 3, 24, 0.0), (900, 108, 4, 40, 0.25), (2049, 100, 8, 32, 0.3),
                                       (1500, 48, 6, 20, 0.0)])
def test_input_side_path_on_bf16_source_rows(cuda, kind, n, f, h, c, p):
    """GATConv(feature_dtype=torch.bfloat16) on the input-side aggregation path (x is data: no d_x):
    the edge kernels gather a bf16 copy of the packed x rows (gatmi_inagg_pack16, 256-byte rows); the
    logits, the softmax, Z, q and every output stay f32.  Bar (1): fed an x that bf16 holds exactly,
    the table is exact and the layer meets the f32 bar against the oracle.  Bar (2): on any x, the
    stated bf16 tolerance against the f32 oracle."""
    from gat_amd import GATConv, _lib, ops
    assert ops.inagg_rows16_supported(f, h, c)
    ei = graph(kind, n, seed=n + h)
    x, w, a_s, a_d, b, go = cases.layer_inputs(n, f, h, c, seed=n)
    ne = O.build_csr(ei, n).num_edges
    mask = cases.keep_mask(ne, h, p, seed=1) if p > 0 else None
    kw = dict(dropout=p, mask=mask, training=p > 0)
    summand = float(go.abs().max() * (x @ w.t()).abs().max())
    for exact in (True, False):
        xin = x.bfloat16().float() if exact else x
        ref_out, ref_g = O.gatconv_grads_autograd(xin, ei, w, a_s, a_d, b, h, c, go, **kw)
        conv = GATConv(f, c, heads=h, dropout=p, input_aggregation=True,
                       feature_dtype=torch.bfloat16).to(cuda).train(p > 0)
        with torch.no_grad():
            conv.lin.weight.copy_(w); conv.att_src.copy_(a_s); conv.att_dst.copy_(a_d); conv.bias.copy_(b)
        conv.injected_mask = mask
        _lib.prof_read(); _lib.prof_enable(True)
        out = conv(xin.to(cuda), ei.to(cuda))
        out.backward(go.to(cuda))
        torch.cuda.synchronize(); _lib.prof_enable(False)
        tags = {t for t, _ in _lib.prof_read()}
        assert {"inagg_pack16", "inagg_fwd", "inagg_bwd_edge"} <= tags and "inagg_table16" not in tags, tags
        grads = dict(weight=conv.lin.weight.grad, att_src=conv.att_src.grad, att_dst=conv.att_dst.grad,
                     bias=conv.bias.grad)
        if exact:
            close(out, ref_out, "out")
            for k in grads:
                close(grads[k], ref_g[k], "d_" + k, floor=summand if k.startswith("att") else 0.0)
        else:
            check_bf16_tolerance(out, grads, ref_out, ref_g)


@pytest.mark.parametrize("n,f,h,c", [(1003, 100, 4, 64), (517, 64, 8, 32), (300, 40, 8, 16), (2050, 112, 3, 24),
                                     (65, 72, 2, 64), (1, 36, 4, 16), (4097, 57, 4, 64), (700, 33, 5, 8)])
def test_pack_kernel_writes_the_same_bf16_table_as_the_conversion_pass(cuda, n, f, h, c):
    """gatmi_inagg_pack16 against gatmi_inagg_pack + gatmi_inagg_table16: xs, a_dst and every byte of
    the 256-byte rows (x pieces, the zero padding behind them, the f32 logits of up to eight heads)."""
    from gat_amd import ops
    assert ops.inagg_rows16_supported(f, h, c)
    x, w, a_s, a_d, _, _ = cases.layer_inputs(n, f, h, c, seed=n + f)
    x, w, a_s, a_d = (t.to(cuda) for t in (x, w, a_s.view(-1), a_d.view(-1)))
    xs, a_dst = ops.inagg_pack(x, w, a_s, a_d, h, c)
    table = ops.inagg_table16(xs, f, h, c)
    xs2, a_dst2, table2 = ops.inagg_pack(x, w, a_s, a_d, h, c, rows16=True)
    assert torch.equal(xs, xs2) and torch.equal(a_dst, a_dst2)
    assert table2.shape == table.shape and torch.equal(table.view(torch.int16), table2.view(torch.int16))


def test_bf16_source_rows_are_refused_where_not_covered(cuda):
    from gat_amd import GATConv, _lib, ops
    assert not ops.inagg_rows16_supported(25, 8, 30) and not ops.inagg_rows16_supported(120, 4, 64)
    ei, n = cases.knn_graph(n=300, f=6, seed=2)
    conv = GATConv(25, 30, heads=8, input_aggregation=True, feature_dtype=torch.bfloat16).to(cuda).eval()
    _lib.prof_read(); _lib.prof_enable(True)
    conv(torch.randn(n, 25, device=cuda), ei.to(cuda))
    torch.cuda.synchronize(); _lib.prof_enable(False)
    assert not {"inagg_table16", "inagg_pack16"} & {t for t, _ in _lib.prof_read()}       # f32 rows, as before


def test_random_graphs_on_bf16_source_rows(cuda):
    """Seeded random draws (hub destinations and sources, duplicate edges, self-loops, cut rows, every
    head count 1..8 the bf16 rows cover, odd feature counts) with a bf16-exact x: the f32 bar."""
    import random
    from gat_amd import GATConv, ops
    rng = random.Random(4321)
    done = 0
    for trial in range(60):
        n = rng.choice([3, 17, 64, 65, 200, 333, 1000])
        e = rng.choice([0, n, 5 * n, 20 * n])
        h = rng.choice([1, 2, 3, 4, 5, 6, 7, 8])
        c = rng.choice([8, 16, 30, 33, 64])
        f = rng.choice([33, 36, 57, 64, 65, 100, 111, 112])
        if not ops.inagg_rows16_supported(f, h, c) or f >= h * ((c + 3) // 4 * 4):
            continue
        p = rng.choice([0.0, 0.3, 0.6])
        seg = rng.choice([64, 128, 512])
        g = torch.Generator().manual_seed(trial)
        ei = torch.randint(0, n, (2, e), generator=g)
        if e > 4:
            ei[1, : e // 3] = ei[1, 0]            # a hub destination
            ei[0, e // 3: e // 2] = ei[0, -1]      # a hub source
        x, w, a_s, a_d, b, go = cases.layer_inputs(n, f, h, c, seed=trial)
        x = x.bfloat16().float()
        ne = O.build_csr(ei, n).num_edges
        mask = cases.keep_mask(ne, h, p, seed=trial) if p > 0 else None
        ref_out, ref_g = O.gatconv_grads_autograd(x, ei, w, a_s, a_d, b, h, c, go, dropout=p, mask=mask,
                                                  training=p > 0)
        conv = GATConv(f, c, heads=h, dropout=p, seg_len=seg, input_aggregation=True,
                       feature_dtype=torch.bfloat16).to(cuda).train(p > 0)
        with torch.no_grad():
            conv.lin.weight.copy_(w); conv.att_src.copy_(a_s); conv.att_dst.copy_(a_d); conv.bias.copy_(b)
        conv.injected_mask = mask
        out = conv(x.to(cuda), ei.to(cuda))
        out.backward(go.to(cuda))
        summand = float(go.abs().max() * (x @ w.t()).abs().max())
        tag = f"trial {trial} n={n} e={e} {f}->{h}x{c} p={p} seg={seg}"
        close(out, ref_out, "out " + tag)
        close(conv.lin.weight.grad, ref_g["weight"], "d_weight " + tag)
        close(conv.att_src.grad, ref_g["att_src"], "d_att_src " + tag, floor=summand)
        close(conv.att_dst.grad, ref_g["att_dst"], "d_att_dst " + tag, floor=summand)
        close(conv.bias.grad, ref_g["bias"], "d_bias " + tag)
        done += 1
    assert done >= 15, done
