"""Size-independent properties at BASELINE.json's FULL configuration (products-shaped: 2,449,029
nodes, 62 M edges, 100 -> 4 x 64), where the oracle cannot run: graph-layout invariants, softmax
weights summing to one, linearity of the backward, agreement of the two layer paths and of the
kernel variants, reproducibility of the hash dropout.  A few seconds on one MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full(cuda):
    from gat_amd import synth
    from gat_amd.graph import cached_graph
    ei, n, f, h, c = synth.workload_graph("products", 0, cuda)
    layout = cached_graph(ei, n)
    return ei, n, f, h, c, layout


def test_full_size_graph_layout_invariants(cuda, full):
    ei, n, f, h, c, layout = full
    self_loops = int((ei[0] == ei[1]).sum())
    assert layout.num_edges == ei.size(1) - self_loops + n            # remove_self_loops + add_self_loops
    rowptr = layout.rowptr.long()
    assert int(rowptr[0]) == 0 and int(rowptr[-1]) == layout.num_edges
    deg = rowptr[1:] - rowptr[:-1]
    assert int(deg.min()) >= 1                                         # every node has its loop
    want = torch.bincount(ei[1][ei[0] != ei[1]], minlength=n) + 1
    assert torch.equal(deg, want)
    # a checksum of the whole edge multiset survives the sort: sum of (src * 31 + dst) mod 2^61
    dst = torch.repeat_interleave(torch.arange(n, device=cuda), deg)
    got = int(((layout.col.long() * 31 + dst) % (1 << 40)).sum())
    keep = ei[0] != ei[1]
    loops = torch.arange(n, device=cuda)
    ref = int(((ei[0][keep] * 31 + ei[1][keep]) % (1 << 40)).sum()) + int(((loops * 31 + loops) % (1 << 40)).sum())
    assert got == ref


@pytest.mark.parametrize("inagg", [True, False])
def test_full_size_constant_rows_give_constant_output(cuda, full, inagg):
    from gat_amd import GATConv
    ei, n, f, h, c, _ = full
    torch.manual_seed(1)
    conv = GATConv(f, c, heads=h, input_aggregation=inagg).to(cuda).eval()
    with torch.no_grad():
        conv.bias.copy_(torch.randn(h * c, device=cuda))
    x = torch.randn(1, f, device=cuda).expand(n, f).contiguous()
    out = conv(x, ei)
    expect = (x[:1] @ conv.lin.weight.t()) + conv.bias
    err = float((out.detach() - expect.detach()).abs().max())
    assert err <= 2e-5 * float(expect.detach().abs().max()), err


def test_full_size_paths_and_kernel_variants_agree(cuda, full):
    """Projected rows against input-side aggregation, and the streaming forward against the
    two-edge register forward, on the same random inputs: forward and every parameter gradient."""
    from gat_amd import GATConv, _lib
    ei, n, f, h, c, _ = full
    g = torch.Generator(device=cuda).manual_seed(3)
    x = torch.randn(n, f, generator=g, device=cuda)
    go = torch.randn(n, h * c, generator=g, device=cuda)
    torch.manual_seed(2)
    conv = GATConv(f, c, heads=h).to(cuda).eval()

    def run(inagg, stream=None, option=None):
        conv.input_aggregation, conv.stream = inagg, stream
        if option:
            _lib.set_option(option, 0)
        try:
            conv.zero_grad()
            out = conv(x, ei)
            out.backward(go)
        finally:
            if option:
                _lib.set_option(option, -1)
        return [out.detach().clone()] + [p.grad.clone() for p in conv.parameters()]

    base = run(True)
    names = ["out"] + [k for k, _ in conv.named_parameters()]
    # Kernel variants of one path see the same fp32 logits: they agree to re-association.  The two
    # PATHS round the logits differently (a_src = <x, W^T att> against <W x, att>), so among the 2.6e8
    # (edge, head) pairs a handful whose logit lies within ~2e-7 of zero take the other LeakyReLU
    # slope -- measured against the fp64 oracle at 1/10 size: 2 of 2.6e7, each worth ~1e-4 of a
    # gradient's largest element.  That is fp32, not a kernel property (the reference's own fp32
    # run flips its own handful): the forward is held to 3e-5, the gradients across paths to 2e-3.
    for label, other, gtol in (("projected rows", run(False), 2e-3),
                               ("register forward", run(True, stream=False), 3e-5),
                               ("head-group backward", run(True, option="GATMI_BWD3"), 3e-5),
                               ("chain in the row loop", run(True, option="GATMI_BWD2"), 3e-5)):
        for name, a, b in zip(names, base, other):
            scale = float(a.abs().max())
            tol = 3e-5 if name == "out" else gtol
            assert float((a - b).abs().max()) <= tol * scale, (label, name)


def test_full_size_backward_is_linear_and_dropout_reproducible(cuda, full):
    from gat_amd import GATConv
    ei, n, f, h, c, _ = full
    g = torch.Generator(device=cuda).manual_seed(4)
    x = torch.randn(n, f, generator=g, device=cuda)
    g1 = torch.randn(n, h * c, generator=g, device=cuda)
    g2 = torch.randn(n, h * c, generator=g, device=cuda)
    torch.manual_seed(5)
    conv = GATConv(f, c, heads=h, dropout=0.425).to(cuda)

    def grads(go, seed):
        conv.train(); conv.zero_grad()
        torch.manual_seed(seed)                                       # the layer draws its hash seed from torch
        out = conv(x, ei)
        out.backward(go)
        return out.detach().clone(), conv.lin.weight.grad.clone(), conv.att_src.grad.clone()

    o1, w1, a1 = grads(g1, 7)
    o1b, w1b, _ = grads(g1, 7)
    assert torch.equal(o1, o1b) and torch.equal(w1, w1b)              # same seed: same mask, same bits
    o3, _, _ = grads(g1, 8)
    assert not torch.equal(o1, o3)                                     # another seed: another mask
    _, w2, a2 = grads(g2, 7)
    _, w12, a12 = grads(g1 + 2 * g2, 7)
    assert float((w1 + 2 * w2 - w12).abs().max() / w12.abs().max()) < 3e-5
    assert float((a1 + 2 * a2 - a12).abs().max() / a12.abs().max()) < 3e-5
    kept = float((o1 != o3).float().mean())
    assert kept > 0.5                                                  # most rows see a different mask


def test_runpy_shaped_million_nodes(cuda):
    """run.py's input layer 25 -> 8x30 on 10^6 nodes with 5 out-edges each (the graph
    /root/reference/gat/converter.py:7-15 emits, at the largest size SURVEY.md 8d names): the
    row-per-lane-group edge kernels, q = g W and out = Z W^T per pair of heads (the latter only
    runs from 3*10^5 rows).  Constant rows in -> x W^T + b out; parameter gradients linear in
    grad_out; and the same step with those three choices switched off agrees to fp32
    re-association."""
    from gat_amd import GATConv, _lib, synth
    n, f, h, c = 1000000, 25, 8, 30
    ei = synth.out_degree_k_graph(n, 5, seed=1, device=cuda)
    torch.manual_seed(0)
    conv = GATConv(f, c, heads=h, dropout=0.0, input_aggregation=True).to(cuda).eval()
    with torch.no_grad():
        conv.bias.normal_()
    x = torch.rand(1, f, device=cuda).expand(n, f).contiguous()
    with torch.no_grad():
        out = conv(x, ei)
        expect = x[:1] @ conv.lin.weight.t() + conv.bias
    assert float((out - expect).abs().max()) <= 2e-5 * float(expect.abs().max())
    xr = torch.rand(n, f, device=cuda)
    g1 = torch.randn(n, h * c, device=cuda)
    g2 = torch.randn(n, h * c, device=cuda)

    def step(go):
        conv.zero_grad()
        o = conv(xr, ei)
        o.backward(go)
        return o.detach(), [p.grad.clone() for p in (conv.lin.weight, conv.att_src, conv.att_dst, conv.bias)]

    _, a = step(g1)
    _, b = step(g2)
    o1, t = step(g1 + 2 * g2)
    for u, v, w in zip(a, b, t):
        assert float((u + 2 * v - w).abs().max()) <= 3e-5 * float(w.abs().max())
    for name in ("GATMI_GROUPROWS", "GATMI_QGROUP", "GATMI_OUTGROUP"):
        _lib.set_option(name, 0)
    try:
        o2, t2 = step(g1 + 2 * g2)
    finally:
        for name in ("GATMI_GROUPROWS", "GATMI_QGROUP", "GATMI_OUTGROUP"):
            _lib.set_option(name, -1)
    assert float((o1 - o2).abs().max()) <= 2e-5 * float(o1.abs().max())
    for name, u, v in zip(("weight", "att_src", "att_dst", "bias"), t, t2):
        # a logit within rounding of zero may take the other LeakyReLU slope
        assert float((u - v).abs().max()) <= 2e-3 * float(v.abs().max()), name
    # The default at this size: fused forward tiles, q / d_W / S = <q, Z> from one pass over grad_out, the
    # edge kernel without its Z rows.  Against the separate kernels of the same layout (0) and the fully
    # fused backward (15) the logits are the same floats: agreement to re-association
    for fused in (0, 15):
        _lib.set_option("GATMI_FUSED", fused)
        try:
            o3, t3 = step(g1 + 2 * g2)
        finally:
            _lib.set_option("GATMI_FUSED", -1)
        assert float((o1 - o3).abs().max()) <= 2e-5 * float(o1.abs().max()), fused
        for name, u, v in zip(("weight", "att_src", "att_dst", "bias"), t, t3):
            assert float((u - v).abs().max()) <= 3e-5 * float(v.abs().max()), (fused, name)


def test_eight_heads_wide_rows_at_scale(cuda):
    """128 -> 8x32 on 200 k nodes / 5 M power-law edges (hub rows cut into segments): the
    multi-edge kernels for 5..8 heads (two edges per wave instruction at this row length) --
    constant rows, linearity of the parameter gradients, agreement with the one-edge kernels."""
    from gat_amd import GATConv, _lib, synth
    n, e, f, h, c = 200000, 5000000, 128, 8, 32
    ei = synth.powerlaw_graph(n, e, 3000, seed=2, device=cuda)
    torch.manual_seed(1)
    conv = GATConv(f, c, heads=h, dropout=0.0, input_aggregation=True, stream=False).to(cuda).eval()
    with torch.no_grad():
        conv.bias.normal_()
        x = torch.randn(1, f, device=cuda).expand(n, f).contiguous()
        out = conv(x, ei)
        expect = x[:1] @ conv.lin.weight.t() + conv.bias
    assert float((out - expect).abs().max()) <= 2e-5 * float(expect.abs().max())
    xr = torch.randn(n, f, device=cuda)
    g1 = torch.randn(n, h * c, device=cuda)
    g2 = torch.randn(n, h * c, device=cuda)

    def step(go):
        conv.zero_grad()
        o = conv(xr, ei)
        o.backward(go)
        return o.detach(), [p.grad.clone() for p in (conv.lin.weight, conv.att_src, conv.att_dst, conv.bias)]

    _, a = step(g1)
    _, b = step(g2)
    o1, t = step(g1 + 2 * g2)
    for u, v, w in zip(a, b, t):
        assert float((u + 2 * v - w).abs().max()) <= 3e-5 * float(w.abs().max())
    _lib.set_option("GATMI_EDGE8", 0)
    try:
        o2, t2 = step(g1 + 2 * g2)
    finally:
        _lib.set_option("GATMI_EDGE8", -1)
    assert float((o1 - o2).abs().max()) <= 3e-5 * float(o1.abs().max())
    for name, u, v in zip(("weight", "att_src", "att_dst", "bias"), t, t2):
        assert float((u - v).abs().max()) <= 2e-3 * float(v.abs().max()), name


def test_split_projection_beyond_four_gib(cuda):
    """Tall projection whose operands pass 4 GiB (4.3 M rows x 256 floats): the three-way split GEMMs
    address a block's rows with 32-bit offsets from the block's first row and clamp them to the end
    of the matrix -- rows at the start, across the 4-GiB mark and at the end against fp64 products,
    d_W against the f32 MFMA kernels."""
    from gat_amd import _lib, ops
    n, f, h, c = 4300003, 256, 4, 64
    g = torch.Generator(device=cuda).manual_seed(3)
    x = torch.randn(n, f, device=cuda, generator=g)
    w = torch.randn(h * c, f, device=cuda, generator=g) / 16
    att = torch.randn(h * c, device=cuda, generator=g)
    hp, a_dst = ops.project_fwd(x, w, att, h, c)
    mark = (1 << 32) // (f * 4)
    rows = torch.cat([torch.arange(0, 300), torch.arange(mark - 300, mark + 300), torch.arange(n - 300, n),
                      torch.randint(0, n, (2000,))]).to(cuda)
    ref = x[rows].double() @ w.double().t()
    assert float((hp[rows].double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    d_hp = torch.randn(n, h * c, device=cuda, generator=g)
    d_a = torch.randn(n, h, device=cuda, generator=g)
    d_w, d_x = ops.project_bwd(x, w, d_hp, h, c, True, d_a, att)
    full = d_hp[rows].double() + d_a[rows].double().repeat_interleave(c, 1) * att.double()
    ref = full @ w.double()
    assert float((d_x[rows].double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    del hp, d_x
    _lib.set_option("GATMI_SPLIT3", 0)
    try:
        d_w0, _ = ops.project_bwd(x, w, d_hp, h, c, False, d_a, att)
    finally:
        _lib.set_option("GATMI_SPLIT3", -1)
    assert float((d_w - d_w0).abs().max()) <= 2e-5 * float(d_w0.abs().max())


def test_full_size_bf16_source_rows(cuda, full):
    """The input layer on bf16 source rows at full size (627 MB table, 64 M gathers of 256-byte rows):
    an x that bf16 holds exactly gives the f32 kernels' results up to re-association; any x stays
    inside the stated bf16 tolerance of tests/test_gpu_bf16.py against the f32 kernels."""
    from gat_amd import GATConv
    ei, n, f, h, c, layout = full
    g = torch.Generator(device=cuda).manual_seed(11)
    go = torch.randn(n, h * c, generator=g, device=cuda)
    for exact in (True, False):
        x = torch.randn(n, f, generator=g, device=cuda)
        if exact:
            x = x.bfloat16().float()
        res = []
        for dt in (None, torch.bfloat16):
            torch.manual_seed(5)
            conv = GATConv(f, c, heads=h, dropout=0.0, input_aggregation=True, feature_dtype=dt).to(cuda)
            out = conv(x, ei)
            out.backward(go)
            res.append((out.detach(), conv.lin.weight.grad.clone(), conv.att_src.grad.clone(), conv.bias.grad.clone()))
        (o32, w32, a32, b32), (o16, w16, a16, b16) = res
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
        l2 = lambda a, b: float((a - b).norm() / b.norm())
        if exact:
            assert rel(o16, o32) < 1e-5 and rel(w16, w32) < 2e-4 and rel(b16, b32) < 1e-5, (rel(o16, o32), rel(w16, w32))
        else:
            assert rel(o16, o32) < 1e-2 and l2(w16, w32) < 5e-2 and l2(a16, a32) < 2.5e-1 and rel(b16, b32) < 1e-5
