"""Exchange "h" in pieces (gat_amd/dist.py _PipelinedFn), on the real kernels:
one GPU plays every rank.  The collectives are replaced by what they compute -- the table is
assembled row by row as TableMap places them, the per-piece d_h of all ranks are summed -- and
every rank's per-piece forward (gatmi_attn_fwd_partial + gatmi_attn_fwd_combine) and backward
(gatmi_attn_bwd_begin / _chunk / _end) go through the C ABI against the CPU oracle of the WHOLE
graph, at the f32 bar of tests/test_gpu_parity.py.  K = 2, 3, 4 pieces, padded and exact tables,
injected attention-dropout masks, bf16 tables, cut rows (seg_len 64 on a star)."""
import pytest
import torch

import cases
from oracle import gat_oracle as O
from test_gpu_parity import close

pytestmark = pytest.mark.gpu


def _piecewise_layer(cuda, ei, n, f, h, c, world, chunks, padding, p=0.0, mask=None, seg=64,
                     bf16=False):
    """-> out [n, H*C], grads dict; every rank's share computed piece by piece."""
    from gat_amd import _lib, ops
    from gat_amd.dist import HipBackend, chunk_edge_lists, make_partition, _piece_cfg
    x, w, a_s, a_d, b, go = cases.layer_inputs(n, f, h, c)
    be = HipBackend(seg_len=seg)
    cfg = ops.AttnConfig(h, c, dropout=p, mask_mode=_lib.MASK_BYTES if mask is not None else _lib.MASK_NONE)
    cfg.feat_bf16 = bf16
    xd, wd, bd, god = x.to(cuda), w.to(cuda), b.to(cuda), go.to(cuda)
    asd, add_ = a_s.to(cuda).view(-1), a_d.to(cuda).view(-1)
    eid = ei.to(cuda)
    parts = [make_partition(eid, n, world, r, "edges", padding) for r in range(world)]
    tm = parts[0].table(chunks)
    assert tm.uneven == (padding == "exact")
    # every rank projects its rows; the "all-gather" = each row at its place in the table
    shards = [be.project_fwd(xd[q.lo:q.hi].contiguous(), wd, add_, cfg) for q in parts]
    hp_all = torch.cat([s_[0] for s_ in shards])
    table = hp_all.new_zeros((tm.total, hp_all.size(1)))
    where = tm.to_table(torch.arange(n, device=cuda))
    table[where] = hp_all
    piece = lambda t, k: t[tm.base[k]: tm.base[k] + tm.rows[k]]
    norm = O.normalize_edges(ei, n)
    d_table = torch.zeros(tm.total, hp_all.size(1), device=cuda)
    outs, d_as, d_ad, d_b, d_adst = [], 0, 0, 0, []
    for q, (hp_own, a_dst) in zip(parts, shards):
        lists = chunk_edge_lists(eid, q, tm)
        layouts = [be.build_chunk_layout(e_, tm.rows[k], q.n_local) if tm.rows[k] else None
                   for k, (e_, _) in enumerate(lists)]
        assert sum(l.num_edges for l in layouts if l is not None) == int(((norm[1] >= q.lo) & (norm[1] < q.hi)).sum())
        if mask is not None:
            # the rank's edges in the normalised list, then the piece's edges among those, then CSR order
            mine = mask[(norm[1] >= q.lo) & (norm[1] < q.hi)].to(cuda)
            cfg.piece_masks = [None if l is None else l.edge_mask_to_csr(mine[sel])
                               for l, (_, sel) in zip(layouts, lists)]
        live = [k for k in range(chunks) if layouts[k] is not None]
        triples = [be.attn_fwd_partial(layouts[k], _piece_cfg(cfg, k), f, piece(table, k), a_dst, asd)
                   for k in live]
        out, lse = be.attn_fwd_combine(triples, cfg, bd, True)
        outs.append(out)
        cb = be.chunked_backward(cfg, f, a_dst, asd, add_, bd, out, lse, god[q.lo:q.hi].contiguous(), chunks)
        for k in live:
            piece(d_table, k).add_(cb.chunk(k, layouts[k], _piece_cfg(cfg, k), piece(table, k)))
        r = cb.end(hp_own)
        d_as, d_ad, d_b = d_as + r[0], d_ad + r[1], d_b + r[2]
        d_adst.append(r[3])
    d_h = d_table[where]
    d_w, d_x = 0, []
    for q, dd in zip(parts, d_adst):
        blk = d_h[q.lo:q.hi].contiguous()
        be.add_dst_term(blk, dd, add_, cfg)
        dw, dx = be.project_bwd(xd[q.lo:q.hi].contiguous(), wd, blk, cfg, True)
        d_w = d_w + dw
        d_x.append(dx)
    return torch.cat(outs), dict(x=torch.cat(d_x), weight=d_w, att_src=d_as, att_dst=d_ad, bias=d_b)


CASES = [("multigraph", "conv1_h30", 2, 2, 0.0), ("star", "arxiv_h64", 3, 3, 0.0),
         ("hub_source", "three_heads", 2, 4, 0.4), ("knn", "cora_l2", 4, 2, 0.0),
         ("star", "hidden_8x32", 2, 4, 0.5), ("multigraph", "arxiv_h64", 3, 2, 0.3),
         ("chain", "conv1_h30", 3, 4, 0.0), ("star", "arxiv_h64", 3, 1, 0.3)]


@pytest.mark.parametrize("padding", ["pad", "exact"])
@pytest.mark.parametrize("gname,sname,world,chunks,p", CASES)
def test_pieces_match_oracle(cuda, gname, sname, world, chunks, p, padding):
    ei, n = cases.GRAPHS[gname]()
    f, h, c = cases.SHAPES[sname]
    x, w, a_s, a_d, b, go = cases.layer_inputs(n, f, h, c)
    ne = O.normalize_edges(ei, n).size(1)
    mask = cases.keep_mask(ne, h, p) if p > 0 else None
    ref_out, ref_g = O.gatconv_grads_autograd(x, ei, w, a_s, a_d, b, h, c, go, dropout=p, mask=mask,
                                              training=p > 0)
    out, g = _piecewise_layer(cuda, ei, n, f, h, c, world, chunks, padding, p, mask)
    summand = float(go.abs().max() * (x @ w.t()).abs().max())
    close(out, ref_out, "out (pieces)")
    close(g["x"], ref_g["x"], "d_x (pieces)")
    close(g["weight"], ref_g["weight"], "d_weight (pieces)")
    close(g["att_src"], ref_g["att_src"], "d_att_src (pieces)", floor=summand)
    close(g["att_dst"], ref_g["att_dst"], "d_att_dst (pieces)", floor=summand)
    close(g["bias"], ref_g["bias"], "d_bias (pieces)")


def test_pieces_are_bit_reproducible_and_agree_with_one_piece(cuda):
    """A fixed chunking gives the same bits on every run (no atomics: piece order and segment order
    are fixed); different chunkings re-associate the softmax sums, so they agree to the f32 bar."""
    ei, n = cases.GRAPHS["star"]()
    f, h, c = cases.SHAPES["arxiv_h64"]
    runs = [_piecewise_layer(cuda, ei, n, f, h, c, 2, k, "pad") for k in (3, 3, 1)]
    assert torch.equal(runs[0][0], runs[1][0])
    for name in runs[0][1]:
        assert torch.equal(runs[0][1][name], runs[1][1][name]), name
    close(runs[0][0], runs[2][0], "out: 3 pieces vs 1")
    close(runs[0][1]["x"], runs[2][1]["x"], "d_x: 3 pieces vs 1")


@pytest.mark.parametrize("world,chunks", [(2, 3), (3, 2)])
def test_pieces_on_bf16_tables(cuda, world, chunks):
    """bf16 feature table in pieces (what `--dtype bf16` puts on the wire).  Same table, same
    rounding points as the one-piece run, so the two agree at the f32 bar; against the oracle with
    those rounding points the output is held to the bf16 tolerance of tests/test_gpu_bf16.py."""
    from gat_amd import synth
    n, f, h, c = 2500, 64, 4, 64
    ei = synth.powerlaw_graph(n, 8 * n, 500, seed=5)
    x, w, a_s, a_d, b, go = cases.layer_inputs(n, f, h, c)
    ref_out, _ = O.gatconv_grads_autograd(x, ei, w, a_s, a_d, b, h, c, go, bf16_tables=True, bf16_matmul=True)
    one = _piecewise_layer(cuda, ei, n, f, h, c, world, 1, "pad", bf16=True)
    many = _piecewise_layer(cuda, ei, n, f, h, c, world, chunks, "pad", bf16=True)
    assert float((many[0].cpu() - ref_out).abs().max() / ref_out.abs().max()) <= 2e-3
    close(many[0], one[0], "out (bf16 pieces vs one piece)")
    summand = float(go.abs().max() * (x @ w.t()).abs().max())
    close(many[1]["bias"], one[1]["bias"], "d_bias (bf16 pieces vs one piece)")
    for name in ("x", "weight"):     # d_h is rounded to bf16 on its way into these GEMMs: an f32
        d = (many[1][name] - one[1][name]).norm() / one[1][name].norm()     # re-association flips ulps
        assert float(d) <= 2e-3, (name, float(d))
    for name in ("att_src", "att_dst"):
        close(many[1][name], one[1][name], f"d_{name} (bf16 pieces vs one piece)", floor=summand)


def test_combine_rejects_too_many_pieces(cuda):
    from gat_amd import _lib, ops
    m = torch.zeros(4, 2, device=cuda)
    acc = torch.zeros(4, 16, device=cuda)
    with pytest.raises(ValueError):
        ops.attn_fwd_combine([(m, m, acc)] * (_lib.MAX_PARTS + 1), ops.AttnConfig(2, 8), None, False)
