"""world_size-2 gloo test of the destination-partitioned layer (gat_amd/dist.py): the
partitioning, the all-gather / reduce-scatter / all-reduce pattern and the deferred
d_a_dst term reproduce the single-process oracle.  The GPU kernels cannot run here, so
the compute steps are a CPU stand-in written with autograd over the oracle's formulas;
what is under test is everything around them."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
from oracle import gat_oracle as O


class OracleBackend:
    """Same interface as gat_amd.dist.HipBackend, compact (unpadded) feature rows."""

    def build_layout(self, local_ei, part):
        return dict(ei=local_ei, part=part, num_edges=local_ei.size(1))

    def project_fwd(self, x, w, att_d, cfg):
        hp = x @ w.t()
        a_dst = (hp.view(-1, cfg.heads, cfg.channels) * att_d.view(1, cfg.heads, cfg.channels)).sum(-1)
        return hp, a_dst

    def _out(self, layout, cfg, hp_full, a_dst, att_s, bias):
        h_, c_ = cfg.heads, cfg.channels
        src, dst = layout["ei"][0], layout["ei"][1]
        n_loc = layout["part"].n_local
        hh = hp_full.view(-1, h_, c_)
        a_s = (hh * att_s.view(1, h_, c_)).sum(-1)
        z = torch.nn.functional.leaky_relu(a_s[src] + a_dst[dst], cfg.negative_slope)
        alpha = O.segment_softmax(z, dst, n_loc)
        out = torch.zeros(n_loc, h_, c_, dtype=hp_full.dtype).index_add_(0, dst, alpha.unsqueeze(-1) * hh[src])
        out = out.view(n_loc, h_ * c_)
        return out + bias if bias is not None else out

    def attn_fwd(self, layout, cfg, f, hp_full, a_dst, att_s, att_d, bias, need_lse):
        with torch.no_grad():
            return self._out(layout, cfg, hp_full, a_dst, att_s, bias), torch.zeros(1)

    def attn_bwd(self, layout, cfg, f, hp_full, a_dst, att_s, att_d, bias, out, lse, grad_out):
        part = layout["part"]
        leaves = [t.detach().clone().requires_grad_(True) for t in (hp_full, a_dst, att_s, bias)]
        with torch.enable_grad():
            o = self._out(layout, cfg, leaves[0], leaves[1], leaves[2], leaves[3])
            o.backward(grad_out)
        d_a_dst = leaves[1].grad
        h_loc = hp_full[part.plo:part.phi].view(-1, cfg.heads, cfg.channels)
        d_att_dst = (d_a_dst.unsqueeze(-1) * h_loc).sum(0).reshape(-1)
        return leaves[0].grad, leaves[2].grad, d_att_dst, leaves[3].grad, d_a_dst

    # ---- source chunks (exchange "h" in pieces): the closed forms of SURVEY.md 8a row a11
    def build_chunk_layout(self, edges, num_src, num_dst):
        return dict(ei=edges, num_src=num_src, num_dst=num_dst, num_edges=edges.size(1))

    def attn_fwd_partial(self, layout, cfg, f, hp_piece, a_dst, att_s):
        h_, c_, n = cfg.heads, cfg.channels, layout["num_dst"]
        src, dst = layout["ei"][0], layout["ei"][1]
        assert hp_piece.size(0) == layout["num_src"]
        hh = hp_piece.view(-1, h_, c_)
        a_s = (hh * att_s.view(1, h_, c_)).sum(-1)
        z = torch.nn.functional.leaky_relu(a_s[src] + a_dst[dst], cfg.negative_slope)
        m = torch.full((n, h_), -float("inf"), dtype=z.dtype).scatter_reduce_(
            0, dst.view(-1, 1).expand(-1, h_), z, "amax", include_self=True)
        p = torch.exp(z - m[dst])
        s = torch.zeros(n, h_, dtype=z.dtype).index_add_(0, dst, p)
        acc = torch.zeros(n, h_, c_, dtype=z.dtype).index_add_(0, dst, p.unsqueeze(-1) * hh[src])
        return m, s, acc.view(n, h_ * c_)

    def attn_fwd_combine(self, parts, cfg, bias, need_lse):
        h_, c_ = cfg.heads, cfg.channels
        big = torch.stack([m for m, _, _ in parts]).max(0).values
        tot, acc = 0, 0
        for m, s, a in parts:
            f = torch.where(torch.isinf(m), torch.zeros_like(m), torch.exp(m - big))
            tot = tot + s * f
            acc = acc + f.unsqueeze(-1) * a.view(-1, h_, c_)
        out = (acc / tot.unsqueeze(-1)).reshape(-1, h_ * c_)
        return (out + bias if bias is not None else out), big + torch.log(tot)

    def chunked_backward(self, cfg, f, a_dst, att_s, att_d, bias, out, lse, grad_out, max_chunks):
        be = self

        class _CB:
            def __init__(self):
                h_, c_ = cfg.heads, cfg.channels
                self.g = grad_out.view(-1, h_, c_)
                o = out if bias is None else out - bias
                self.S = (self.g * o.view(-1, h_, c_)).sum(-1)
                self.d_a_dst = torch.zeros_like(a_dst)
                self.d_att_src = torch.zeros(h_, c_, dtype=a_dst.dtype)

            def chunk(self, k, layout, cfg_k, hp_piece):
                h_, c_ = cfg.heads, cfg.channels
                src, dst = layout["ei"][0], layout["ei"][1]
                hh = hp_piece.view(-1, h_, c_)
                a_s = (hh * att_s.view(1, h_, c_)).sum(-1)
                u = a_s[src] + a_dst[dst]
                alpha = torch.exp(torch.nn.functional.leaky_relu(u, cfg.negative_slope) - lse[dst])
                d_at = (self.g[dst] * hh[src]).sum(-1)
                du = (alpha * d_at - alpha * self.S[dst]) * torch.where(u > 0, torch.ones_like(u), torch.full_like(u, cfg.negative_slope))
                d_h = torch.zeros_like(hh).index_add_(0, src, alpha.unsqueeze(-1) * self.g[dst])
                d_a_src = torch.zeros_like(a_s).index_add_(0, src, du)
                d_h += d_a_src.unsqueeze(-1) * att_s.view(1, h_, c_)
                self.d_a_dst.index_add_(0, dst, du)
                self.d_att_src += (d_a_src.unsqueeze(-1) * hh).sum(0)
                return d_h.view(hp_piece.size(0), -1)

            def end(self, hp_own):
                h_, c_ = cfg.heads, cfg.channels
                d_ad = (self.d_a_dst.unsqueeze(-1) * hp_own.view(-1, h_, c_)).sum(0).reshape(-1)
                return self.d_att_src.reshape(-1), d_ad, grad_out.sum(0), self.d_a_dst

        return _CB()

    def add_dst_term(self, d_hp, d_a_dst, att_d, cfg):
        d_hp.view(-1, cfg.heads, cfg.channels).add_(
            d_a_dst.unsqueeze(-1) * att_d.view(1, cfg.heads, cfg.channels))

    def project_bwd(self, x, w, d_hp, cfg, need_dx):
        return d_hp.t() @ x, (d_hp @ w if need_dx else None)


    # input-side aggregation stand-in: packed rows are x | a_src (no padding here)
    def inagg_supported(self, f, cfg):
        return True

    def inagg_logit_columns(self, f, cfg):
        return f, cfg.heads

    def inagg_pack(self, x, w, att_s, att_d, cfg):
        h_, c_ = cfg.heads, cfg.channels
        hh = (x @ w.t()).view(-1, h_, c_)
        a_src = (hh * att_s.view(1, h_, c_)).sum(-1)
        a_dst = (hh * att_d.view(1, h_, c_)).sum(-1)
        return torch.cat([x, a_src], dim=1), a_dst

    def _inagg_out(self, layout, cfg, x_full, a_src, a_dst, w, bias):
        h_, c_ = cfg.heads, cfg.channels
        src, dst = layout["ei"][0], layout["ei"][1]
        n_loc = layout["part"].n_local
        z = torch.nn.functional.leaky_relu(a_src[src] + a_dst[dst], cfg.negative_slope)
        alpha = O.segment_softmax(z, dst, n_loc)
        agg = torch.zeros(n_loc, h_, x_full.size(1), dtype=x_full.dtype).index_add_(
            0, dst, alpha.unsqueeze(-1) * x_full[src].unsqueeze(1))
        out = torch.einsum("nhf,hcf->nhc", agg, w.view(h_, c_, -1)).reshape(n_loc, h_ * c_)
        return out + bias if bias is not None else out

    def inagg_fwd(self, layout, cfg, table, a_dst, w, att_s, att_d, bias, need_lse):
        f = w.size(1)
        with torch.no_grad():
            out = self._inagg_out(layout, cfg, table[:, :f], table[:, f:f + cfg.heads], a_dst, w, bias)
        return out, torch.zeros(1), torch.zeros(1)

    def inagg_bwd(self, layout, cfg, table, a_dst, w, att_s, att_d, bias, z, lse, grad_out):
        part, f, h_, c_ = layout["part"], w.size(1), cfg.heads, cfg.channels
        x_full = table[:, :f]
        wl, asl, adl, bl = [t.detach().clone().requires_grad_(True) for t in (w, att_s, att_d, bias)]
        with torch.enable_grad():   # logits recomputed from the leaves so that d_W sees them
            hh = (x_full @ wl.t()).view(-1, h_, c_)
            a_src = (hh * asl.view(1, h_, c_)).sum(-1)
            a_d = (hh[part.plo:part.phi] * adl.view(1, h_, c_)).sum(-1)
            self._inagg_out(layout, cfg, x_full, a_src, a_d, wl, bl).backward(grad_out)
        return wl.grad, asl.grad, adl.grad, bl.grad


def _worker(rank, world, port, n, gname, exchange, balance, q, chunks=1, padding="auto"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gat_amd.dist import PartitionedGATConv
        torch.set_num_threads(1)
        ei, n_ = cases.GRAPHS[gname]() if n is None else cases.random_multigraph(n=n, e=6 * n)
        f, h, c = 6, 3, 5
        x, w, a_s, a_d, b, go = [t.double() for t in cases.layer_inputs(n_, f, h, c)]
        conv = PartitionedGATConv(f, c, heads=h, backend=OracleBackend(), exchange=exchange,
                                  balance=balance, chunks=chunks, padding=padding).double().eval()
        part = conv.partition_for(ei, n_)
        assert part.uneven == (padding == "exact") or padding == "auto"
        if balance == "edges" and gname == "star":      # the hub's range is much shorter than the rest
            assert part.bounds is not None and part.bounds[1] < n_ // world
        with torch.no_grad():
            conv.lin.weight.copy_(w); conv.att_src.copy_(a_s); conv.att_dst.copy_(a_d); conv.bias.copy_(b)
        inagg = exchange in ("xs", "as")
        xl = x[part.lo:part.hi].clone().requires_grad_(not inagg)
        steps = 3 if exchange == "as" else 1     # second step: cached rows, fresh logits
        for step in range(steps):
            conv.zero_grad()
            if step == 1:                         # the weights moved: a_src must be re-gathered
                with torch.no_grad():
                    w = w * 1.25; conv.lin.weight.copy_(w)
            if step == 2:                         # ANOTHER x of the same shape, most likely at the
                x = x.flip(0) * 0.5               # address just freed: the cached rows are stale
                del xl
                xl = x[part.lo:part.hi].clone()
            out = conv(xl, ei, n_)
            out.backward(go[part.lo:part.hi])
        ref_out, ref_g = O.gatconv_grads_autograd(x, ei, w, a_s, a_d, b, h, c, go)
        tol = dict(rtol=1e-9, atol=1e-10)
        torch.testing.assert_close(out.detach(), ref_out[part.lo:part.hi], **tol)
        if not inagg:
            torch.testing.assert_close(xl.grad, ref_g["x"][part.lo:part.hi], **tol)
        torch.testing.assert_close(conv.lin.weight.grad, ref_g["weight"], **tol)
        torch.testing.assert_close(conv.att_src.grad, ref_g["att_src"], **tol)
        torch.testing.assert_close(conv.att_dst.grad, ref_g["att_dst"], **tol)
        torch.testing.assert_close(conv.bias.grad, ref_g["bias"], **tol)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    finally:
        dist.destroy_process_group()


def _stale_rank_worker(rank, world, port, q):
    """exchange "as": rank 1 alone passes a NEW x tensor at the second step.  Its cache misses, the
    other ranks' caches hit -- the MIN all-reduce makes every rank take the full gather, and the
    result is that of the new x everywhere (the branch must be decided collectively)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gat_amd.dist import PartitionedGATConv
        torch.set_num_threads(1)
        ei, n_ = cases.random_multigraph(n=40, e=240)
        f, h, c = 6, 3, 5
        x, w, a_s, a_d, b, go = [t.double() for t in cases.layer_inputs(n_, f, h, c)]
        conv = PartitionedGATConv(f, c, heads=h, backend=OracleBackend(), exchange="as",
                                  balance="nodes").double().eval()
        part = conv.partition_for(ei, n_)
        with torch.no_grad():
            conv.lin.weight.copy_(w); conv.att_src.copy_(a_s); conv.att_dst.copy_(a_d); conv.bias.copy_(b)
        xl = x[part.lo:part.hi].clone()
        conv(xl, ei, n_)                                   # fills every rank's cache
        x2 = x.clone()
        x2[part.lo:part.hi] = x[part.lo:part.hi] * (2.0 if rank == 1 else 1.0)
        x2_all = x.clone()
        lo1 = (n_ + world - 1) // world
        x2_all[lo1:2 * lo1] *= 2.0                          # what the whole x is after rank 1's change
        if rank == 1:
            xl = x2[part.lo:part.hi].clone()               # a new tensor object on this rank only
        out = conv(xl, ei, n_)
        ref_out, _ = O.gatconv_grads_autograd(x2_all, ei, w, a_s, a_d, b, h, c, go)
        torch.testing.assert_close(out.detach(), ref_out[part.lo:part.hi], rtol=1e-9, atol=1e-10)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    finally:
        dist.destroy_process_group()


def test_cache_hit_of_exchange_as_is_decided_by_all_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stale_rank_worker, args=(r, 3, 29681, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def _run(world, n, gname, exchange, port, balance="nodes", chunks=1, padding="auto"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, gname, exchange, balance, q, chunks, padding))
             for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


@pytest.mark.parametrize("world,n,gname,exchange,port", [
    (2, 50, None, "h", 29611), (2, 51, None, "x", 29612), (2, None, "star", "h", 29613),
    (3, 7, None, "h", 29614), (2, None, "star", "x", 29615), (3, 7, None, "x", 29616),
    (2, 50, None, "xs", 29617), (2, 51, None, "as", 29618), (3, 7, None, "as", 29619),
    (2, None, "star", "as", 29620), (3, None, "star", "xs", 29621)])
def test_partitioned_layer_equals_single_process(world, n, gname, exchange, port):
    _run(world, n, gname, exchange, port)


@pytest.mark.parametrize("world,n,gname,exchange,port", [
    (2, None, "star", "h", 29641), (3, None, "star", "as", 29642), (2, 50, None, "x", 29643),
    (3, 50, None, "xs", 29644), (3, None, "hub_source", "h", 29645), (4, 9, None, "as", 29646)])
def test_partitioned_layer_with_edge_balanced_ranges(world, n, gname, exchange, port):
    """Destination ranges of equal in-edge count -- unequal node counts, shards
    padded to the longest, sources renumbered into the padded table: same results."""
    _run(world, n, gname, exchange, port, balance="edges")


@pytest.mark.parametrize("world,n,gname,balance,chunks,padding,port", [
    (2, 50, None, "nodes", 2, "pad", 29661), (3, 50, None, "edges", 3, "pad", 29662),
    (2, None, "star", "edges", 4, "pad", 29663), (3, None, "hub_source", "edges", 2, "exact", 29664),
    (3, 7, None, "nodes", 4, "pad", 29665), (2, 51, None, "edges", 3, "exact", 29666),
    (4, 9, None, "edges", 4, "exact", 29667)])
def test_pipelined_exchange_equals_single_process(world, n, gname, balance, chunks, padding, port):
    """Exchange "h" in pieces: the all-gather of h issued as one asynchronous
    collective per piece, attention over the sources of piece k from a layout of its own, the
    (m, s, acc) triples joined at the end; backward per piece with its reduce-scatter in flight --
    output and every gradient equal the single-process oracle, on padded (equal-shard collectives)
    and exact (per-owner broadcasts / reductions) tables, incl. pieces that hold no rows (7 nodes
    over 3 ranks in 4 pieces) and ranks that own none."""
    _run(world, n, gname, "h", port, balance, chunks, padding)


@pytest.mark.parametrize("world,n,gname,exchange,port", [
    (2, 50, None, "h", 29671), (3, None, "star", "as", 29672), (3, 50, None, "x", 29673),
    (2, 51, None, "xs", 29674)])
def test_partitioned_layer_on_exact_tables(world, n, gname, exchange, port):
    """padding="exact": tables of exactly N rows indexed by global id, per-owner collectives --
    every one-piece exchange mode."""
    _run(world, n, gname, exchange, port, "edges", 1, "exact")


def test_padding_is_bounded_on_a_degree_ordered_graph():
    """on a power-law graph whose nodes are ordered by degree the edge-balanced
    range that holds the low-degree tail is several times N / P long; padding every shard to it
    would multiply the collectives' volume and the gathered table.  make_partition then switches to
    exact tables (ratio 1.0); a graph whose ranges are near-equal keeps the equal-shard form."""
    from gat_amd import synth
    from gat_amd.dist import PAD_LIMIT, make_partition
    n, e, world = 20000, 300000, 8
    ei = synth.powerlaw_graph(n, e, 4000, seed=3)
    order = torch.argsort(torch.bincount(ei[1], minlength=n), descending=True)
    rank_of = torch.empty_like(order); rank_of[order] = torch.arange(n)
    skewed = rank_of[ei]
    padded = make_partition(skewed, n, world, 0, padding="pad")
    assert padded.padding_ratio > 2.0, padded.padding_ratio        # what round 2 moved
    auto = make_partition(skewed, n, world, 0)
    assert auto.uneven and auto.padding_ratio == 1.0 and auto.padded_nodes == n
    for k in (1, 4):
        tm = auto.table(k)
        assert tm.total == n                                        # no padding rows in any piece
        rows = tm.to_table(torch.arange(n))
        assert rows.unique().numel() == n and int(rows.max()) == n - 1
    mild = make_partition(ei, n, world, 0)                          # hubs spread over the id range
    assert not mild.uneven and mild.padding_ratio <= PAD_LIMIT
    for k in (1, 2, 3, 4):                                          # pieces are padded to ceil(block / k)
        assert mild.table(k).total <= mild.world * (mild.block + k)


def test_balanced_bounds_split_the_in_edges_evenly():
    from gat_amd import synth
    from gat_amd.dist import Partition, balanced_bounds, local_edges, make_partition
    n, e, world = 20000, 300000, 8
    ei = synth.powerlaw_graph(n, e, 4000, seed=3)
    # hubs at the low ids (a real graph ordered by degree): equal node blocks are badly skewed
    order = torch.argsort(torch.bincount(ei[1], minlength=n), descending=True)
    rank_of = torch.empty_like(order); rank_of[order] = torch.arange(n)
    ei = rank_of[ei]
    bounds = balanced_bounds(ei, n, world)
    assert bounds[0] == 0 and bounds[-1] == n and list(bounds) == sorted(bounds)
    per_edges = [local_edges(ei, Partition(n, world, r, bounds)).size(1) for r in range(world)]
    per_nodes = [local_edges(ei, Partition(n, world, r)).size(1) for r in range(world)]
    total = sum(per_edges)
    assert total == sum(per_nodes) == O.build_csr(ei, n).num_edges
    assert max(per_edges) <= 1.05 * total / world + 4000      # within 5 % + one hub row
    assert max(per_nodes) > 2 * total / world                 # what equal blocks would have given
    # padded ids: a bijection of the nodes into [0, world * block), rank-major
    part = make_partition(ei, n, world, 3, padding="pad")
    assert part.bounds == bounds
    pid = part.to_padded(torch.arange(n))
    assert pid.unique().numel() == n and int(pid.max()) < part.padded_nodes
    for r in range(world):
        pr = Partition(n, world, r, bounds)
        assert torch.equal(pid[pr.lo:pr.hi], torch.arange(pr.plo, pr.phi))


def test_exchange_mode_resolved_for_every_bench_workload():
    """What `exchange="auto"` picks for each synth.WORKLOADS entry (bench.py --gpus N): the input
    layers take the input-side path with only a_src on the wire ("as"); layers whose input is wider
    than the output (Cora-shaped) or carries a gradient (the hidden layers) all-gather h."""
    from gat_amd import ops, synth
    from gat_amd.dist import PartitionedGATConv
    want = {"cora": "h", "cora_uniform": "h", "arxiv": "as", "arxiv_uniform": "as", "products": "as",
            "products_uniform": "as", "runpy10k": "as", "runpy100k": "as", "runpy1m": "as",
            "arxiv_hidden": "h", "products_hidden": "h"}
    assert set(want) == set(synth.WORKLOADS)
    for name, (n, e, f, h, c, _) in synth.WORKLOADS.items():
        conv = PartitionedGATConv(f, c, heads=h)
        x = torch.zeros(4, f, requires_grad=name.endswith("_hidden"))
        assert conv.resolve_exchange(x, ops.AttnConfig(h, c)) == want[name], name
        if want[name] == "h":                                  # forcing the input-side modes raises
            conv.exchange = "as"
            with pytest.raises(RuntimeError, match="exchange='as'"):
                conv.resolve_exchange(x, ops.AttnConfig(h, c))


@pytest.mark.parametrize("balance", ["nodes", "edges"])
def test_local_edges_keep_reference_order(balance):
    from gat_amd.dist import local_edges, make_partition
    ei, n = cases.random_multigraph()
    norm = O.normalize_edges(ei, n)
    for rank in range(3):
        part = make_partition(ei, n, 3, rank, balance)
        loc = local_edges(ei, part)
        sel = (norm[1] >= part.lo) & (norm[1] < part.hi)
        assert torch.equal(loc[0], part.to_padded(norm[0][sel]))
        assert torch.equal(loc[1] + part.lo, norm[1][sel])
    assert sum(make_partition(ei, n, 3, r, balance).n_local for r in range(3)) == n


def _model_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gat_amd.dist import PartitionedGAT
        torch.set_num_threads(1)
        ei, n = cases.random_multigraph(n=61, e=400)
        g = torch.Generator().manual_seed(11)
        x = torch.randn(n, 9, generator=g, dtype=torch.float64)
        y = torch.randint(0, 3, (n,), generator=g)
        torch.manual_seed(5)
        ref = O.OracleGAT(9, 3, hidden_dim=4, dropout=0.0).double()
        with torch.no_grad():       # biases that are not zero, BatchNorm parameters that are not 1 / 0
            for p in ref.parameters():
                if p.dim() == 1:
                    p.add_(torch.randn(p.shape, generator=g, dtype=torch.float64) * 0.3)
        model = PartitionedGAT(9, 3, hidden_dim=4, dropout=0.0, backend=OracleBackend()).double()
        model.load_state_dict(ref.state_dict())
        part = model.partition_for(ei, n)
        for step in range(2):        # second step: running statistics carried over
            ref.train(); ref.zero_grad()
            ref_logp = ref(x, ei)
            ref_loss = torch.nn.functional.nll_loss(ref_logp, y)
            ref_loss.backward()
            model.train(); model.zero_grad()
            logp = model(x[part.lo:part.hi], ei, n)
            share = model.loss_share(logp, y[part.lo:part.hi], n)
            share.backward()
            model.sync_grads()
            tol = dict(rtol=1e-8, atol=1e-10)
            torch.testing.assert_close(logp.detach(), ref_logp.detach()[part.lo:part.hi], **tol)
            torch.testing.assert_close(model.global_loss(share), ref_loss.detach(), **tol)
            rp = dict(ref.named_parameters())
            for name, p in model.named_parameters():
                torch.testing.assert_close(p.grad, rp[name].grad, **tol, msg=lambda m, name=name: f"{name}: {m}")
            for name, b in model.named_buffers():
                torch.testing.assert_close(b, dict(ref.named_buffers())[name], **tol,
                                           msg=lambda m, name=name: f"{name}: {m}")
        ref.eval(); model.eval()
        with torch.no_grad():
            torch.testing.assert_close(model(x[part.lo:part.hi], ei, n), ref(x, ei)[part.lo:part.hi], **tol)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,port", [(2, 29631), (3, 29632)])
def test_partitioned_network_equals_single_process(world, port):
    """The 3-layer network with BatchNorm statistics and the loss taken over all ranks' rows:
    log-probabilities, loss, every parameter gradient and the running statistics of two training
    steps, then eval mode, against the single-process oracle network."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_model_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def _dropout_model_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch.nn.functional as F
        from gat_amd.dist import PartitionedGAT, make_partition
        torch.set_num_threads(1)
        torch.manual_seed(1234)                # the same on every rank, as a launcher would seed them
        p = 0.5
        ei, n = cases.random_multigraph(n=61, e=400)
        g = torch.Generator().manual_seed(11)
        x = torch.randn(n, 9, generator=g, dtype=torch.float64)
        y = torch.randint(0, 3, (n,), generator=g)
        ref = O.OracleGAT(9, 3, hidden_dim=4, dropout=0.0).double()     # feature dropout by hand below
        model = PartitionedGAT(9, 3, hidden_dim=4, dropout=p, backend=OracleBackend()).double()
        model.load_state_dict(ref.state_dict())
        part = model.partition_for(ei, n)
        parts = [make_partition(ei, n, world, r) for r in range(world)]
        # every rank draws its masks from a generator of its own: replay all of them
        gens = [torch.Generator().manual_seed(torch.initial_seed() + 7919 * (r + 1)) for r in range(world)]
        for step in range(3):
            masks = [torch.cat([torch.rand((parts[r].n_local, cols), generator=gens[r]) >= p
                                for r in range(world)]) for cols in (9, 32, 32)]
            assert not torch.equal(masks[0][:parts[0].n_local][:5], masks[0][parts[1].lo:][:5])
            drop = lambda t, m: t * m.double() / (1.0 - p)
            ref.train(); ref.zero_grad()
            h1 = ref.bn1(F.elu(ref.conv1(drop(x, masks[0]), ei)))
            h2 = ref.bn2(F.elu(ref.conv2(drop(h1, masks[1]), ei)))
            ref_logp = F.log_softmax(ref.conv3(drop(h2, masks[2]), ei), dim=1)
            ref_loss = F.nll_loss(ref_logp, y)
            ref_loss.backward()
            model.train(); model.zero_grad()
            # conv1's input is a fresh dropout output every step (exchange "as": its cached
            # source table must not survive from the step before)
            logp = model(x[part.lo:part.hi], ei, n)
            share = model.loss_share(logp, y[part.lo:part.hi], n)
            share.backward()
            model.sync_grads()
            tol = dict(rtol=1e-8, atol=1e-10)
            torch.testing.assert_close(logp.detach(), ref_logp.detach()[part.lo:part.hi], **tol)
            torch.testing.assert_close(model.global_loss(share), ref_loss.detach(), **tol)
            rp = dict(ref.named_parameters())
            for name, prm in model.named_parameters():
                torch.testing.assert_close(prm.grad, rp[name].grad, **tol,
                                           msg=lambda m, name=name: f"step {step} {name}: {m}")
            with torch.no_grad():               # a plain SGD step on both, so the steps differ
                for name, prm in model.named_parameters():
                    prm.sub_(0.1 * prm.grad); rp[name].sub_(0.1 * rp[name].grad)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,port", [(2, 29651), (3, 29652)])
def test_partitioned_network_with_feature_dropout(world, port):
    """Three training steps with feature dropout 0.5: every rank draws its own masks (replayed
    here from the ranks' generator seeds), conv1's input is a new tensor each step -- the oracle
    network fed the same masks gives the same log-probabilities, loss and gradients."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dropout_model_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
