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
        h_loc = hp_full[part.lo:part.hi].view(-1, cfg.heads, cfg.channels)
        d_att_dst = (d_a_dst.unsqueeze(-1) * h_loc).sum(0).reshape(-1)
        return leaves[0].grad, leaves[2].grad, d_att_dst, leaves[3].grad, d_a_dst

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
            a_d = (hh[part.lo:part.hi] * adl.view(1, h_, c_)).sum(-1)
            self._inagg_out(layout, cfg, x_full, a_src, a_d, wl, bl).backward(grad_out)
        return wl.grad, asl.grad, adl.grad, bl.grad


def _worker(rank, world, port, n, gname, exchange, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gat_amd.dist import Partition, PartitionedGATConv
        torch.set_num_threads(1)
        ei, n_ = cases.GRAPHS[gname]() if n is None else cases.random_multigraph(n=n, e=6 * n)
        f, h, c = 6, 3, 5
        x, w, a_s, a_d, b, go = [t.double() for t in cases.layer_inputs(n_, f, h, c)]
        part = Partition(n_, world, rank)
        conv = PartitionedGATConv(f, c, heads=h, backend=OracleBackend(), exchange=exchange).double().eval()
        with torch.no_grad():
            conv.lin.weight.copy_(w); conv.att_src.copy_(a_s); conv.att_dst.copy_(a_d); conv.bias.copy_(b)
        inagg = exchange in ("xs", "as")
        xl = x[part.lo:part.hi].clone().requires_grad_(not inagg)
        steps = 2 if exchange == "as" else 1     # second step: cached rows, fresh logits
        for step in range(steps):
            conv.zero_grad()
            if step == 1:                         # the weights moved: a_src must be re-gathered
                with torch.no_grad():
                    w = w * 1.25; conv.lin.weight.copy_(w)
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


def _run(world, n, gname, exchange, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, gname, exchange, q)) for r in range(world)]
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


def test_local_edges_keep_reference_order():
    from gat_amd.dist import Partition, local_edges
    ei, n = cases.random_multigraph()
    norm = O.normalize_edges(ei, n)
    for rank in range(3):
        part = Partition(n, 3, rank)
        loc = local_edges(ei, part)
        sel = (norm[1] >= part.lo) & (norm[1] < part.hi)
        assert torch.equal(loc[0], norm[0][sel])
        assert torch.equal(loc[1] + part.lo, norm[1][sel])
    assert sum(Partition(n, 3, r).n_local for r in range(3)) == n


def _model_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gat_amd.dist import Partition, PartitionedGAT
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
        part = Partition(n, world, rank)
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
