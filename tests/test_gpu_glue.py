"""GPU parity of the fused inter-layer glue (gat_amd/csrc/glue.hip, C ABI gatmi_glue_*) with
the reference's three calls  F.elu -> BatchNorm1d -> Dropout  (/root/reference/gat/model.py:
40-42) restated in CPU torch (oracle.gat_oracle.elu_bn_dropout_ref), the dropout draw replaced
by the materialised hash mask.  fp32, rtol 1e-5, atol 1e-5 x max|ref|."""
import copy

import pytest
import torch
from torch import nn

from oracle import gat_oracle as O

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def close(got, ref, name, floor=0.0):
    """`floor`: size of one summand, for results that are differences of nearly equal sums (the
    BatchNorm backward of a tiny batch: with two rows d_x is exactly 0 in exact arithmetic)."""
    got, ref = got.detach().cpu(), ref.detach().cpu()
    atol = RTOL * max(float(ref.abs().max()), floor, 1e-30)
    torch.testing.assert_close(got, ref, rtol=RTOL, atol=atol, msg=lambda m: f"{name}: {m}")


def make_bn(d, seed):
    g = torch.Generator().manual_seed(seed)
    bn = nn.BatchNorm1d(d)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(d, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(d, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(d, generator=g) * 0.2)
        bn.running_var.copy_(torch.rand(d, generator=g) + 0.5)
    return bn


@pytest.mark.parametrize("n,d,p,training", [
    (1000, 240, 0.425, True), (1000, 240, 0.0, True), (333, 256, 0.3, True), (64, 7, 0.5, True),
    (5000, 64, 0.425, True), (2, 16, 0.0, True), (1000, 240, 0.425, False), (257, 30, 0.0, False),
    (70000, 240, 0.425, True)])
def test_glue_matches_torch(cuda, n, d, p, training):
    from gat_amd.glue import dropout_mask, elu_bn_dropout
    g = torch.Generator().manual_seed(n + d)
    x = torch.randn(n, d, generator=g) * 1.5 + 0.3        # a mean offset: exercises the shifted sums
    gy = torch.randn(n, d, generator=g)
    bn_ref = make_bn(d, 7).train(training)
    bn_dev = copy.deepcopy(bn_ref).to(cuda).train(training)
    seed = 1234567 + n
    mask = dropout_mask(n, d, p, seed, cuda).cpu() if (training and p > 0) else None
    xr = x.clone().requires_grad_(True)
    ref = O.elu_bn_dropout_ref(xr, bn_ref, mask, p)
    ref.backward(gy)
    xd = x.to(cuda).requires_grad_(True)
    out = elu_bn_dropout(xd, bn_dev, p, training, seed=seed)
    out.backward(gy.to(cuda))
    close(out, ref, "y")
    e = torch.nn.functional.elu(x)
    rstd = 1.0 / torch.sqrt((e.var(0, unbiased=False) if training else bn_ref.running_var) + bn_ref.eps)
    close(xd.grad, xr.grad, "d_x",
          floor=float(gy.abs().max() * (bn_ref.weight.detach().abs() * rstd).max()))
    close(bn_dev.weight.grad, bn_ref.weight.grad, "d_gamma")
    close(bn_dev.bias.grad, bn_ref.bias.grad, "d_beta")
    close(bn_dev.running_mean, bn_ref.running_mean, "running_mean")
    close(bn_dev.running_var, bn_ref.running_var, "running_var")
    assert int(bn_dev.num_batches_tracked) == int(bn_ref.num_batches_tracked)
    if mask is not None:
        assert abs(float(mask.float().mean()) - (1 - p)) < 0.02


def test_glue_mean_offset_is_stable(cuda):
    """|mean| >> std: a sum / sum-of-squares variance would lose every digit here."""
    from gat_amd.glue import elu_bn_dropout
    n, d = 20000, 32
    g = torch.Generator().manual_seed(0)
    x = 500.0 + 0.01 * torch.randn(n, d, generator=g)
    bn_ref, bn_dev = make_bn(d, 3).train(), None
    bn_dev = copy.deepcopy(bn_ref).to(cuda).train()
    ref = O.elu_bn_dropout_ref(x.double(), copy.deepcopy(bn_ref).double(), None, 0.0).float()
    out = elu_bn_dropout(x.to(cuda), bn_dev, 0.0, True)
    assert float((out.detach().cpu() - ref.detach()).abs().max()) < 2e-2      # x itself carries ~3e-5 of rounding at 500


def test_model_with_fused_glue_matches_stock_glue(cuda):
    """The 3-layer network with the fused glue == the same network with the three stock ops
    (dropout off so that both draw no mask): loss and every gradient."""
    import cases
    from gat_amd.data import Data
    from gat_amd.model import GAT
    ei, n = cases.knn_graph(n=300, f=25, seed=5)
    g = torch.Generator().manual_seed(1)
    data = Data(torch.rand(n, 25, generator=g), ei, torch.randint(0, 2, (n,), generator=g))
    torch.manual_seed(3)
    a = GAT(torch.optim.SGD, 25, 2, dropout=0.0, hidden_dim=30, lr=0.05, device=cuda, verbose=False)
    b = copy.deepcopy(a)
    b.fused_glue = False
    losses = []
    for m in (a, b):
        m.train()
        out = m(data)
        loss = torch.nn.functional.nll_loss(out, data.y.to(cuda))
        loss.backward()
        losses.append(float(loss.detach()))
    assert abs(losses[0] - losses[1]) < 1e-5 * max(abs(losses[1]), 1.0)
    for (na, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        ref = pb.grad
        # a layer bias in front of elu -> BatchNorm: its gradient is a column sum of d_x, whose
        # BatchNorm part cancels exactly -- what is left is small against the summands
        scale = 2e-4 if na in ("conv1.bias", "conv2.bias") else 2e-5
        atol = scale * max(float(ref.abs().max()), 1e-6)
        torch.testing.assert_close(pa.grad, ref, rtol=2e-5, atol=atol, msg=lambda m: f"{na}: {m}")
    close(a.bn1.running_var, b.bn1.running_var, "bn1.running_var")


# ---- log_softmax + nll_loss tail (gat_amd/csrc/tail.hip, C ABI gatmi_tail_*) ------------------
@pytest.mark.parametrize("n,k", [(1000, 2), (1000, 7), (169, 40), (5000, 47), (3, 1), (70000, 40),
                                 (513, 64), (300, 100), (64, 256), (1, 5)])
def test_tail_matches_torch(cuda, n, k):
    """Forward log-probabilities and mean loss, backward through the loss alone and through
    loss + an extra gradient on the log-probabilities, against stock CPU torch ops."""
    from gat_amd.glue import log_softmax_nll
    g = torch.Generator().manual_seed(n * 131 + k)
    x = torch.randn(n, k, generator=g) * 3.0
    y = torch.randint(0, k, (n,), generator=g)
    extra = torch.randn(n, k, generator=g)
    for with_extra in (False, True):
        xr = x.clone().requires_grad_(True)
        logp_r, loss_r = O.log_softmax_nll_ref(xr, y)
        (loss_r * 1.7 + ((logp_r * extra).sum() if with_extra else 0.0)).backward()
        xd = x.to(cuda).requires_grad_(True)
        logp_d, loss_d = log_softmax_nll(xd, y.to(cuda))
        (loss_d * 1.7 + ((logp_d * extra.to(cuda)).sum() if with_extra else 0.0)).backward()
        close(logp_d, logp_r, "logp")
        torch.testing.assert_close(loss_d.cpu(), loss_r.detach(), rtol=2e-6, atol=1e-7)
        close(xd.grad, xr.grad, "d_logits", floor=float(extra.abs().max()) if with_extra else 1.7 / n)


def test_tail_ignores_out_of_range_targets(cuda):
    """torch's ignore_index (-100): not counted in the mean, zero gradient row."""
    from gat_amd.glue import log_softmax_nll
    g = torch.Generator().manual_seed(5)
    x = torch.randn(200, 6, generator=g)
    y = torch.randint(0, 6, (200,), generator=g)
    y[::7] = -100
    xr = x.clone().requires_grad_(True)
    _, loss_r = O.log_softmax_nll_ref(xr, y)
    loss_r.backward()
    xd = x.to(cuda).requires_grad_(True)
    _, loss_d = log_softmax_nll(xd, y.to(cuda))
    loss_d.backward()
    torch.testing.assert_close(loss_d.cpu(), loss_r.detach(), rtol=2e-6, atol=1e-7)
    close(xd.grad, xr.grad, "d_logits", floor=1.0 / 200)
    assert float(xd.grad[::7].abs().max()) == 0.0


class _ReplayExchange:
    """One process plays every rank of a row partition.  Round 0 over the ranks RECORDS what each
    contributes to the forward's all-gather; round 1 REPLAYS the gathered statistics and records the
    backward's sums (now taken with the right statistics); round 2 replays both -- what the
    collectives of gat_amd.dist._GlueExchange would have returned."""

    def __init__(self, world):
        self.world, self.rank, self.round = world, 0, 0
        self.vecs, self.sums = {}, {}

    def gather(self, vec):
        if self.round == 0:
            self.vecs[self.rank] = vec.clone()
            return vec.view(1, -1)                 # placeholder: only this rank's rows
        return torch.stack([self.vecs[r] for r in range(self.world)])

    def sum_(self, t):
        if self.round < 2:
            self.sums[self.rank] = t.clone()
            return t
        t.copy_(sum(self.sums[r] for r in range(self.world)))
        return t


@pytest.mark.parametrize("n,d,p,bounds", [(1000, 240, 0.425, (0, 300, 1000)), (777, 256, 0.0, (0, 5, 6, 777)),
                                          (333, 30, 0.3, (0, 0, 200, 333)), (5000, 64, 0.425, (0, 1250, 2500, 3750, 5000))])
def test_row_partition_glue_matches_torch_on_all_rows(cuda, n, d, p, bounds):
    """gatmi_glue_stats / _apply / _bwd_sums / _bwd_apply (PartitionedGAT's ELU -> BatchNorm -> dropout,
    VERDICT r2 #6): every rank's rows through the C ABI with the collectives replayed, against stock
    CPU torch on ALL rows -- y, d_x, the summed d_gamma / d_beta, the running statistics; a rank with
    a single row and a rank with none included."""
    from gat_amd.glue import dropout_mask, elu_bn_dropout_rows
    world = len(bounds) - 1
    g = torch.Generator().manual_seed(n + d)
    x = torch.randn(n, d, generator=g) * 1.5 + 0.3
    x[bounds[1]:] += 0.7                                   # the ranks' means differ: Chan's cross term matters
    gy = torch.randn(n, d, generator=g)
    bn_ref = make_bn(d, 9).train()
    seeds = [424242 + 17 * r for r in range(world)]
    mask = None
    if p > 0:
        mask = torch.cat([dropout_mask(bounds[r + 1] - bounds[r], d, p, seeds[r], cuda).cpu()
                          if bounds[r + 1] > bounds[r] else torch.zeros(0, d, dtype=torch.bool)
                          for r in range(world)])
    xr = x.clone().requires_grad_(True)
    bn_check = copy.deepcopy(bn_ref)
    ref = O.elu_bn_dropout_ref(xr, bn_check, mask, p)
    ref.backward(gy)
    ex = _ReplayExchange(world)
    outs, dxs, dgs, dbs, bns = [], [], 0, 0, []
    for ex.round in (0, 1, 2):
        for r in range(world):
            ex.rank = r
            bn_dev = copy.deepcopy(bn_ref).to(cuda).train()
            xd = x[bounds[r]:bounds[r + 1]].to(cuda).requires_grad_(True)
            out = elu_bn_dropout_rows(xd, bn_dev, p, True, n, ex, seed=seeds[r])
            out.backward(gy[bounds[r]:bounds[r + 1]].to(cuda))
            if ex.round == 2:
                outs.append(out.detach().cpu()); dxs.append(xd.grad.cpu()); bns.append(bn_dev)
                dgs = dgs + bn_dev.weight.grad.cpu(); dbs = dbs + bn_dev.bias.grad.cpu()
    close(torch.cat(outs), ref, "y (row partition)")
    e = torch.nn.functional.elu(x)
    rstd = 1.0 / torch.sqrt(e.var(0, unbiased=False) + bn_ref.eps)
    close(torch.cat(dxs), xr.grad, "d_x (row partition)",
          floor=float(gy.abs().max() * (bn_ref.weight.detach().abs() * rstd).max()))
    close(dgs, bn_check.weight.grad, "d_gamma (sum over ranks)")
    close(dbs, bn_check.bias.grad, "d_beta (sum over ranks)")
    for bn_dev in bns:                                     # identical on every rank, as on one process
        close(bn_dev.running_mean, bn_check.running_mean, "running_mean")
        close(bn_dev.running_var, bn_check.running_var, "running_var")
        assert int(bn_dev.num_batches_tracked) == int(bn_check.num_batches_tracked)


def test_row_partition_glue_eval_mode_is_local(cuda):
    from gat_amd.glue import elu_bn_dropout, elu_bn_dropout_rows
    bn = make_bn(48, 3).to(cuda).eval()
    x = torch.randn(200, 48, device=cuda)
    a = elu_bn_dropout(x, bn, 0.4, False)
    b = elu_bn_dropout_rows(x[:77].contiguous(), bn, 0.4, False, 200, _ReplayExchange(2))
    close(b, a[:77], "eval mode")        # rstd: torch.rsqrt here, 1 / sqrtf there
