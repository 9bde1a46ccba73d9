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
    assert float((out.cpu() - ref).abs().max()) < 2e-2      # x itself carries ~3e-5 of rounding at 500


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
        losses.append(float(loss))
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
