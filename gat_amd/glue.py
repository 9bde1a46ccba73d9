"""ELU -> BatchNorm1d -> Dropout between two GAT layers, fused (gat_amd/csrc/glue.hip).

The reference writes it as three calls, /root/reference/gat/model.py:40-42 and :43-45:
    x = F.elu(conv(x, edge_index)); x = self.bn(x); x = self.dropout(x)
`elu_bn_dropout(x, bn, p)` computes the same thing from the same `nn.BatchNorm1d` module (its
weight, bias and running statistics stay where checkpoints expect them) in two passes over the
activations forward and two backward.  GPU only, fp32.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import Tensor, nn

from . import _lib


def _desc(rows, cols, training, eps, momentum, p, seed, gamma, beta, rmean, rvar):
    g = _lib.Glue()
    g.rows, g.cols, g.training = rows, cols, int(training)
    g.eps, g.momentum, g.dropout_p = float(eps), float(momentum), float(p)
    if isinstance(seed, Tensor):       # a device-resident seed (HIP-graph replays): pass its address
        g.seed_on_device, g.seed = 1, seed.data_ptr()
    else:
        g.seed_on_device, g.seed = 0, int(seed) & 0xFFFFFFFFFFFFFFFF
    g.gamma, g.beta = gamma.data_ptr(), beta.data_ptr()
    g.running_mean = None if rmean is None else rmean.data_ptr()
    g.running_var = None if rvar is None else rvar.data_ptr()
    return g


def _ws(cols, dev):
    return torch.empty(int(_lib.lib().gatmi_glue_workspace_bytes(cols)), dtype=torch.uint8, device=dev)


class _GlueFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, training, eps, momentum, p, seed):
        lib, dev = _lib.lib(), x.device
        x, gamma, beta = x.contiguous(), gamma.contiguous(), beta.contiguous()
        n, d = x.shape
        with torch.cuda.device(dev):
            y = torch.empty_like(x)
            mean = torch.empty(d, dtype=torch.float32, device=dev)
            rstd = torch.empty(d, dtype=torch.float32, device=dev)
            ws = _ws(d, dev)
            g = _desc(n, d, training, eps, momentum, p, seed, gamma, beta, rmean, rvar)
            _lib.check(lib.gatmi_glue_fwd(C.byref(g), x.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                          rstd.data_ptr(), ws.data_ptr(), ws.numel(),
                                          torch.cuda.current_stream(dev).cuda_stream), "gatmi_glue_fwd")
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.cfg = (training, eps, momentum, p, seed)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        training, eps, momentum, p, seed = ctx.cfg
        lib, dev = _lib.lib(), x.device
        n, d = x.shape
        with torch.cuda.device(dev):
            gx = torch.empty_like(x)
            dg, db = torch.empty_like(gamma), torch.empty_like(beta)
            ws = _ws(d, dev)
            # running statistics are not touched by the backward: pass the saved ones as such
            g = _desc(n, d, training, eps, momentum, p, seed, gamma, beta, mean, rstd)
            _lib.check(lib.gatmi_glue_bwd(C.byref(g), x.data_ptr(), grad_y.contiguous().data_ptr(),
                                          mean.data_ptr(), rstd.data_ptr(), gx.data_ptr(),
                                          dg.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(),
                                          torch.cuda.current_stream(dev).cuda_stream), "gatmi_glue_bwd")
        return gx, dg, db, None, None, None, None, None, None, None


SEED_STEP = 0x1E3779B97F4A7C15      # what a device-resident seed advances by per use (odd, < 2^63)


def elu_bn_dropout(x: Tensor, bn: nn.BatchNorm1d, p: float, training: bool, seed=None) -> Tensor:
    """dropout(bn(elu(x))) with `bn`'s parameters and running statistics (updated in training,
    like the module would).  `seed` fixes the dropout mask (tests); default: drawn from torch's
    generator.  `seed` may be a one-element int64 CUDA tensor: it is advanced in place on the
    device and read by the kernels from there, so the call can be captured in a HIP graph and
    every replay draws a new mask."""
    if not x.is_cuda:
        raise RuntimeError("gat_amd.glue runs on the GPU only (no CPU fallback)")
    if x.dim() != 2 or x.size(1) != bn.num_features:
        raise ValueError(f"x must be [N, {bn.num_features}], got {tuple(x.shape)}")
    if x.dtype != torch.float32:
        raise TypeError(f"x must be float32, got {x.dtype}")
    if not bn.affine or not bn.track_running_stats or bn.momentum is None:
        raise ValueError("elu_bn_dropout covers nn.BatchNorm1d's defaults (affine, running statistics, "
                         "momentum), as the reference constructs it at model.py:27,29")
    drop = training and p > 0.0
    if isinstance(seed, Tensor):
        if drop:
            seed.add_(SEED_STEP)
        else:
            seed = 0
    elif seed is None:
        seed = int(torch.empty((), dtype=torch.int64).random_().item()) if drop else 0
    if training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return _GlueFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.eps,
                         bn.momentum, p if training else 0.0, seed)


class _RowPartitionGlueFn(torch.autograd.Function):
    """dropout(bn(elu(x))) on this rank's rows of a row-partitioned batch: BatchNorm's statistics and
    the backward's two sums are those of ALL ranks' rows (gatmi_glue_stats / _apply / _bwd_sums /
    _bwd_apply with a collective in between).  `exchange` supplies the collectives:
    exchange.gather(vec) -> [world, len(vec)] (every rank's vector), exchange.sum_(t) (in place)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, training, eps, momentum, p, seed, total_rows, exchange):
        lib, dev = _lib.lib(), x.device
        x, gamma, beta = x.contiguous(), gamma.contiguous(), beta.contiguous()
        n, d = x.shape
        st = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            y = torch.empty_like(x)
            g = _desc(n, d, training, eps, momentum, p, seed, gamma, beta, rmean, rvar)
            if training:
                mean_l = torch.empty(d, dtype=torch.float32, device=dev)
                m2_l = torch.empty(d, dtype=torch.float32, device=dev)
                ws = _ws(d, dev)
                _lib.check(lib.gatmi_glue_stats(C.byref(g), x.data_ptr(), mean_l.data_ptr(), m2_l.data_ptr(),
                                                ws.data_ptr(), ws.numel(), st), "gatmi_glue_stats")
                # Chan's combination of the ranks' (rows, mean, M2), in float64, identically on every rank
                mine = torch.cat([torch.tensor([float(n)], dtype=torch.float64, device=dev),
                                  mean_l.double(), m2_l.double()])
                allv = exchange.gather(mine)                       # [world, 1 + 2 d]
                cnt, means, m2s = allv[:, :1], allv[:, 1:1 + d], allv[:, 1 + d:]
                tot = float(total_rows)
                mean64 = (cnt * means).sum(0) / tot
                m2 = (m2s + cnt * (means - mean64) ** 2).sum(0)
                var = m2 / tot
                mean = mean64.float()
                rstd = torch.rsqrt(var + eps).float()
                with torch.no_grad():        # nn.BatchNorm1d's update: unbiased variance, momentum
                    rmean.mul_(1 - momentum).add_(momentum * mean)
                    rvar.mul_(1 - momentum).add_((momentum * m2 / max(tot - 1.0, 1.0)).float())
            else:
                mean = rmean.clone()
                rstd = torch.rsqrt(rvar + eps)
            _lib.check(lib.gatmi_glue_apply(C.byref(g), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                            y.data_ptr(), st), "gatmi_glue_apply")
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.cfg = (training, eps, momentum, p, seed, total_rows, exchange)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        training, eps, momentum, p, seed, total_rows, exchange = ctx.cfg
        lib, dev = _lib.lib(), x.device
        n, d = x.shape
        st = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            gy = grad_y.contiguous()
            gx = torch.empty_like(x)
            sums = torch.empty(2, d, dtype=torch.float32, device=dev)
            ws = _ws(d, dev)
            g = _desc(n, d, training, eps, momentum, p, seed, gamma, beta, mean, rstd)
            _lib.check(lib.gatmi_glue_bwd_sums(C.byref(g), x.data_ptr(), gy.data_ptr(), mean.data_ptr(),
                                               rstd.data_ptr(), sums.data_ptr(), ws.data_ptr(), ws.numel(), st),
                       "gatmi_glue_bwd_sums")
            # this rank's share of d_beta / d_gamma goes to autograd (the caller all-reduces parameter
            # gradients once, with the others); the apply pass needs the sums over all rows now
            d_beta, d_gamma = sums[0].clone(), sums[1].clone()
            if training:
                exchange.sum_(sums)
            _lib.check(lib.gatmi_glue_bwd_apply(C.byref(g), int(total_rows), x.data_ptr(), gy.data_ptr(),
                                                mean.data_ptr(), rstd.data_ptr(), sums.data_ptr(),
                                                gx.data_ptr(), st), "gatmi_glue_bwd_apply")
        return gx, d_gamma, d_beta, None, None, None, None, None, None, None, None, None


def elu_bn_dropout_rows(x: Tensor, bn: nn.BatchNorm1d, p: float, training: bool, total_rows: int,
                        exchange, seed=None) -> Tensor:
    """elu_bn_dropout for one rank's rows of a row-partitioned batch of `total_rows` rows
    (gat_amd.dist.PartitionedGAT): same module, same results as one process on all rows would get.
    The gradients of bn.weight / bn.bias come back as this rank's partial sums."""
    if not x.is_cuda:
        raise RuntimeError("gat_amd.glue runs on the GPU only (no CPU fallback)")
    if x.dim() != 2 or x.size(1) != bn.num_features or x.dtype != torch.float32:
        raise ValueError(f"x must be float32 [n, {bn.num_features}], got {x.dtype} {tuple(x.shape)}")
    if not bn.affine or not bn.track_running_stats or bn.momentum is None:
        raise ValueError("elu_bn_dropout_rows covers nn.BatchNorm1d's defaults (affine, running statistics, momentum)")
    if total_rows < x.size(0) or total_rows < 1:
        raise ValueError(f"{x.size(0)} local rows of {total_rows} in all")
    drop = training and p > 0.0
    if seed is None:
        seed = int(torch.empty((), dtype=torch.int64).random_().item()) if drop else 0
    if training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return _RowPartitionGlueFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.eps,
                                     bn.momentum, p if training else 0.0, seed, int(total_rows), exchange)


class _TailFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        lib, dev = _lib.lib(), logits.device
        logits, target = logits.contiguous(), target.contiguous()
        n, k = logits.shape
        with torch.cuda.device(dev):
            logp = torch.empty_like(logits)
            stats = torch.empty(2, dtype=torch.float32, device=dev)
            ws = torch.empty(int(lib.gatmi_tail_workspace_bytes()), dtype=torch.uint8, device=dev)
            _lib.check(lib.gatmi_tail_fwd(n, k, logits.data_ptr(), target.data_ptr(), logp.data_ptr(),
                                          stats.data_ptr(), ws.data_ptr(), ws.numel(),
                                          torch.cuda.current_stream(dev).cuda_stream), "gatmi_tail_fwd")
        ctx.save_for_backward(logp, target, stats)
        return logp, stats[0]

    @staticmethod
    def backward(ctx, grad_logp, grad_loss):
        logp, target, stats = ctx.saved_tensors
        lib, dev = _lib.lib(), logp.device
        n, k = logp.shape
        with torch.cuda.device(dev):
            gx = torch.empty_like(logp)
            gl = None if grad_loss is None else grad_loss.to(torch.float32).contiguous()
            gp = None if grad_logp is None else grad_logp.contiguous()
            _lib.check(lib.gatmi_tail_bwd(n, k, logp.data_ptr(), target.data_ptr(), stats.data_ptr(),
                                          None if gl is None else gl.data_ptr(),
                                          None if gp is None else gp.data_ptr(), gx.data_ptr(),
                                          torch.cuda.current_stream(dev).cuda_stream), "gatmi_tail_bwd")
        return gx, None


def log_softmax_nll(logits: Tensor, target: Tensor):
    """(F.log_softmax(logits, dim=1), F.nll_loss(of that, target)) -- /root/reference/gat/model.py:47
    and :53 -- in one pass forward and one backward (gat_amd/csrc/tail.hip).  GPU only, fp32
    logits [N, classes <= 256], int64 targets."""
    if not logits.is_cuda or not target.is_cuda:
        raise RuntimeError("gat_amd.glue runs on the GPU only (no CPU fallback)")
    if logits.dim() != 2 or target.dim() != 1 or target.size(0) != logits.size(0):
        raise ValueError(f"logits must be [N, K] and target [N], got {tuple(logits.shape)}, {tuple(target.shape)}")
    if logits.dtype != torch.float32 or target.dtype != torch.int64:
        raise TypeError(f"logits must be float32 and target int64, got {logits.dtype}, {target.dtype}")
    if logits.size(1) > 256:
        raise ValueError("log_softmax_nll covers up to 256 classes")
    return _TailFn.apply(logits, target)


def dropout_mask(rows: int, cols: int, p: float, seed: int, device) -> Tensor:
    """The keep-mask elu_bn_dropout uses for (p, seed), as a bool tensor [rows, cols]."""
    mask = torch.empty(rows, cols, dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.lib().gatmi_glue_mask(rows, cols, float(p), int(seed), mask.data_ptr(),
                                              torch.cuda.current_stream(device).cuda_stream),
                   "gatmi_glue_mask")
    return mask.bool()
