"""GATConv: the layer the reference constructs at /root/reference/gat/model.py:26,28,30
and calls at :40,:43,:46 -- same constructor arguments, same forward(x, edge_index)
signature, same parameter names as torch-geometric 2.5.x (att_src, att_dst, bias,
lin.weight) -- computed by the HIP kernels of libgat_mi355.so.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor, nn

from . import _lib, ops
from .graph import DEFAULT_SEG_LEN, GraphLayout, cached_graph


class _GATConvFn(torch.autograd.Function):
    """projection -> fused attention/softmax/aggregation, and their backward."""

    @staticmethod
    def forward(ctx, x, weight, att_src, att_dst, bias, layout: GraphLayout, cfg: ops.AttnConfig):
        n, f = x.shape
        if layout.num_nodes != n or layout.num_dst != n:
            raise ValueError(f"graph has {layout.num_nodes} nodes but x has {n} rows")
        x = x.contiguous()
        weight = weight.contiguous()
        att_s = att_src.contiguous().view(-1)
        att_d = att_dst.contiguous().view(-1)
        bias_c = None if bias is None else bias.contiguous()
        need_grad = any(ctx.needs_input_grad[:5])
        if cfg.feat_bf16:
            hp, a_dst = ops.project_fwd_bf16(x, weight, att_d, cfg.heads, cfg.channels)
        else:
            hp, a_dst = ops.project_fwd(x, weight, att_d, cfg.heads, cfg.channels)
        out, lse = ops.attn_fwd(layout, cfg, f, hp, a_dst, att_s, att_d, bias_c, need_grad)
        if need_grad:
            ctx.save_for_backward(x, weight, att_s, att_d, bias_c, hp, a_dst, out, lse)
            ctx.layout, ctx.cfg, ctx.att_shape = layout, cfg, att_src.shape
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, weight, att_s, att_d, bias_c, hp, a_dst, out, lse = ctx.saved_tensors
        cfg, layout = ctx.cfg, ctx.layout
        d_hp, d_as, d_ad, d_b, d_a_dst = ops.attn_bwd(
            layout, cfg, x.size(1), hp, a_dst, att_s, att_d, bias_c, out, lse,
            grad_out.contiguous(), defer_dst_term=True)
        bwd = ops.project_bwd_bf16 if cfg.feat_bf16 else ops.project_bwd
        d_w, d_x = bwd(x, weight, d_hp, cfg.heads, cfg.channels, ctx.needs_input_grad[0], d_a_dst, att_d)
        return d_x, d_w, d_as.view(ctx.att_shape), d_ad.view(ctx.att_shape), d_b, None, None


class _GATConvInAggFn(torch.autograd.Function):
    """The same layer for an input that needs no gradient and is narrower than the output:
    aggregate the x rows per head first, project afterwards (gat_amd/csrc/inagg.hip)."""

    @staticmethod
    def forward(ctx, x, weight, att_src, att_dst, bias, layout: GraphLayout, cfg: ops.AttnConfig):
        n = x.size(0)
        if layout.num_nodes != n or layout.num_dst != n:
            raise ValueError(f"graph has {layout.num_nodes} nodes but x has {n} rows")
        x = x.contiguous()
        weight = weight.contiguous()
        att_s = att_src.contiguous().view(-1)
        att_d = att_dst.contiguous().view(-1)
        bias_c = None if bias is None else bias.contiguous()
        need_grad = any(ctx.needs_input_grad[1:5])
        # bf16 source rows for the gathers (cfg.rows16) come out of the pack kernel with the f32 table,
        # which stays for the row-local terms
        if getattr(cfg, "rows16", False):
            xs, a_dst, xs16 = ops.inagg_pack(x, weight, att_s, att_d, cfg.heads, cfg.channels, rows16=True)
        else:
            (xs, a_dst), xs16 = ops.inagg_pack(x, weight, att_s, att_d, cfg.heads, cfg.channels), None
        out, z, lse = ops.inagg_fwd(layout, cfg, xs, a_dst, weight, att_s, att_d, bias_c, need_grad, xs16)
        if need_grad:
            ctx.save_for_backward(xs, a_dst, weight, att_s, att_d, bias_c, z, lse, xs16)
            ctx.layout, ctx.cfg, ctx.att_shape = layout, cfg, att_src.shape
        return out

    @staticmethod
    def backward(ctx, grad_out):
        xs, a_dst, weight, att_s, att_d, bias_c, z, lse, xs16 = ctx.saved_tensors
        d_w, d_as, d_ad, d_b = ops.inagg_bwd(ctx.layout, ctx.cfg, xs, a_dst, weight, att_s, att_d,
                                             bias_c, z, lse, grad_out.contiguous(), xs16)
        return None, d_w, d_as.view(ctx.att_shape), d_ad.view(ctx.att_shape), d_b, None, None


def _glorot_(t: Tensor) -> Tensor:
    bound = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        return t.uniform_(-bound, bound)


class GATConv(nn.Module):
    """Graph attention layer (Velickovic et al.) with the constructor the reference
    uses -- GATConv(in, out, heads=8, dropout=p) and GATConv(in, out, heads=1,
    concat=True, dropout=p) at model.py:26,28,30 -- and torch-geometric's defaults for
    everything else (negative_slope=0.2, add_self_loops=True, bias=True)."""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, concat: bool = True,
                 negative_slope: float = 0.2, dropout: float = 0.0, add_self_loops: bool = True,
                 bias: bool = True, seg_len: int = DEFAULT_SEG_LEN,
                 input_aggregation: Optional[bool] = None, stream: Optional[bool] = None,
                 feature_dtype: Optional[torch.dtype] = None):
        super().__init__()
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.negative_slope, self.dropout = concat, negative_slope, dropout
        self.add_self_loops = add_self_loops
        self.seg_len = seg_len
        # None: aggregate x before projecting whenever that is possible and cheaper (x needs
        # no gradient, F < H*Cp); True/False force the choice (True raises when impossible)
        self.input_aggregation = input_aggregation
        # input-side aggregation forward: None = by table size, True/False force the LDS-DMA
        # streaming kernel / the register pipeline (tests, A/B runs)
        self.stream = stream
        # torch.bfloat16: on the projected-row path keep the projected rows as a bf16 table and run
        # the projection on the bf16 matrix cores (f32 accumulation, f32 softmax and sums); on the
        # input-side path the edge kernels gather a bf16 copy of the packed x rows (logits, Z, q and
        # outputs f32).  The tolerance is stated separately: tests/test_gpu_bf16.py.  Shapes the
        # bf16 kernels do not cover (ops.project_bf16_supported, ops.inagg_rows16_supported) run
        # in f32 as before.
        if feature_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError(f"feature_dtype must be None, torch.float32 or torch.bfloat16, got {feature_dtype}")
        self.feature_dtype = feature_dtype
        self.lin = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(heads * out_channels if concat else out_channels))
        else:
            self.register_parameter("bias", None)
        self.injected_mask: Optional[Tensor] = None  # test hook, [E', H] in normalised edge order
        # One-element int64 CUDA tensor, or None.  Set: the attention-dropout seed lives on the
        # device -- advanced in place before every training forward, read by the kernels from
        # there (GATMI_MASK_HASH_DEV) -- so a step captured in a HIP graph draws a new mask at
        # every replay.  None: a fresh seed from torch's CPU generator per call (a by-value
        # kernel argument, which a graph would freeze).
        self.device_seed: Optional[Tensor] = None
        self.reset_parameters()

    def reset_parameters(self) -> None:
        _glorot_(self.lin.weight)
        _glorot_(self.att_src)
        _glorot_(self.att_dst)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        if x.dim() != 2 or x.size(1) != self.in_channels:
            raise ValueError(f"x must be [N, {self.in_channels}], got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("gat_amd.GATConv runs on the GPU only (no CPU fallback); "
                               "move the model and data to the MI355X device")
        if x.dtype != torch.float32:
            raise TypeError(f"x must be float32, got {x.dtype}")
        layout = cached_graph(edge_index, x.size(0), self.add_self_loops, self.seg_len)
        mask_mode, seed, mask = _lib.MASK_NONE, 0, None
        if self.training and self.dropout > 0.0:
            if self.injected_mask is not None:
                mask_mode, mask = _lib.MASK_BYTES, layout.edge_mask_to_csr(self.injected_mask)
            elif self.device_seed is not None:
                from .glue import SEED_STEP
                self.device_seed.add_(SEED_STEP)
                mask_mode, mask = _lib.MASK_HASH_DEV, self.device_seed
            else:
                mask_mode = _lib.MASK_HASH
                seed = int(torch.empty((), dtype=torch.int64).random_().item())
        fused_bias = self.bias if self.concat else None
        cfg = ops.AttnConfig(self.heads, self.out_channels, self.negative_slope, self.dropout,
                             mask_mode, seed, mask)
        cfg.stream = 0 if self.stream is None else (1 if self.stream else 2)
        x_grad = x.requires_grad and torch.is_grad_enabled()
        use_inagg = self.input_aggregation
        if use_inagg is None:
            use_inagg = not x_grad and ops.inagg_supported(
                self.in_channels, self.heads, self.out_channels,
                layout.num_edges / max(layout.num_dst, 1))
        elif use_inagg and x_grad:
            raise RuntimeError("input_aggregation=True cannot produce a gradient for x")
        cfg.feat_bf16 = (not use_inagg and self.feature_dtype == torch.bfloat16 and
                         ops.project_bf16_supported(self.in_channels, self.heads, self.out_channels))
        # the same switch on the input-side path: the edge kernels gather bf16 copies of the x rows
        cfg.rows16 = bool(use_inagg and self.feature_dtype == torch.bfloat16 and
                          ops.inagg_rows16_supported(self.in_channels, self.heads, self.out_channels))
        fn = _GATConvInAggFn if use_inagg else _GATConvFn
        out = fn.apply(x, self.lin.weight, self.att_src, self.att_dst, fused_bias, layout, cfg)
        if not self.concat:
            out = out.view(-1, self.heads, self.out_channels).mean(dim=1)
            if self.bias is not None:
                out = out + self.bias
        return out

    def extra_repr(self) -> str:
        return f"{self.in_channels}, {self.out_channels}, heads={self.heads}"
