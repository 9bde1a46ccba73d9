"""GATConv: the layer the reference constructs at /root/reference/gat/model.py:26,28,30
and calls at :40,:43,:46 -- same constructor arguments, same forward(x, edge_index)
signature, same parameter names as torch-geometric 2.5.x (att_src, att_dst, bias,
lin.weight) -- computed by the HIP kernels of libgat_mi355.so.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch
from torch import Tensor, nn

from . import _lib
from .graph import DEFAULT_SEG_LEN, GraphLayout, cached_graph


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _ws(nbytes: int, dev) -> Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


class _AttnArgs:
    """Owns the ctypes descriptor plus the tensors it points into."""

    def __init__(self, shape, negative_slope, p, mask_mode, seed, mask, hp, a_dst, att_src,
                 att_dst, bias):
        self.keep = (mask, hp, a_dst, att_src, att_dst, bias)
        a = _lib.Attn()
        a.shape = shape
        a.negative_slope = negative_slope
        a.dropout_p = p
        a.mask_mode = mask_mode
        a.seed = seed
        a.mask = _ptr(mask)
        a.hp, a.a_dst = _ptr(hp), _ptr(a_dst)
        a.att_src, a.att_dst, a.bias = _ptr(att_src), _ptr(att_dst), _ptr(bias)
        self.c = a


class _GATConvFn(torch.autograd.Function):
    """projection -> fused attention/softmax/aggregation, and their backward."""

    @staticmethod
    def forward(ctx, x, weight, att_src, att_dst, bias, layout: GraphLayout, heads, channels,
                negative_slope, p, mask_mode, seed, mask):
        lib = _lib.lib()
        dev = x.device
        n, f = x.shape
        if layout.num_nodes != n:
            raise ValueError(f"graph has {layout.num_nodes} nodes but x has {n} rows")
        x = x.contiguous()
        weight = weight.contiguous()
        att_s = att_src.contiguous().view(-1)
        att_d = att_dst.contiguous().view(-1)
        bias_c = None if bias is None else bias.contiguous()
        shape = _lib.Shape(n, f, heads, channels)
        cp = (channels + 3) // 4 * 4
        need_grad = any(ctx.needs_input_grad[:5])
        with torch.cuda.device(dev):
            st = _stream(dev)
            hp = torch.empty(n, heads * cp, dtype=torch.float32, device=dev)
            a_dst = torch.empty(n, heads, dtype=torch.float32, device=dev)
            ws = _ws(lib.gatmi_project_workspace_bytes(C.byref(shape)), dev)
            _lib.check(lib.gatmi_project_fwd(C.byref(shape), x.data_ptr(), weight.data_ptr(),
                                             att_d.data_ptr(), hp.data_ptr(), a_dst.data_ptr(),
                                             ws.data_ptr(), ws.numel(), st), "gatmi_project_fwd")
            out = torch.empty(n, heads * channels, dtype=torch.float32, device=dev)
            lse = torch.empty(n, heads, dtype=torch.float32, device=dev) if need_grad else None
            args = _AttnArgs(shape, negative_slope, p, mask_mode, seed, mask, hp, a_dst, att_s,
                             att_d, bias_c)
            ws2 = _ws(lib.gatmi_attn_fwd_workspace_bytes(layout.ref, C.byref(shape)), dev)
            _lib.check(lib.gatmi_attn_fwd(layout.ref, C.byref(args.c), out.data_ptr(), _ptr(lse),
                                          ws2.data_ptr(), ws2.numel(), st), "gatmi_attn_fwd")
        if need_grad:
            ctx.save_for_backward(x, weight, att_s, att_d, bias_c, hp, a_dst, out, lse, mask)
            ctx.layout = layout
            ctx.cfg = (heads, channels, negative_slope, p, mask_mode, seed)
            ctx.att_shape = att_src.shape
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.lib()
        x, weight, att_s, att_d, bias_c, hp, a_dst, out, lse, mask = ctx.saved_tensors
        heads, channels, negative_slope, p, mask_mode, seed = ctx.cfg
        layout = ctx.layout
        dev = x.device
        n, f = x.shape
        shape = _lib.Shape(n, f, heads, channels)
        grad_out = grad_out.contiguous()
        with torch.cuda.device(dev):
            st = _stream(dev)
            d_hp = torch.empty_like(hp)
            d_as = torch.empty_like(att_s)
            d_ad = torch.empty_like(att_d)
            d_b = torch.empty_like(bias_c) if bias_c is not None else None
            args = _AttnArgs(shape, negative_slope, p, mask_mode, seed, mask, hp, a_dst, att_s,
                             att_d, bias_c)
            ws = _ws(lib.gatmi_attn_bwd_workspace_bytes(layout.ref, C.byref(shape)), dev)
            _lib.check(lib.gatmi_attn_bwd(layout.ref, C.byref(args.c), out.data_ptr(),
                                          lse.data_ptr(), grad_out.data_ptr(), d_hp.data_ptr(),
                                          d_as.data_ptr(), d_ad.data_ptr(), _ptr(d_b),
                                          ws.data_ptr(), ws.numel(), st), "gatmi_attn_bwd")
            d_w = torch.empty_like(weight)
            d_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
            ws2 = _ws(lib.gatmi_project_workspace_bytes(C.byref(shape)), dev)
            _lib.check(lib.gatmi_project_bwd(C.byref(shape), x.data_ptr(), weight.data_ptr(),
                                             d_hp.data_ptr(), d_w.data_ptr(), _ptr(d_x),
                                             ws2.data_ptr(), ws2.numel(), st), "gatmi_project_bwd")
        return (d_x, d_w, d_as.view(ctx.att_shape), d_ad.view(ctx.att_shape), d_b,
                None, None, None, None, None, None, None, None)


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
                 bias: bool = True, seg_len: int = DEFAULT_SEG_LEN):
        super().__init__()
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.negative_slope, self.dropout = concat, negative_slope, dropout
        self.add_self_loops = add_self_loops
        self.seg_len = seg_len
        self.lin = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(heads * out_channels if concat else out_channels))
        else:
            self.register_parameter("bias", None)
        self.injected_mask: Optional[Tensor] = None  # test hook, [E', H] in normalised edge order
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
            else:
                mask_mode = _lib.MASK_HASH
                seed = int(torch.empty((), dtype=torch.int64).random_().item())
        fused_bias = self.bias if self.concat else None
        out = _GATConvFn.apply(x, self.lin.weight, self.att_src, self.att_dst, fused_bias, layout,
                               self.heads, self.out_channels, float(self.negative_slope),
                               float(self.dropout), mask_mode, seed, mask)
        if not self.concat:
            out = out.view(-1, self.heads, self.out_channels).mean(dim=1)
            if self.bias is not None:
                out = out + self.bias
        return out

    def extra_repr(self) -> str:
        return f"{self.in_channels}, {self.out_channels}, heads={self.heads}"
