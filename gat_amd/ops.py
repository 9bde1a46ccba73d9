"""Tensor-level wrappers of the C ABI (include/gat_mi355.h): each function allocates its
outputs and workspace with torch, passes raw device pointers and the current stream to
libgat_mi355.so, and raises RuntimeError on a non-zero status.  GPU only."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from .graph import GraphLayout


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _ws(nbytes: int, dev) -> Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


def padded_channels(channels: int) -> int:
    return (channels + 3) // 4 * 4


def _expect(t: Optional[Tensor], shape, dtype, name: str, dev=None) -> None:
    """Host-side operand check before a kernel launch: the kernels index by the sizes the
    descriptors state and do no bounds tests of their own, so a tensor that is shorter than stated
    would be a device memory fault, not an exception."""
    if t is None:
        return
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if not t.is_cuda or (dev is not None and t.device != dev):
        raise RuntimeError(f"{name} must be a tensor on {dev or 'the GPU'} (got {t.device}); gat_amd has no CPU path")


class AttnConfig:
    """Everything the fused attention kernels need besides tensors."""

    def __init__(self, heads: int, channels: int, negative_slope: float = 0.2, dropout: float = 0.0,
                 mask_mode: int = _lib.MASK_NONE, seed: int = 0, mask: Optional[Tensor] = None):
        self.heads, self.channels = heads, channels
        self.negative_slope, self.dropout = float(negative_slope), float(dropout)
        self.mask_mode, self.seed, self.mask = mask_mode, seed, mask
        self.stream = 0     # input-side aggregation forward: 0 auto, 1 streaming, 2 registers
        self.feat_bf16 = False   # projected-row path: hp is a bfloat16 table (attn_bf16_supported)

    def c_args(self, num_dst, in_features, hp, a_dst, att_src, att_dst, bias, num_src=None, num_edges=None):
        dev = a_dst.device
        hc = self.heads * self.channels
        if hp is not None:
            rows = hp.size(0) if num_src is None else num_src
            _expect(hp, (rows, self.heads * padded_channels(self.channels)),
                    torch.bfloat16 if self.feat_bf16 else torch.float32, "feature table", dev)
        _expect(a_dst, (num_dst, self.heads), torch.float32, "a_dst", dev)
        for name, t in (("att_src", att_src), ("att_dst", att_dst), ("bias", bias)):
            if t is not None:
                _expect(t.view(-1), (hc,), torch.float32, name, dev)
        if self.mask_mode == _lib.MASK_BYTES:
            if self.mask is None or num_edges is None:
                raise ValueError("mask_mode BYTES needs a mask and the layout's edge count")
            _expect(self.mask, (num_edges, self.heads), torch.uint8, "dropout mask", dev)
        if self.mask_mode == _lib.MASK_HASH_DEV:
            if self.mask is None or self.mask.numel() != 1 or self.mask.dtype != torch.int64 or not self.mask.is_cuda:
                raise ValueError("mask_mode HASH_DEV needs a one-element int64 device tensor (the seed)")
        a = _lib.Attn()
        a.shape = _lib.Shape(num_dst, in_features, self.heads, self.channels)
        a.negative_slope, a.dropout_p = self.negative_slope, self.dropout
        a.mask_mode, a.seed, a.mask = self.mask_mode, self.seed, _ptr(self.mask)
        if (hp.dtype == torch.bfloat16) != bool(self.feat_bf16):
            raise TypeError(f"feature table is {hp.dtype} but feat_bf16={self.feat_bf16}")
        a.feat_bf16 = int(bool(self.feat_bf16))
        a.hp, a.a_dst = _ptr(hp), _ptr(a_dst)
        a.att_src, a.att_dst, a.bias = _ptr(att_src), _ptr(att_dst), _ptr(bias)
        return a


def _alloc_table(n: int, cols: int, dtype, dev, alloc_rows: Optional[int]) -> Tensor:
    """[max(n, alloc_rows), cols]; the rows behind the first n (a shard padded for an equal-shard
    collective) are zero"""
    rows = n if alloc_rows is None else max(int(alloc_rows), n)
    t = torch.empty(rows, cols, dtype=dtype, device=dev)
    if rows > n:
        t[n:].zero_()
    return t


def _project_operands(x, weight, att_dst, heads, channels, d_hp=None, d_a_dst=None):
    n, f = x.shape
    dev = x.device
    _expect(x, (n, f), torch.float32, "x", dev)
    _expect(weight, (heads * channels, f), torch.float32, "weight", dev)
    if att_dst is not None:
        _expect(att_dst.view(-1), (heads * channels,), torch.float32, "att_dst", dev)
    _expect(d_hp, (n, heads * padded_channels(channels)), torch.float32, "d_hp", dev)
    _expect(d_a_dst, (n, heads), torch.float32, "d_a_dst", dev)


def project_fwd(x: Tensor, weight: Tensor, att_dst: Tensor, heads: int, channels: int,
                alloc_rows: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """h = lin(x) in the head-padded layout [N, H*Cp] and a_dst [N, H].  alloc_rows: the table is
    allocated with that many rows (>= N; the extra rows are zero)."""
    lib, dev = _lib.lib(), x.device
    n, f = x.shape
    _project_operands(x, weight, att_dst, heads, channels)
    shape = _lib.Shape(n, f, heads, channels)
    with torch.cuda.device(dev):
        hp = _alloc_table(n, heads * padded_channels(channels), torch.float32, dev, alloc_rows)
        a_dst = torch.empty(n, heads, dtype=torch.float32, device=dev)
        ws = _ws(lib.gatmi_project_workspace_bytes(C.byref(shape)), dev)
        _lib.check(lib.gatmi_project_fwd(C.byref(shape), x.data_ptr(), weight.data_ptr(),
                                         att_dst.data_ptr(), hp.data_ptr(), a_dst.data_ptr(),
                                         ws.data_ptr(), ws.numel(), _stream(dev)), "gatmi_project_fwd")
    return hp, a_dst


def project_bwd(x: Tensor, weight: Tensor, d_hp: Tensor, heads: int, channels: int,
                need_dx: bool, d_a_dst: Optional[Tensor] = None, att_dst: Optional[Tensor] = None
                ) -> Tuple[Tensor, Optional[Tensor]]:
    """d_weight, d_x.  With d_a_dst/att_dst the term d_a_dst * att_dst still missing from
    d_hp is added while its tiles are staged (see gatmi_project_bwd)."""
    lib, dev = _lib.lib(), x.device
    n, f = x.shape
    _project_operands(x, weight, att_dst, heads, channels, d_hp, d_a_dst)
    shape = _lib.Shape(n, f, heads, channels)
    with torch.cuda.device(dev):
        d_w = torch.empty_like(weight)
        d_x = torch.empty_like(x) if need_dx else None
        ws = _ws(lib.gatmi_project_workspace_bytes(C.byref(shape)), dev)
        _lib.check(lib.gatmi_project_bwd(C.byref(shape), x.data_ptr(), weight.data_ptr(),
                                         d_hp.data_ptr(), _ptr(d_a_dst), _ptr(att_dst),
                                         d_w.data_ptr(), _ptr(d_x), ws.data_ptr(), ws.numel(),
                                         _stream(dev)), "gatmi_project_bwd")
    return d_w, d_x


def project_bf16_supported(in_features: int, heads: int, channels: int) -> bool:
    """True when the bf16 projection + bf16 feature table cover this layer (gatmi_project_bf16_supported)."""
    shape = _lib.Shape(0, in_features, heads, channels)
    return bool(_lib.lib().gatmi_project_bf16_supported(C.byref(shape)))


def project_fwd_bf16(x: Tensor, weight: Tensor, att_dst: Tensor, heads: int, channels: int,
                     alloc_rows: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """bf16 MFMA projection: the bfloat16 feature table [N, H*Cp] and a_dst [N, H] (f32)."""
    lib, dev = _lib.lib(), x.device
    n, f = x.shape
    _project_operands(x, weight, att_dst, heads, channels)
    shape = _lib.Shape(n, f, heads, channels)
    with torch.cuda.device(dev):
        hp16 = _alloc_table(n, heads * padded_channels(channels), torch.bfloat16, dev, alloc_rows)
        a_dst = torch.empty(n, heads, dtype=torch.float32, device=dev)
        ws = _ws(lib.gatmi_project_bf16_workspace_bytes(C.byref(shape)), dev)
        _lib.check(lib.gatmi_project_fwd_bf16(C.byref(shape), x.data_ptr(), weight.data_ptr(),
                                              att_dst.data_ptr(), hp16.data_ptr(), a_dst.data_ptr(),
                                              ws.data_ptr(), ws.numel(), _stream(dev)),
                   "gatmi_project_fwd_bf16")
    return hp16, a_dst


def project_bwd_bf16(x: Tensor, weight: Tensor, d_hp: Tensor, heads: int, channels: int,
                     need_dx: bool, d_a_dst: Optional[Tensor] = None, att_dst: Optional[Tensor] = None
                     ) -> Tuple[Tensor, Optional[Tensor]]:
    """d_weight, d_x of the projection on the bf16 matrix cores (operands rounded, f32 sums)."""
    lib, dev = _lib.lib(), x.device
    n, f = x.shape
    _project_operands(x, weight, att_dst, heads, channels, d_hp, d_a_dst)
    shape = _lib.Shape(n, f, heads, channels)
    with torch.cuda.device(dev):
        d_w = torch.empty_like(weight)
        d_x = torch.empty_like(x) if need_dx else None
        ws = _ws(lib.gatmi_project_bf16_workspace_bytes(C.byref(shape)), dev)
        _lib.check(lib.gatmi_project_bwd_bf16(C.byref(shape), x.data_ptr(), weight.data_ptr(),
                                              d_hp.data_ptr(), _ptr(d_a_dst), _ptr(att_dst),
                                              d_w.data_ptr(), _ptr(d_x), ws.data_ptr(), ws.numel(),
                                              _stream(dev)), "gatmi_project_bwd_bf16")
    return d_w, d_x


def attn_fwd(layout: GraphLayout, cfg: AttnConfig, in_features: int, hp: Tensor, a_dst: Tensor,
             att_src: Tensor, att_dst: Tensor, bias: Optional[Tensor], need_lse: bool
             ) -> Tuple[Tensor, Optional[Tensor]]:
    """out [num_dst, H*C] (+ lse [num_dst, H] when a backward will follow)."""
    lib, dev = _lib.lib(), hp.device
    n = layout.num_dst
    with torch.cuda.device(dev):
        a = cfg.c_args(n, in_features, hp, a_dst, att_src, att_dst, bias, layout.num_nodes, layout.num_edges)
        out = torch.empty(n, cfg.heads * cfg.channels, dtype=torch.float32, device=dev)
        lse = torch.empty(n, cfg.heads, dtype=torch.float32, device=dev) if need_lse else None
        ws = _ws(lib.gatmi_attn_fwd_workspace_bytes(layout.ref, C.byref(a.shape)), dev)
        _lib.check(lib.gatmi_attn_fwd(layout.ref, C.byref(a), out.data_ptr(), _ptr(lse),
                                      ws.data_ptr(), ws.numel(), _stream(dev)), "gatmi_attn_fwd")
    return out, lse


def attn_bwd(layout: GraphLayout, cfg: AttnConfig, in_features: int, hp: Tensor, a_dst: Tensor,
             att_src: Tensor, att_dst: Tensor, bias: Optional[Tensor], out: Tensor, lse: Tensor,
             grad_out: Tensor, defer_dst_term: bool = False):
    """-> d_hp [num_src, Dp], d_att_src, d_att_dst, d_bias, d_a_dst (None unless deferred)."""
    lib, dev = _lib.lib(), hp.device
    with torch.cuda.device(dev):
        a = cfg.c_args(layout.num_dst, in_features, hp, a_dst, att_src, att_dst, bias, layout.num_nodes,
                       layout.num_edges)
        hc = cfg.heads * cfg.channels
        _expect(out, (layout.num_dst, hc), torch.float32, "out", dev)
        _expect(lse, (layout.num_dst, cfg.heads), torch.float32, "lse", dev)
        _expect(grad_out, (layout.num_dst, hc), torch.float32, "grad_out", dev)
        d_hp = torch.empty(layout.num_nodes, hp.size(1), dtype=torch.float32, device=dev)
        d_as = torch.empty_like(att_src)
        d_ad = torch.empty_like(att_dst)
        d_b = torch.empty_like(bias) if bias is not None else None
        d_a_dst = (torch.empty(layout.num_dst, cfg.heads, dtype=torch.float32, device=dev)
                   if defer_dst_term else None)
        query = lib.gatmi_attn_bwd_workspace_bytes_bf16 if cfg.feat_bf16 else lib.gatmi_attn_bwd_workspace_bytes
        ws = _ws(query(layout.ref, C.byref(a.shape)), dev)
        _lib.check(lib.gatmi_attn_bwd(layout.ref, C.byref(a), out.data_ptr(), lse.data_ptr(),
                                      grad_out.data_ptr(), d_hp.data_ptr(), d_as.data_ptr(),
                                      d_ad.data_ptr(), _ptr(d_b), _ptr(d_a_dst), ws.data_ptr(),
                                      ws.numel(), _stream(dev)), "gatmi_attn_bwd")
    return d_hp, d_as, d_ad, d_b, d_a_dst


# ------------------------------------------------------------------ source chunks (pipelined exchange)
def attn_fwd_partial(layout: GraphLayout, cfg: AttnConfig, in_features: int, hp_chunk: Tensor,
                     a_dst: Tensor, att_src: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """One source chunk of the forward: hp_chunk [layout.num_nodes, Dp] is the chunk's slice of the
    feature table, `layout` the chunk's own rectangular layout.  -> the un-normalised triple
    (m [num_dst, H], s [num_dst, H], acc [num_dst, Dp]) of every destination row."""
    lib, dev = _lib.lib(), a_dst.device
    n = layout.num_dst
    if hp_chunk.size(0) != layout.num_nodes:
        raise ValueError(f"chunk table has {hp_chunk.size(0)} rows, its layout {layout.num_nodes} sources")
    with torch.cuda.device(dev):
        a = cfg.c_args(n, in_features, hp_chunk, a_dst, att_src, None, None, layout.num_nodes, layout.num_edges)
        dp = cfg.heads * padded_channels(cfg.channels)
        m = torch.empty(n, cfg.heads, dtype=torch.float32, device=dev)
        s = torch.empty(n, cfg.heads, dtype=torch.float32, device=dev)
        acc = torch.empty(n, dp, dtype=torch.float32, device=dev)
        part = _lib.AttnPartial(m.data_ptr(), s.data_ptr(), acc.data_ptr())
        ws = _ws(lib.gatmi_attn_fwd_workspace_bytes(layout.ref, C.byref(a.shape)), dev)
        _lib.check(lib.gatmi_attn_fwd_partial(layout.ref, C.byref(a), C.byref(part), ws.data_ptr(),
                                              ws.numel(), _stream(dev)), "gatmi_attn_fwd_partial")
    return m, s, acc


def attn_fwd_combine(parts, cfg: AttnConfig, bias: Optional[Tensor], need_lse: bool
                     ) -> Tuple[Tensor, Optional[Tensor]]:
    """The triples of all chunks (in this order) -> out [num_dst, H*C], lse [num_dst, H]."""
    lib = _lib.lib()
    if not 1 <= len(parts) <= _lib.MAX_PARTS:
        raise ValueError(f"{len(parts)} chunks: between 1 and {_lib.MAX_PARTS}")
    dev, n = parts[0][0].device, parts[0][0].size(0)
    dp = cfg.heads * padded_channels(cfg.channels)
    for m, s, acc in parts:
        _expect(m, (n, cfg.heads), torch.float32, "partial m", dev)
        _expect(s, (n, cfg.heads), torch.float32, "partial s", dev)
        _expect(acc, (n, dp), torch.float32, "partial acc", dev)
    if bias is not None:
        _expect(bias.view(-1), (cfg.heads * cfg.channels,), torch.float32, "bias", dev)
    with torch.cuda.device(dev):
        arr = (_lib.AttnPartial * len(parts))(*[
            _lib.AttnPartial(m.data_ptr(), s.data_ptr(), acc.data_ptr()) for m, s, acc in parts])
        shape = _lib.Shape(n, 1, cfg.heads, cfg.channels)
        out = torch.empty(n, cfg.heads * cfg.channels, dtype=torch.float32, device=dev)
        lse = torch.empty(n, cfg.heads, dtype=torch.float32, device=dev) if need_lse else None
        _lib.check(lib.gatmi_attn_fwd_combine(C.byref(shape), len(parts), arr, _ptr(bias),
                                              out.data_ptr(), _ptr(lse), _stream(dev)),
                   "gatmi_attn_fwd_combine")
    return out, lse


class ChunkedBackward:
    """gatmi_attn_bwd_begin / _chunk / _end: the backward of the fused attention over source chunks.
    begin() prepares the destination rows once; chunk(k, ...) returns the complete d_h of chunk k's
    source rows (ready for its reduce-scatter) and accumulates d_a_dst; end() the parameter sums."""

    def __init__(self, cfg: AttnConfig, in_features: int, a_dst: Tensor, att_src: Tensor,
                 att_dst: Tensor, bias: Optional[Tensor], out: Tensor, lse: Tensor, grad_out: Tensor,
                 max_chunks: int):
        self.lib, self.dev = _lib.lib(), a_dst.device
        self.cfg, self.f, self.k = cfg, in_features, int(max_chunks)
        self.a_dst, self.att_src, self.att_dst, self.bias = a_dst, att_src, att_dst, bias
        self.grad_out = grad_out
        self.n = a_dst.size(0)
        self.first = True
        hc = cfg.heads * cfg.channels
        _expect(a_dst, (self.n, cfg.heads), torch.float32, "a_dst", self.dev)
        _expect(out, (self.n, hc), torch.float32, "out", self.dev)
        _expect(lse, (self.n, cfg.heads), torch.float32, "lse", self.dev)
        _expect(grad_out, (self.n, hc), torch.float32, "grad_out", self.dev)
        for name, t in (("att_src", att_src), ("att_dst", att_dst), ("bias", bias)):
            if t is not None:
                _expect(t.view(-1), (hc,), torch.float32, name, self.dev)
        with torch.cuda.device(self.dev):
            shape = _lib.Shape(self.n, in_features, cfg.heads, cfg.channels)
            nbytes = self.lib.gatmi_attn_bwd_node_workspace_bytes(C.byref(shape), self.k, int(bool(cfg.feat_bf16)))
            if nbytes == 0:
                raise RuntimeError(f"chunked backward: {max_chunks} chunks (at most {_lib.MAX_PARTS}) "
                                   f"or layer {cfg.heads} x {cfg.channels} not covered")
            self.node_ws = _ws(nbytes, self.dev)
            self.d_a_dst = torch.empty(self.n, cfg.heads, dtype=torch.float32, device=self.dev)
            a = self._args(None)
            _lib.check(self.lib.gatmi_attn_bwd_begin(C.byref(a), self.k, out.data_ptr(), lse.data_ptr(),
                                                     grad_out.data_ptr(), self.node_ws.data_ptr(),
                                                     self.node_ws.numel(), _stream(self.dev)),
                       "gatmi_attn_bwd_begin")

    def _args(self, hp, cfg=None, rows=None, num_edges=None):
        cfg = cfg or self.cfg
        if hp is not None:
            _expect(hp, (rows, cfg.heads * padded_channels(cfg.channels)),
                    torch.bfloat16 if cfg.feat_bf16 else torch.float32, "feature table", self.dev)
        if cfg.mask_mode == _lib.MASK_BYTES and num_edges is not None:   # begin / end read no mask
            if cfg.mask is None:
                raise ValueError("mask_mode BYTES needs the piece's mask")
            _expect(cfg.mask, (num_edges, cfg.heads), torch.uint8, "dropout mask", self.dev)
        a = _lib.Attn()
        a.shape = _lib.Shape(self.n, self.f, cfg.heads, cfg.channels)
        a.negative_slope, a.dropout_p = cfg.negative_slope, cfg.dropout
        a.mask_mode, a.seed, a.mask = cfg.mask_mode, cfg.seed, _ptr(cfg.mask)
        a.feat_bf16 = int(bool(cfg.feat_bf16))
        if hp is not None and (hp.dtype == torch.bfloat16) != bool(cfg.feat_bf16):
            raise TypeError(f"feature table is {hp.dtype} but feat_bf16={cfg.feat_bf16}")
        a.hp, a.a_dst = _ptr(hp), _ptr(self.a_dst)
        a.att_src, a.att_dst, a.bias = _ptr(self.att_src), _ptr(self.att_dst), _ptr(self.bias)
        return a

    def chunk(self, k: int, layout: GraphLayout, cfg_k: AttnConfig, hp_chunk: Tensor,
              d_hp_out: Optional[Tensor] = None) -> Tensor:
        """-> d_hp of the chunk's source rows [layout.num_nodes, Dp] (f32), into d_hp_out if given."""
        if hp_chunk.size(0) != layout.num_nodes or layout.num_dst != self.n:
            raise ValueError("chunk table / layout do not match")
        with torch.cuda.device(self.dev):
            dp = cfg_k.heads * padded_channels(cfg_k.channels)
            d_hp = d_hp_out if d_hp_out is not None else torch.empty(
                layout.num_nodes, dp, dtype=torch.float32, device=self.dev)
            if (d_hp.shape != (layout.num_nodes, dp) or d_hp.dtype != torch.float32
                    or not d_hp.is_contiguous()):
                raise ValueError("d_hp_out must be a contiguous float32 [chunk rows, Dp] tensor")
            a = self._args(hp_chunk, cfg_k, layout.num_nodes, layout.num_edges)
            ws = _ws(self.lib.gatmi_attn_bwd_chunk_workspace_bytes(layout.ref, C.byref(a.shape)), self.dev)
            _lib.check(self.lib.gatmi_attn_bwd_chunk(
                layout.ref, C.byref(a), int(k), self.k, int(not self.first), self.grad_out.data_ptr(),
                self.node_ws.data_ptr(), self.node_ws.numel(), d_hp.data_ptr(), self.d_a_dst.data_ptr(),
                ws.data_ptr(), ws.numel(), _stream(self.dev)), "gatmi_attn_bwd_chunk")
            self.first = False
        return d_hp

    def end(self, hp_own: Tensor):
        """hp_own: the table rows of the destination rows themselves.  -> d_att_src, d_att_dst, d_bias, d_a_dst"""
        with torch.cuda.device(self.dev):
            if self.first:                        # no chunk ran (no edges at all)
                self.d_a_dst.zero_()
            a = self._args(hp_own, None, self.n)
            d_as = torch.empty_like(self.att_src)
            d_ad = torch.empty_like(self.att_dst)
            d_b = torch.empty_like(self.bias) if self.bias is not None else None
            _lib.check(self.lib.gatmi_attn_bwd_end(C.byref(a), self.k, self.d_a_dst.data_ptr(),
                                                   self.node_ws.data_ptr(), self.node_ws.numel(),
                                                   d_as.data_ptr(), d_ad.data_ptr(), _ptr(d_b),
                                                   _stream(self.dev)), "gatmi_attn_bwd_end")
        return d_as, d_ad, d_b, self.d_a_dst


def attn_bf16_supported(heads: int, channels: int) -> bool:
    """True when the fused attention kernels take a bfloat16 feature table for this layer."""
    shape = _lib.Shape(0, 1, heads, channels)
    return bool(_lib.lib().gatmi_attn_bf16_supported(C.byref(shape)))


def add_dst_term(d_hp: Tensor, d_a_dst: Tensor, att_dst: Tensor, heads: int, channels: int) -> None:
    """d_hp[i,h,:] += d_a_dst[i,h] * att_dst[h,:] in place (rows of one shard)."""
    lib, dev = _lib.lib(), d_hp.device
    shape = _lib.Shape(d_hp.size(0), 1, heads, channels)
    _expect(d_hp, (d_hp.size(0), heads * padded_channels(channels)), torch.float32, "d_hp", dev)
    _expect(d_a_dst, (d_hp.size(0), heads), torch.float32, "d_a_dst", dev)
    _expect(att_dst.view(-1), (heads * channels,), torch.float32, "att_dst", dev)
    with torch.cuda.device(dev):
        _lib.check(lib.gatmi_add_dst_term(C.byref(shape), d_hp.data_ptr(), d_a_dst.data_ptr(),
                                          att_dst.data_ptr(), _stream(dev)), "gatmi_add_dst_term")


# ------------------------------------------------------------------ input-side aggregation
def inagg_supported(in_features: int, heads: int, channels: int, edges_per_node: float = None
                    ) -> bool:
    """True when the input-side aggregation kernels cover this layer AND it pays.  Bytes per
    destination node of one forward + backward (fp32): the edge kernels gather a row per edge
    in each direction; the input-side path also writes/reads the per-head aggregates Z and q
    [H*Fp] six times, the projected-row path its [Dp] tables about ten times.  Without a degree
    (edges_per_node None) the test is the row width alone."""
    dp = heads * padded_channels(channels)
    shape = _lib.Shape(0, in_features, heads, channels)
    if in_features >= dp or not bool(_lib.lib().gatmi_inagg_supported(C.byref(shape))):
        return False
    if edges_per_node is None:
        return True
    # float4 slots per lane of the one-edge kernels (inagg.hip make_in_geo): beyond 2 the
    # per-edge arithmetic, not the bytes, decides, and the projected-row kernels (1 slot) win.
    # The multi-edge kernels (one float4 per lane: H <= 4 or 5..8 heads, rows of <= 32 float4)
    # have no such limit, but with 8 heads in a lane they run about 1.5x the byte model
    # (tools/edge8_ab.py, 200k nodes, degree 25: 128 -> 8x32 2.80 ms against 2.51 ms projected,
    # 100 -> 8x30 2.51 / 2.57, 100 -> 6x64 2.63 / 3.81; degree 7: 100 -> 8x30 1.91 / 1.50)
    h2 = 1 << (heads - 1).bit_length()
    lph = padded_channels(in_features) // 4
    lh = min(64 // h2, 1 << (lph - 1).bit_length())
    multi_edge = h2 <= 8 and lph <= 32 and h2 * lh == 64
    if -(-lph // lh) > 2 and not multi_edge:
        return False
    fp = padded_channels(in_features)
    fs = fp + padded_channels(heads)
    cost_in = (2 * edges_per_node * fs + 6 * heads * fp) * (1.5 if h2 == 8 else 1.0)
    cost_rows = 2 * edges_per_node * dp + 10 * dp
    return cost_in < cost_rows


def inagg_pack(x: Tensor, weight: Tensor, att_src: Tensor, att_dst: Tensor, heads: int,
               channels: int, rows16: bool = False):
    """rows of x -> packed source rows xs [n, Fs] (x | a_src | pad) and a_dst [n, H].  rows16: also the
    bf16 table of those rows (gatmi_inagg_pack16: what inagg_table16(xs) would make, written by the
    same kernel) -> (xs, a_dst, xs16)."""
    lib, dev = _lib.lib(), x.device
    n, f = x.shape
    shape = _lib.Shape(n, f, heads, channels)
    fs = int(lib.gatmi_inagg_xs_floats(C.byref(shape)))
    if fs == 0:
        raise RuntimeError(f"input-side aggregation does not cover F={f} H={heads}")
    _project_operands(x, weight, att_dst, heads, channels)
    _expect(att_src.view(-1), (heads * channels,), torch.float32, "att_src", dev)
    with torch.cuda.device(dev):
        xs = torch.empty(n, fs, dtype=torch.float32, device=dev)
        a_dst = torch.empty(n, heads, dtype=torch.float32, device=dev)
        ws = _ws(lib.gatmi_inagg_pack_workspace_bytes(C.byref(shape)), dev)
        if rows16:
            row = int(lib.gatmi_inagg_xs16_bytes(C.byref(shape)))
            if row == 0:
                raise RuntimeError(f"bf16 source rows do not cover F={f} H={heads}")
            xs16 = torch.empty(n, row // 2, dtype=torch.bfloat16, device=dev)
            _lib.check(lib.gatmi_inagg_pack16(C.byref(shape), x.data_ptr(), weight.data_ptr(),
                                              att_src.data_ptr(), att_dst.data_ptr(), xs.data_ptr(),
                                              a_dst.data_ptr(), xs16.data_ptr(), ws.data_ptr(), ws.numel(),
                                              _stream(dev)),
                       "gatmi_inagg_pack16")
            return xs, a_dst, xs16
        _lib.check(lib.gatmi_inagg_pack(C.byref(shape), x.data_ptr(), weight.data_ptr(),
                                        att_src.data_ptr(), att_dst.data_ptr(), xs.data_ptr(),
                                        a_dst.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)),
                   "gatmi_inagg_pack")
    return xs, a_dst


def inagg_rows16_supported(in_features: int, heads: int, channels: int) -> bool:
    """bf16 source rows for the input-side gathers: F <= 112 and H <= 4, or 5..8 heads with F > 32
    (inagg.hip rows16_cover)."""
    shape = _lib.Shape(0, in_features, heads, channels)
    return int(_lib.lib().gatmi_inagg_xs16_bytes(C.byref(shape))) > 0


def inagg_table16(xs: Tensor, in_features: int, heads: int, channels: int) -> Tensor:
    """packed f32 source rows xs [n, Fs] -> their bf16 copy in 256-byte rows ([n, 128] bfloat16: 14
    pieces of 8 channels, then the four f32 logits, which are not rounded)."""
    lib, dev = _lib.lib(), xs.device
    shape = _lib.Shape(xs.size(0), in_features, heads, channels)
    row = int(lib.gatmi_inagg_xs16_bytes(C.byref(shape)))
    if row == 0:
        raise RuntimeError(f"bf16 source rows do not cover F={in_features} H={heads}")
    if int(lib.gatmi_inagg_xs_floats(C.byref(shape))) != xs.size(1):
        raise ValueError("xs row width does not match this layer")
    _expect(xs, tuple(xs.shape), torch.float32, "xs", dev)
    with torch.cuda.device(dev):
        xs16 = torch.empty(xs.size(0), row // 2, dtype=torch.bfloat16, device=dev)
        _lib.check(lib.gatmi_inagg_table16(C.byref(shape), xs.data_ptr(), xs16.data_ptr(), _stream(dev)),
                   "gatmi_inagg_table16")
    return xs16


def _inagg_args(cfg: AttnConfig, num_dst, in_features, xs, a_dst, weight, att_src, att_dst, bias,
                xs16=None):
    dev, hc = xs.device, cfg.heads * cfg.channels
    _expect(xs, tuple(xs.shape), torch.float32, "xs", dev)
    _expect(a_dst, (num_dst, cfg.heads), torch.float32, "a_dst", dev)
    _expect(weight, (hc, in_features), torch.float32, "weight", dev)
    for name, t in (("att_src", att_src), ("att_dst", att_dst), ("bias", bias)):
        if t is not None:
            _expect(t.view(-1), (hc,), torch.float32, name, dev)
    a = _lib.InAgg()
    a.shape = _lib.Shape(num_dst, in_features, cfg.heads, cfg.channels)
    a.negative_slope, a.dropout_p = cfg.negative_slope, cfg.dropout
    a.mask_mode, a.seed, a.mask = cfg.mask_mode, cfg.seed, _ptr(cfg.mask)
    a.stream = int(getattr(cfg, "stream", 0))
    a.xs, a.a_dst, a.weight = xs.data_ptr(), a_dst.data_ptr(), weight.data_ptr()
    a.att_src, a.att_dst, a.bias = _ptr(att_src), _ptr(att_dst), _ptr(bias)
    if xs16 is not None and (xs16.dtype != torch.bfloat16 or xs16.size(0) != xs.size(0) or not xs16.is_contiguous()):
        raise ValueError("xs16 must be the contiguous bfloat16 table gatmi_inagg_table16 made of xs")
    a.xs16 = _ptr(xs16)
    return a


def inagg_fwd(layout: GraphLayout, cfg: AttnConfig, xs: Tensor, a_dst: Tensor, weight: Tensor,
              att_src: Tensor, att_dst: Tensor, bias: Optional[Tensor], need_lse: bool,
              xs16: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
    """xs [num_src, Fs] (all nodes), a_dst [num_dst, H] -> out [num_dst, H*C], z [num_dst, H*Fp],
    lse [num_dst, H]."""
    lib, dev = _lib.lib(), xs.device
    n = layout.num_dst
    if xs.size(0) != layout.num_nodes or a_dst.size(0) != n:
        raise ValueError(f"xs has {xs.size(0)} rows / a_dst {a_dst.size(0)}, the graph "
                         f"{layout.num_nodes} sources / {n} destinations")
    with torch.cuda.device(dev):
        a = _inagg_args(cfg, n, weight.size(1), xs, a_dst, weight, att_src, att_dst, bias, xs16)
        if int(lib.gatmi_inagg_xs_floats(C.byref(a.shape))) != xs.size(1):
            raise ValueError("xs row width does not match this layer")
        if cfg.mask_mode == _lib.MASK_BYTES:
            _expect(cfg.mask, (layout.num_edges, cfg.heads), torch.uint8, "dropout mask", dev)
        zf = int(lib.gatmi_inagg_z_floats(C.byref(a.shape)))
        out = torch.empty(n, cfg.heads * cfg.channels, dtype=torch.float32, device=dev)
        z = torch.empty(n, zf, dtype=torch.float32, device=dev)
        lse = torch.empty(n, cfg.heads, dtype=torch.float32, device=dev) if need_lse else None
        ws = _ws(lib.gatmi_inagg_fwd_workspace_bytes(layout.ref, C.byref(a.shape)), dev)
        _lib.check(lib.gatmi_inagg_fwd(layout.ref, C.byref(a), out.data_ptr(), z.data_ptr(),
                                       _ptr(lse), ws.data_ptr(), ws.numel(), _stream(dev)),
                   "gatmi_inagg_fwd")
    return out, z, lse


def inagg_bwd(layout: GraphLayout, cfg: AttnConfig, xs: Tensor, a_dst: Tensor, weight: Tensor,
              att_src: Tensor, att_dst: Tensor, bias: Optional[Tensor], z: Tensor, lse: Tensor,
              grad_out: Tensor, xs16: Optional[Tensor] = None):
    """-> d_weight, d_att_src, d_att_dst, d_bias (partial sums on a destination partition)."""
    lib, dev = _lib.lib(), xs.device
    if xs.size(0) != layout.num_nodes:
        raise ValueError(f"xs has {xs.size(0)} rows, the graph {layout.num_nodes} sources")
    with torch.cuda.device(dev):
        a = _inagg_args(cfg, layout.num_dst, weight.size(1), xs, a_dst, weight, att_src, att_dst,
                        bias, xs16)
        if int(lib.gatmi_inagg_xs_floats(C.byref(a.shape))) != xs.size(1):
            raise ValueError("xs row width does not match this layer")
        zf = int(lib.gatmi_inagg_z_floats(C.byref(a.shape)))
        _expect(z, (layout.num_dst, zf), torch.float32, "z", dev)
        _expect(lse, (layout.num_dst, cfg.heads), torch.float32, "lse", dev)
        _expect(grad_out, (layout.num_dst, cfg.heads * cfg.channels), torch.float32, "grad_out", dev)
        if cfg.mask_mode == _lib.MASK_BYTES:
            _expect(cfg.mask, (layout.num_edges, cfg.heads), torch.uint8, "dropout mask", dev)
        d_w = torch.empty_like(weight)
        d_as = torch.empty_like(att_src)
        d_ad = torch.empty_like(att_dst)
        d_b = torch.empty_like(bias) if bias is not None else None
        ws = _ws(lib.gatmi_inagg_bwd_workspace_bytes(layout.ref, C.byref(a.shape)), dev)
        _lib.check(lib.gatmi_inagg_bwd(layout.ref, C.byref(a), z.data_ptr(), lse.data_ptr(),
                                       grad_out.data_ptr(), d_w.data_ptr(), d_as.data_ptr(),
                                       d_ad.data_ptr(), _ptr(d_b), ws.data_ptr(), ws.numel(),
                                       _stream(dev)), "gatmi_inagg_bwd")
    return d_w, d_as, d_ad, d_b
