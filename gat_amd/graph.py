"""Graph layout: edge_index -> destination-sorted CSR with self-loops (+ transpose).

The reference redoes remove_self_loops/add_self_loops inside every GATConv.forward
(three layers, every epoch: /root/reference/gat/model.py:40,43,46).  Here the layout
is built once per edge_index on the device by libgat_mi355.so and cached; the three
layers and all epochs share it.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Dict, Tuple

import torch

from . import _lib

DEFAULT_SEG_LEN = 512


class GraphLayout:
    """Device arrays of one graph plus the C descriptor the kernels take."""

    def __init__(self, desc: _lib.Graph, tensors: Dict[str, torch.Tensor], device):
        self.desc = desc
        self._t = tensors  # keeps the device memory behind `desc` alive
        self.device = device
        self.num_nodes = int(desc.num_src)
        self.num_dst = int(desc.num_dst)
        self.num_edges = int(desc.num_edges)
        self.seg_len = int(desc.seg_len)

    def __getattr__(self, name):
        t = self.__dict__.get("_t", {})
        if name in t:
            return t[name]
        raise AttributeError(name)

    @property
    def ref(self):
        return C.byref(self.desc)

    def edge_mask_to_csr(self, mask: torch.Tensor) -> torch.Tensor:
        """[E', H] keep-mask in normalised-edge-list order -> uint8 in dst-CSR order."""
        return mask.to(self.device)[self.perm.long()].to(torch.uint8).contiguous()


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def build_graph(edge_index: torch.Tensor, num_nodes: int, add_self_loops: bool = True,
                seg_len: int = DEFAULT_SEG_LEN, num_dst: int = None,
                dst_offset: int = 0) -> GraphLayout:
    """num_dst/dst_offset describe a destination partition: edge_index[1] then holds LOCAL
    destination ids in [0, num_dst), edge_index[0] global source ids in [0, num_nodes)."""
    num_dst = num_nodes if num_dst is None else num_dst
    if edge_index.dim() != 2 or edge_index.size(0) != 2:
        raise ValueError(f"edge_index must be [2, E], got {tuple(edge_index.shape)}")
    if edge_index.dtype != torch.int64:
        raise TypeError(f"edge_index must be int64 (torch.long), got {edge_index.dtype}")
    if not edge_index.is_cuda:
        raise RuntimeError("gat_amd runs on the GPU only: edge_index must be a CUDA/HIP tensor")
    lib = _lib.lib()
    dev = edge_index.device
    ei = edge_index.contiguous()
    e = ei.size(1)
    sizes = _lib.GraphSizes()
    _lib.check(lib.gatmi_graph_sizes_query(e, num_nodes, num_dst, int(add_self_loops), seg_len,
                                           C.byref(sizes)), "gatmi_graph_sizes_query")
    with torch.cuda.device(dev):
        i32 = lambda n: torch.empty(max(int(n), 1), dtype=torch.int32, device=dev)
        t = dict(rowptr=i32(num_dst + 1), col=i32(sizes.max_edges), perm=i32(sizes.max_edges),
                 rowptr_t=i32(num_nodes + 1), col_t=i32(sizes.max_edges),
                 pos_t=i32(sizes.max_edges), d2s=i32(sizes.max_edges))
        for p in ("plan", "plan_t"):
            t[p + ".item_row"] = i32(sizes.max_items)
            t[p + ".item_beg"] = i32(sizes.max_items)
            t[p + ".item_slot"] = i32(sizes.max_items)
            t[p + ".split_row"] = i32(sizes.max_split_rows)
            t[p + ".split_slot"] = i32(sizes.max_split_rows + 1)
        ws = torch.empty(int(sizes.workspace_bytes), dtype=torch.uint8, device=dev)
        g = _lib.Graph()
        for k in ("rowptr", "col", "perm", "rowptr_t", "col_t", "pos_t", "d2s"):
            setattr(g, k, t[k].data_ptr())
        for p, plan in (("plan", g.plan), ("plan_t", g.plan_t)):
            for k in ("item_row", "item_beg", "item_slot", "split_row", "split_slot"):
                setattr(plan, k, t[f"{p}.{k}"].data_ptr())
        _lib.check(lib.gatmi_graph_build(ei.data_ptr(), e, num_nodes, num_dst, int(add_self_loops),
                                         seg_len, C.byref(g), ws.data_ptr(), ws.numel(),
                                         _stream_ptr(dev)), "gatmi_graph_build")
        g.dst_offset = dst_offset
    ep = int(g.num_edges)
    for k in ("col", "perm", "col_t", "pos_t", "d2s"):
        t[k] = t[k][:ep]
    return GraphLayout(g, t, dev)


# edge_index identity -> layout.  Keyed on storage pointer, shape and version counter so
# an in-place edit of edge_index invalidates the entry; weak on the tensor so a freed
# edge_index (whose address may be reused) drops out.
_CACHE: Dict[Tuple, Tuple[weakref.ref, GraphLayout]] = {}


def cached_graph(edge_index: torch.Tensor, num_nodes: int, add_self_loops: bool = True,
                 seg_len: int = DEFAULT_SEG_LEN) -> GraphLayout:
    key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index.stride(),
           edge_index._version, num_nodes,
           bool(add_self_loops), seg_len, str(edge_index.device))
    hit = _CACHE.get(key)
    if hit is not None and hit[0]() is not None:  # owner alive => same storage, same contents
        return hit[1]
    layout = build_graph(edge_index, num_nodes, add_self_loops, seg_len)
    for k in [k for k, (r, _) in _CACHE.items() if r() is None]:
        del _CACHE[k]
    _CACHE[key] = (weakref.ref(edge_index), layout)
    return layout


def clear_graph_cache() -> None:
    _CACHE.clear()
