"""Destination-partitioned GAT layer over several GPUs (BASELINE.json
config 5): one process per GPU, torch.distributed over RCCL ("nccl" backend) on xGMI.

The reference is single-process (no torch.distributed anywhere in /root/reference); what
is partitioned here is its one layer call, /root/reference/gat/model.py:40,43,46, on a
graph too large or too slow for one GPU.  Softmax and aggregation are independent per
destination node, so rank r owns a contiguous block of destinations, their rows of x and
the in-edges of those destinations:

  forward   project own rows -> ALL-GATHER h (the one exchange step) -> fused attention
            over the own destination rows (sources are global ids into the gathered h)
  backward  fused backward over the own edges -> partial d_h for ALL sources ->
            REDUCE-SCATTER d_h (adjoint of the all-gather) -> + d_a_dst att_dst on own rows
            -> own d_x and partial d_W -> ALL-REDUCE of d_W, d_att_src, d_att_dst, d_bias
  (exchange="x": all-gather the input rows instead and project them on every rank; see
  _PartitionedFn)

When the layer's input needs no gradient and is narrower than its output (the reference's
conv1: x is `data.x`), the input-side aggregation kernels run instead (gat_amd/csrc/inagg.hip,
_PartitionedInAggFn): every edge gathers a packed source row  x_j | a_src[j]  and the backward
needs no exchange at all besides the parameter all-reduce.
  exchange="xs"  pack own rows -> ALL-GATHER the packed rows (Fs floats per node) every step
  exchange="as"  the x part of the packed table is static data: it is all-gathered once per
                 x tensor and kept; every step all-gathers only the projected source logits
                 a_src [N, H] (they change with the weights) into the table's last columns

Ranks own contiguous destination ranges holding about the same number of in-edges
(`balanced_bounds`; `balance="nodes"`: equal blocks of ceil(N / P) nodes).  Shards are padded to
the longest range so the collectives stay the plain equal-shard forms; gathered tables are
indexed by padded ids (Partition.to_padded).  The compute steps go through a small backend
object: `HipBackend` is the product (C ABI, GPU only); tests swap in a CPU stand-in to
check the partition + collective algebra under gloo.
"""
from __future__ import annotations

import json
import os
import sys
import time
import weakref
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor, nn

from . import _lib


class TableMap:
    """Where every node's row sits in a gathered table that arrives in `chunks` pieces.

    Owner o holds the contiguous node range [bounds[o], bounds[o+1]); its rows are cut into
    `chunks` consecutive sub-ranges of sub[o] rows, and piece k of the table is the concatenation
    over the owners of their k-th sub-range -- so that piece k is ONE collective (an all-gather of
    the k-th sub-range of every rank) that can be in flight while piece k-1 is being consumed.
      padded (uneven False): every owner's sub-range is padded to the same sub = ceil(block / chunks)
        rows, block = the longest range: plain equal-shard collectives, world * sub rows per piece;
      exact (uneven True): sub[o] = ceil(len_o / chunks), no padding rows at all; the collectives are
        per-owner broadcasts / reductions (dist.gather_piece / reduce_piece).
    chunks == 1, padded, is the rank-major layout of the one-piece exchange modes."""

    def __init__(self, bounds: Tuple[int, ...], chunks: int = 1, uneven: bool = False):
        self.bounds = tuple(int(b) for b in bounds)
        self.world = len(self.bounds) - 1
        self.chunks = int(chunks)
        self.uneven = bool(uneven)
        if self.chunks < 1:
            raise ValueError(f"chunks must be >= 1, got {chunks}")
        self.lens = [self.bounds[o + 1] - self.bounds[o] for o in range(self.world)]
        block = max(max(self.lens), 1)
        k = self.chunks
        if self.uneven:
            self.sub = [(l + k - 1) // k for l in self.lens]
        else:
            self.sub = [(block + k - 1) // k] * self.world
        # cnt[s][o]: real rows of owner o in piece s; cap[s][o]: rows it occupies there
        self.cnt = [[min(max(self.lens[o] - s * self.sub[o], 0), self.sub[o]) for o in range(self.world)]
                    for s in range(k)]
        self.cap = self.cnt if self.uneven else [list(self.sub) for _ in range(k)]
        self.pre = [[sum(self.cap[s][:o]) for o in range(self.world)] for s in range(k)]
        self.rows = [sum(self.cap[s]) for s in range(k)]
        self.base = [sum(self.rows[:s]) for s in range(k)]
        self.total = sum(self.rows)
        self._lut = {}

    def _tensors(self, device):
        t = self._lut.get(device)
        if t is None:
            i64 = lambda v: torch.tensor(v, dtype=torch.int64, device=device)
            start = [[self.base[s] + self.pre[s][o] for o in range(self.world)] for s in range(self.chunks)]
            t = (i64(self.bounds), i64([max(v, 1) for v in self.sub]), i64(start))
            self._lut[device] = t
        return t

    def locate(self, ids: Tensor) -> Tuple[Tensor, Tensor]:
        """global node ids -> (row in the table, piece that holds it)"""
        b, sub, start = self._tensors(ids.device)
        owner = torch.searchsorted(b, ids, right=True) - 1
        t = ids - b[owner]
        so = sub[owner]
        piece = torch.div(t, so, rounding_mode="floor")
        return start[piece, owner] + (t - piece * so), piece

    def to_table(self, ids: Tensor) -> Tensor:
        return self.locate(ids)[0]

    def own(self, rank: int, s: int) -> Tuple[int, int, int]:
        """(first table row, first local row, count) of `rank`'s rows in piece s"""
        return self.base[s] + self.pre[s][rank], s * self.sub[rank], self.cnt[s][rank]


@dataclass
class Partition:
    """Rank r owns the contiguous destination range [bounds[r], bounds[r+1]).  `bounds` None:
    equal blocks of ceil(N / P) nodes.  Given (balanced_bounds): ranges of unequal length that
    hold about the same number of in-edges -- what the per-rank work is proportional to
    (SURVEY.md 8e).  uneven False: every collective is the plain equal-shard form, shards are
    padded to `block` = the longest range, and the gathered tables are indexed by PADDED ids
    (rank * block + offset inside the rank's range; `to_padded`) -- for equal blocks those are
    the global ids themselves.  uneven True (make_partition picks it when padding would add more
    than 25 % rows): tables hold exactly N rows, indexed by global id, and the collectives move
    every rank's exact range (gather_rows / scatter_rows)."""
    num_nodes: int
    world: int
    rank: int
    bounds: Optional[Tuple[int, ...]] = None
    uneven: bool = False

    def __post_init__(self):
        if self.bounds is not None:
            b = tuple(int(v) for v in self.bounds)
            if (len(b) != self.world + 1 or b[0] != 0 or b[-1] != self.num_nodes
                    or any(b[i] > b[i + 1] for i in range(self.world))):
                raise ValueError(f"bounds {b} do not split {self.num_nodes} nodes over {self.world} ranks")
            self.bounds = b
        self._maps = {}

    @property
    def all_bounds(self) -> Tuple[int, ...]:
        """the ranges written out (equal blocks when `bounds` is None)"""
        if self.bounds is not None:
            return self.bounds
        blk = (self.num_nodes + self.world - 1) // self.world
        return tuple(min(r * blk, self.num_nodes) for r in range(self.world + 1))

    def table(self, chunks: int = 1) -> TableMap:
        m = self._maps.get(chunks)
        if m is None:
            m = self._maps[chunks] = TableMap(self.all_bounds, chunks, self.uneven)
        return m

    @property
    def block(self) -> int:
        """rows a rank's shard occupies in an equal-shard collective (the longest range)"""
        if self.bounds is None:
            return (self.num_nodes + self.world - 1) // self.world
        return max(max(self.bounds[i + 1] - self.bounds[i] for i in range(self.world)), 1)

    @property
    def lo(self) -> int:
        return self.all_bounds[self.rank]

    @property
    def hi(self) -> int:
        return self.all_bounds[self.rank + 1]

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    @property
    def padded_nodes(self) -> int:
        """rows of a gathered one-piece table"""
        return self.num_nodes if self.uneven else self.block * self.world

    @property
    def padding_ratio(self) -> float:
        """rows the collectives move per real row (1.0 = nothing wasted)"""
        return self.padded_nodes / max(self.num_nodes, 1)

    @property
    def plo(self) -> int:
        """first row of this rank's shard in a gathered one-piece table"""
        return self.lo if self.uneven else self.rank * self.block

    @property
    def phi(self) -> int:
        return self.plo + self.n_local

    def to_padded(self, ids: Tensor) -> Tensor:
        """global node ids -> rows of a gathered one-piece table"""
        if self.bounds is None or self.uneven:
            return ids
        b = torch.tensor(self.bounds, dtype=ids.dtype, device=ids.device)
        owner = torch.searchsorted(b, ids, right=True) - 1
        return owner * self.block + (ids - b[owner])


PAD_LIMIT = 1.25     # make_partition: beyond this many table rows per node the exact collectives run


def balanced_bounds(edge_index: Tensor, num_nodes: int, world: int) -> Tuple[int, ...]:
    """Contiguous destination ranges with about equal in-edge counts (after self-loop
    normalisation: every node counts its own loop): bounds[r] = the first node at which the
    prefix sum of in-degrees reaches r / P of the total.  Computed from the whole edge_index,
    so every rank derives the same bounds without talking to the others."""
    src, dst = edge_index[0], edge_index[1]
    deg = torch.bincount(dst[src != dst], minlength=num_nodes) + 1
    pref = torch.cumsum(deg, 0)
    total = int(pref[-1]) if num_nodes else 0
    targets = torch.tensor([total * r // world for r in range(1, world)], dtype=pref.dtype,
                           device=pref.device)
    cuts = torch.searchsorted(pref, targets, right=False) + 1 if world > 1 else targets
    cuts = [min(int(c), num_nodes) for c in cuts.tolist()]
    bounds = [0] + cuts + [num_nodes]
    for i in range(1, len(bounds)):                # monotone (hubs can make a target repeat)
        bounds[i] = max(bounds[i], bounds[i - 1])
    return tuple(bounds)


_PARTS: Dict[Tuple, Tuple["weakref.ref", Tuple[int, ...]]] = {}


def make_partition(edge_index: Tensor, num_nodes: int, world: int, rank: int,
                   balance: str = "edges", padding: str = "auto") -> Partition:
    """The partition every layer of a network (and its caller, who slices x and y) shares:
    balance = "edges" (balanced_bounds, cached per edge_index object) or "nodes" (equal blocks).
    padding = "pad": equal-shard collectives on shards padded to the longest range; "exact": no
    padding rows, per-owner collectives; "auto": padded unless that moves more than PAD_LIMIT
    rows per node (a degree-ordered graph: the range holding the low-degree tail is several
    times N / P long) -- every rank decides from the same bounds."""
    if padding not in ("auto", "pad", "exact"):
        raise ValueError(f"padding must be 'auto', 'pad' or 'exact', got {padding!r}")
    if balance == "nodes":
        return Partition(num_nodes, world, rank, None, padding == "exact")
    if balance != "edges":
        raise ValueError(f"balance must be 'edges' or 'nodes', got {balance!r}")
    key = (id(edge_index), edge_index._version, num_nodes, world)
    hit = _PARTS.get(key)
    if hit is None or hit[0]() is not edge_index:
        for k in [k for k, (r, _) in _PARTS.items() if r() is None]:
            del _PARTS[k]
        hit = (weakref.ref(edge_index), balanced_bounds(edge_index, num_nodes, world))
        _PARTS[key] = hit
    part = Partition(num_nodes, world, rank, hit[1])
    part.uneven = padding == "exact" or (padding == "auto" and part.block * world > PAD_LIMIT * max(num_nodes, 1))
    return part


def local_edges(edge_index: Tensor, part: Partition) -> Tensor:
    """This rank's share of the NORMALISED edge list: in-edges of its destinations in their
    original order with self-loops removed, then one loop per own node (exactly what
    remove_self_loops + add_self_loops leaves for those destinations).  Row 0 holds the
    sources as rows of the gathered table (padded ids; == global ids for equal blocks), row 1
    local destination ids."""
    src, dst = edge_index[0], edge_index[1]
    keep = (dst >= part.lo) & (dst < part.hi) & (src != dst)
    loops = torch.arange(part.n_local, dtype=edge_index.dtype, device=edge_index.device)
    return torch.stack([torch.cat([part.to_padded(src[keep]), loops + part.plo]),
                        torch.cat([dst[keep] - part.lo, loops])])


def chunk_edge_lists(edge_index: Tensor, part: Partition, tm: TableMap):
    """This rank's share of the normalised edge list (as local_edges) split by the piece of the
    table that holds each edge's source.  -> per piece (edges [2, E_s], sel [E_s]): row 0 = source
    rows counted from the piece's first row, row 1 = local destination ids, both in the order of
    the rank's list; sel = positions of these edges in that list (for injected dropout masks)."""
    src, dst = edge_index[0], edge_index[1]
    keep = (dst >= part.lo) & (dst < part.hi) & (src != dst)
    loops = torch.arange(part.n_local, dtype=edge_index.dtype, device=edge_index.device)
    rows, piece = tm.locate(torch.cat([src[keep], loops + part.lo]))
    ldst = torch.cat([dst[keep] - part.lo, loops])
    out = []
    for k in range(tm.chunks):
        sel = torch.nonzero(piece == k).view(-1)
        out.append((torch.stack([rows[sel] - tm.base[k], ldst[sel]]), sel))
    return out


# ------------------------------------------------------------------ collectives
def _host_staged(t: Tensor, group) -> bool:
    """gloo rehearsals of the GPU path (two ranks sharing one card): stage device tensors through
    the host ourselves.  gloo's own device path (side streams + pinned copies issued from its
    worker threads) stalled for seconds to minutes per collective when two processes shared a GPU."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_gather_rows(shard: Tensor, world: int, group=None) -> Tensor:
    """equal shards -> [world * rows, ...]"""
    if _host_staged(shard, group):
        return all_gather_rows(shard.cpu(), world, group).to(shard.device)
    out = shard.new_empty((world * shard.size(0),) + tuple(shard.shape[1:]))
    if dist.get_backend(group) == "gloo":         # gloo: the list form (no flat all-gather on CPU tensors)
        dist.all_gather(list(out.chunk(world, 0)), shard.contiguous(), group=group)
    else:                                         # RCCL: an error here is an error, not a cue to retry
        dist.all_gather_into_tensor(out, shard.contiguous(), group=group)
    return out


def reduce_scatter_rows(full: Tensor, world: int, rank: int, group=None) -> Tensor:
    if _host_staged(full, group):
        return reduce_scatter_rows(full.cpu(), world, rank, group).to(full.device)
    out = full.new_empty((full.size(0) // world,) + tuple(full.shape[1:]))
    if dist.get_backend(group) == "gloo":         # gloo has no reduce-scatter: reduce, then slice
        tmp = full.clone()
        dist.all_reduce(tmp, group=group)
        out.copy_(tmp.chunk(world, 0)[rank])
    else:
        dist.reduce_scatter_tensor(out, full.contiguous(), group=group)
    return out


def _global_rank(group, r: int) -> int:
    return r if group is None else dist.get_global_rank(group, r)


class _Pending:
    """Collectives in flight (async_op works) and what to do once they are done.  wait() makes the
    CURRENT stream wait for them (RCCL: no host block) and runs the epilogue once."""

    def __init__(self, works=(), after=None):
        self.works, self.after = [w for w in works if w is not None], after

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []
        if self.after is not None:
            after, self.after = self.after, None
            after()


COMM_MS: Dict[str, list] = {}      # bench: name -> [(start event, end event)] of timed collectives
_COMM_TIMING = False


def comm_timing(on: bool) -> None:
    """bench.py --gpus N: bracket every collective of the layer with events on the current
    stream (mode "h" pieces: the time the consumer actually WAITED, i.e. what overlap did not hide)."""
    global _COMM_TIMING
    _COMM_TIMING = bool(on)
    COMM_MS.clear()


class _Timed:
    def __init__(self, name: str, ref: Tensor):
        self.on = _COMM_TIMING and ref.is_cuda
        self.name = name
        if self.on:
            self.t0 = torch.cuda.Event(enable_timing=True)
            self.t1 = torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.on:
            self.t0.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.t1.record()
            COMM_MS.setdefault(self.name, []).append((self.t0, self.t1))


def comm_ms_read() -> Dict[str, float]:
    """-> {name: total milliseconds since comm_timing(True)} (synchronises)"""
    torch.cuda.synchronize()
    out = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in COMM_MS.items()}
    COMM_MS.clear()
    return out


def gather_piece(local: Tensor, tm: TableMap, s: int, rank: int, out: Tensor, group=None,
                 async_op: bool = False) -> _Pending:
    """Piece s of the table: every rank's s-th sub-range -> out [tm.rows[s], ...] (a view of the
    table).  `local` holds this rank's rows from row 0; padded maps need tm.chunks * tm.sub[rank]
    rows of it (zero rows behind the real ones).  Padded: one equal-shard all-gather; exact: one
    broadcast per owner of its exact rows."""
    _, l0, cnt = tm.own(rank, s)
    if tm.rows[s] == 0:
        return _Pending()
    if _host_staged(local, group):                 # gloo rehearsal on a shared card: through the host
        host = local.new_empty((tm.rows[s],) + tuple(local.shape[1:]), device="cpu")
        gather_piece(local.cpu(), tm, s, rank, host, group).wait()
        out.copy_(host)
        return _Pending()
    if not tm.uneven:
        sub = tm.sub[rank]
        src = local[l0:l0 + sub]
        if src.size(0) != sub:
            raise ValueError(f"padded piece {s} needs rows [{l0}, {l0 + sub}) of the local shard, "
                             f"which has {local.size(0)}")
        if dist.get_backend(group) == "gloo":
            w = dist.all_gather(list(out.chunk(tm.world, 0)), src.contiguous(), group=group, async_op=async_op)
        else:
            w = dist.all_gather_into_tensor(out, src.contiguous(), group=group, async_op=async_op)
        return _Pending([w])
    works = []
    for o in range(tm.world):
        c = tm.cnt[s][o]
        if c == 0:
            continue
        view = out[tm.pre[s][o]: tm.pre[s][o] + c]
        if o == rank:
            view.copy_(local[l0:l0 + c])
        works.append(dist.broadcast(view, src=_global_rank(group, o), group=group, async_op=async_op))
    return _Pending(works)


def reduce_piece(full: Tensor, tm: TableMap, s: int, rank: int, out: Tensor, group=None,
                 async_op: bool = False) -> _Pending:
    """The adjoint: piece s of a table of partial sums [tm.rows[s], ...] -> this rank's rows of
    it, summed over the ranks, into out [tm.sub[rank] (padded) or tm.cnt[s][rank] (exact), ...]."""
    if tm.rows[s] == 0:
        return _Pending()
    if _host_staged(full, group):
        host = out.new_empty(out.shape, device="cpu")
        reduce_piece(full.cpu(), tm, s, rank, host, group).wait()
        out.copy_(host)
        return _Pending()
    gloo = dist.get_backend(group) == "gloo"
    if not tm.uneven and not gloo:
        return _Pending([dist.reduce_scatter_tensor(out, full, group=group, async_op=async_op)])
    if not tm.uneven:                             # gloo has no reduce-scatter: reduce, then slice
        tmp = full.clone()
        w = dist.all_reduce(tmp, group=group, async_op=async_op)
        lo = tm.pre[s][rank]
        return _Pending([w], lambda: out.copy_(tmp[lo:lo + out.size(0)]))
    works, mine = [], None
    for o in range(tm.world):
        c = tm.cnt[s][o]
        if c == 0:
            continue
        view = full[tm.pre[s][o]: tm.pre[s][o] + c]
        works.append(dist.reduce(view, dst=_global_rank(group, o), group=group, async_op=async_op))
        if o == rank:
            mine = view
    return _Pending(works, (lambda: out.copy_(mine)) if mine is not None else None)


def gather_rows(local: Tensor, part: Partition, group=None) -> Tensor:
    """every rank's rows -> the one-piece table [part.padded_nodes, ...]"""
    if not part.uneven:
        return all_gather_rows(_pad_rows(local, part.block), part.world, group)
    out = local.new_empty((part.num_nodes,) + tuple(local.shape[1:]))
    gather_piece(local, part.table(1), 0, part.rank, out, group).wait()
    return out


def scatter_rows(full: Tensor, part: Partition, group=None) -> Tensor:
    """reduce-scatter of a one-piece table of partial sums -> this rank's rows [n_local, ...]"""
    if not part.uneven:
        return reduce_scatter_rows(full, part.world, part.rank, group)[:part.n_local]
    out = full.new_empty((part.n_local,) + tuple(full.shape[1:]))
    reduce_piece(full.contiguous(), part.table(1), 0, part.rank, out, group).wait()
    return out


def all_reduce_(t: Tensor, group=None, op=None) -> Tensor:
    """in-place all-reduce (sum unless `op`)"""
    kw = {} if op is None else {"op": op}
    if _host_staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, group=group, **kw)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group, **kw)
    return t


def all_reduce_many(tensors, group=None) -> None:
    """one all-reduce for several small gradients (latency-bound: pack them)."""
    live = [t for t in tensors if t is not None]
    flat = torch.cat([t.reshape(-1) for t in live])
    all_reduce_(flat, group)
    off = 0
    for t in live:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()


# ------------------------------------------------------------------ compute backend
class HipBackend:
    """The product path: libgat_mi355.so through gat_amd.ops."""

    def __init__(self, seg_len: int = 512):
        self.seg_len = seg_len

    def build_layout(self, local_ei: Tensor, part: Partition):
        from .graph import build_graph
        return build_graph(local_ei, part.padded_nodes, add_self_loops=False, seg_len=self.seg_len,
                           num_dst=part.n_local, dst_offset=part.plo)

    def project_fwd(self, x, w, att_d, cfg):
        from . import ops
        if cfg.feat_bf16:      # bf16 table: half the bytes on the wire of exchange "h" and in the gathers
            return ops.project_fwd_bf16(x, w, att_d, cfg.heads, cfg.channels)
        return ops.project_fwd(x, w, att_d, cfg.heads, cfg.channels)

    def bf16_supported(self, f, cfg):
        from . import ops
        return ops.project_bf16_supported(f, cfg.heads, cfg.channels)

    # source chunks (exchange "h" pipelined: PartitionedGATConv(chunks=K))
    def project_fwd_padded(self, x, w, att_d, cfg, rows):
        """project_fwd into a table of `rows` >= len(x) rows (zero rows behind the real ones), so the
        equal-shard pieces of a padded map are views of it"""
        from . import ops
        fwd = ops.project_fwd_bf16 if cfg.feat_bf16 else ops.project_fwd
        return fwd(x, w, att_d, cfg.heads, cfg.channels, alloc_rows=rows)

    def build_chunk_layout(self, edges: Tensor, num_src: int, num_dst: int):
        from .graph import build_graph
        return build_graph(edges, num_src, add_self_loops=False, seg_len=self.seg_len, num_dst=num_dst)

    def attn_fwd_partial(self, layout, cfg, f, hp_piece, a_dst, att_s):
        from . import ops
        return ops.attn_fwd_partial(layout, cfg, f, hp_piece, a_dst, att_s)

    def attn_fwd_combine(self, parts, cfg, bias, need_lse):
        from . import ops
        return ops.attn_fwd_combine(parts, cfg, bias, need_lse)

    def chunked_backward(self, cfg, f, a_dst, att_s, att_d, bias, out, lse, grad_out, max_chunks):
        from . import ops
        return ops.ChunkedBackward(cfg, f, a_dst, att_s, att_d, bias, out, lse, grad_out, max_chunks)

    def attn_fwd(self, layout, cfg, f, hp_full, a_dst, att_s, att_d, bias, need_lse):
        from . import ops
        return ops.attn_fwd(layout, cfg, f, hp_full, a_dst, att_s, att_d, bias, need_lse)

    def attn_bwd(self, layout, cfg, f, hp_full, a_dst, att_s, att_d, bias, out, lse, grad_out):
        from . import ops
        return ops.attn_bwd(layout, cfg, f, hp_full, a_dst, att_s, att_d, bias, out, lse, grad_out,
                            defer_dst_term=True)

    def add_dst_term(self, d_hp, d_a_dst, att_d, cfg):
        from . import ops
        ops.add_dst_term(d_hp, d_a_dst, att_d, cfg.heads, cfg.channels)

    def project_bwd(self, x, w, d_hp, cfg, need_dx):
        from . import ops
        bwd = ops.project_bwd_bf16 if cfg.feat_bf16 else ops.project_bwd
        return bwd(x, w, d_hp.contiguous(), cfg.heads, cfg.channels, need_dx)

    # input-side aggregation (x without gradient, F < H*Cp)
    def inagg_supported(self, f, cfg):
        from . import ops
        return ops.inagg_supported(f, cfg.heads, cfg.channels)

    def inagg_logit_columns(self, f, cfg):
        """(first, count): where a_src sits in a packed row."""
        fp = (f + 3) // 4 * 4
        return fp, (cfg.heads + 3) // 4 * 4

    def inagg_pack(self, x, w, att_s, att_d, cfg):
        from . import ops
        return ops.inagg_pack(x, w, att_s, att_d, cfg.heads, cfg.channels)

    def inagg_fwd(self, layout, cfg, xs_full, a_dst, w, att_s, att_d, bias, need_lse, xs16=None):
        from . import ops
        return ops.inagg_fwd(layout, cfg, xs_full, a_dst, w, att_s, att_d, bias, need_lse, xs16)

    def inagg_bwd(self, layout, cfg, xs_full, a_dst, w, att_s, att_d, bias, z, lse, grad_out, xs16=None):
        from . import ops
        return ops.inagg_bwd(layout, cfg, xs_full, a_dst, w, att_s, att_d, bias, z, lse, grad_out, xs16)

    # bf16 copies of the packed rows for the gathers (feature_dtype=torch.bfloat16 on exchange xs / as)
    def inagg_rows16_supported(self, f, cfg):
        from . import ops
        return ops.inagg_rows16_supported(f, cfg.heads, cfg.channels)

    def inagg_table16(self, table, f, cfg):
        from . import ops
        return ops.inagg_table16(table, f, cfg.heads, cfg.channels)


def _pad_rows(t: Tensor, rows: int) -> Tensor:
    if t.size(0) == rows:
        return t.contiguous()
    out = t.new_zeros((rows,) + tuple(t.shape[1:]))
    out[:t.size(0)] = t
    return out


class _PartitionedFn(torch.autograd.Function):
    """exchange == "h": all-gather the projected features (Dp columns), reduce-scatter d_h.
    exchange == "x": all-gather the INPUT rows (F columns) and project all of them on every
    rank; the backward then needs no reduce-scatter of d_h -- d_W = sum_r d_h_r^T x is
    finished by the parameter all-reduce, and only a layer whose input needs a gradient
    reduce-scatters d_x (F columns).  Cheaper on the wire whenever F < Dp."""

    @staticmethod
    def forward(ctx, x_local, weight, att_src, att_dst, bias, part, layout, cfg, be, group,
                exchange):
        x_local, weight = x_local.contiguous(), weight.contiguous()
        att_s, att_d = att_src.contiguous().view(-1), att_dst.contiguous().view(-1)
        bias_c = None if bias is None else bias.contiguous()
        f = x_local.size(1)
        if exchange == "x":
            with _Timed("allgather_x", x_local):
                x_full = gather_rows(x_local, part, group)
            hp_full, a_dst_full = be.project_fwd(x_full, weight, att_d, cfg)
            a_dst = a_dst_full[part.plo:part.phi].contiguous()
            x_saved = x_full
        else:
            hp_local, a_dst = be.project_fwd(x_local, weight, att_d, cfg)
            with _Timed("allgather_h", hp_local):
                hp_full = gather_rows(hp_local, part, group)
            x_saved = x_local
        need_grad = any(ctx.needs_input_grad[:5])
        out, lse = be.attn_fwd(layout, cfg, f, hp_full, a_dst, att_s, att_d, bias_c, need_grad)
        if need_grad:
            ctx.save_for_backward(x_saved, weight, att_s, att_d, bias_c, hp_full, a_dst, out, lse)
            ctx.misc = (part, layout, cfg, be, group, att_src.shape, exchange)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x_saved, weight, att_s, att_d, bias_c, hp_full, a_dst, out, lse = ctx.saved_tensors
        part, layout, cfg, be, group, att_shape, exchange = ctx.misc
        need_dx = ctx.needs_input_grad[0]
        d_hp_part, d_as, d_ad, d_b, d_a_dst = be.attn_bwd(
            layout, cfg, x_saved.size(1), hp_full, a_dst, att_s, att_d, bias_c, out, lse,
            grad_out.contiguous())
        if exchange == "x":
            own = d_hp_part[part.plo:part.phi]        # a view: the term lands in d_hp_part
            be.add_dst_term(own, d_a_dst, att_d, cfg)
            d_w, d_x_part = be.project_bwd(x_saved, weight, d_hp_part, cfg, need_dx)
            d_x = None
            if need_dx:
                with _Timed("reduce_scatter_dx", d_x_part):
                    d_x = scatter_rows(d_x_part, part, group)
        else:
            with _Timed("reduce_scatter_dh", d_hp_part):
                d_hp = scatter_rows(d_hp_part, part, group)
            d_hp = d_hp.contiguous()
            be.add_dst_term(d_hp, d_a_dst, att_d, cfg)
            d_w, d_x = be.project_bwd(x_saved, weight, d_hp, cfg, need_dx)
        with _Timed("allreduce_params", d_w):
            all_reduce_many([d_w, d_as, d_ad, d_b], group)
        return (d_x, d_w, d_as.view(att_shape), d_ad.view(att_shape), d_b,
                None, None, None, None, None, None)


_SEED_STEP = 0x9E3779B97F4A7C15      # per-piece dropout streams: seed + k * this (mod 2^63)


def _piece_cfg(cfg, k: int):
    """The attention config of piece k: a hash-dropout stream of its own (the pieces' layouts count
    their edges from 0 each, so one seed would repeat the keep pattern piece after piece), or the
    piece's slice of an injected byte mask (tests: cfg.piece_masks)."""
    import copy
    c = copy.copy(cfg)
    if cfg.mask_mode in (_lib.MASK_HASH,):
        c.seed = (int(cfg.seed) + k * _SEED_STEP) % (1 << 63)
    masks = getattr(cfg, "piece_masks", None)
    if masks is not None:
        c.mask = masks[k]
    return c


class _PipelinedFn(torch.autograd.Function):
    """exchange == "h" in `tm.chunks` pieces, compute overlapped with the collectives (SURVEY.md 8e
    "Overlap"): the all-gather of the projected rows is issued as one asynchronous collective per
    piece (every rank's k-th sub-range), the fused attention runs over the sources of piece k --
    a rectangular layout of its own -- as soon as it has arrived and while piece k+1 is in flight,
    and the per-piece (m, s, acc) triples are joined at the end.  Backward: the edge pass over
    piece k's transposed CSR yields the complete d_h of that piece's rows; its reduce-scatter
    starts while piece k+1 computes."""

    @staticmethod
    def forward(ctx, x_local, weight, att_src, att_dst, bias, part, tm, layouts, cfg, be, group):
        x_local, weight = x_local.contiguous(), weight.contiguous()
        att_s, att_d = att_src.contiguous().view(-1), att_dst.contiguous().view(-1)
        bias_c = None if bias is None else bias.contiguous()
        f, k, rank, n_loc = x_local.size(1), tm.chunks, part.rank, part.n_local
        rows_alloc = n_loc if tm.uneven else k * tm.sub[rank]
        padded = getattr(be, "project_fwd_padded", None)
        if padded is not None:
            hp_local, a_dst = padded(x_local, weight, att_d, cfg, rows_alloc)
        else:
            hp_local, a_dst = be.project_fwd(x_local, weight, att_d, cfg)
            hp_local = _pad_rows(hp_local, rows_alloc)
        table = hp_local.new_empty((tm.total, hp_local.size(1)))
        piece = lambda t, s: t[tm.base[s]: tm.base[s] + tm.rows[s]]
        pend = [gather_piece(hp_local, tm, s, rank, piece(table, s), group, async_op=True)
                for s in range(k)]
        need_grad = any(ctx.needs_input_grad[:5])
        parts = []
        for s in range(k):
            with _Timed("wait_gather_h", table):
                pend[s].wait()
            if tm.rows[s] and n_loc:
                parts.append(be.attn_fwd_partial(layouts[s], _piece_cfg(cfg, s), f, piece(table, s),
                                                 a_dst, att_s))
        if parts:
            out, lse = be.attn_fwd_combine(parts, cfg, bias_c, need_grad)
        else:
            out = x_local.new_zeros((0, weight.size(0)))
            lse = x_local.new_zeros((0, cfg.heads))
        if need_grad:
            ctx.save_for_backward(x_local, weight, att_s, att_d, bias_c, table, hp_local, a_dst, out, lse)
            ctx.misc = (part, tm, layouts, cfg, be, group, att_src.shape)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x_local, weight, att_s, att_d, bias_c, table, hp_local, a_dst, out, lse = ctx.saved_tensors
        part, tm, layouts, cfg, be, group, att_shape = ctx.misc
        need_dx = ctx.needs_input_grad[0]
        f, k, rank, n_loc = x_local.size(1), tm.chunks, part.rank, part.n_local
        dp = table.size(1)
        piece = lambda t, s: t[tm.base[s]: tm.base[s] + tm.rows[s]]
        cb = None
        if n_loc:
            cb = be.chunked_backward(cfg, f, a_dst, att_s, att_d, bias_c, out, lse,
                                     grad_out.contiguous(), k)
        d_own = x_local.new_zeros((n_loc if tm.uneven else k * tm.sub[rank], dp))
        pend = []
        for s in range(k):
            if tm.rows[s] == 0:
                continue
            if cb is not None:
                d_piece = cb.chunk(s, layouts[s], _piece_cfg(cfg, s), piece(table, s))
            else:                                  # a rank without rows still takes part in the sums
                d_piece = x_local.new_zeros((tm.rows[s], dp))
            _, l0, cnt = tm.own(rank, s)
            pend.append(reduce_piece(d_piece, tm, s, rank, d_own[l0: l0 + (cnt if tm.uneven else tm.sub[rank])],
                                     group, async_op=True))
        if cb is not None:
            d_as, d_ad, d_b, d_a_dst = cb.end(hp_local[:n_loc])
        else:
            d_as, d_ad = torch.zeros_like(att_s), torch.zeros_like(att_d)
            d_b = None if bias_c is None else torch.zeros_like(bias_c)
            d_a_dst = x_local.new_zeros((0, cfg.heads))
        with _Timed("wait_scatter_dh", table):
            for p in pend:
                p.wait()
        d_hp = d_own[:n_loc]
        be.add_dst_term(d_hp, d_a_dst, att_d, cfg)
        d_w, d_x = be.project_bwd(x_local, weight, d_hp, cfg, need_dx)
        with _Timed("allreduce_params", table):
            all_reduce_many([d_w, d_as, d_ad, d_b], group)
        return (d_x, d_w, d_as.view(att_shape), d_ad.view(att_shape), d_b,
                None, None, None, None, None, None)


class _PartitionedInAggFn(torch.autograd.Function):
    """Input-side aggregation on a destination partition.  `table` is the packed source table
    of ALL nodes [padded_nodes, Fs]: exchange "xs" fills it by all-gathering the packed rows,
    exchange "as" keeps its x columns from an earlier call (static data) and all-gathers only
    the a_src columns.  The backward is local; the parameter gradients are all-reduced."""

    @staticmethod
    def forward(ctx, x_local, weight, att_src, att_dst, bias, part, layout, cfg, be, group,
                exchange, cache):
        x_local, weight = x_local.contiguous(), weight.contiguous()
        att_s, att_d = att_src.contiguous().view(-1), att_dst.contiguous().view(-1)
        bias_c = None if bias is None else bias.contiguous()
        xs_local, a_dst = be.inagg_pack(x_local, weight, att_s, att_d, cfg)
        if exchange == "xs":
            with _Timed("allgather_xs", xs_local):
                table = gather_rows(xs_local, part, group)
        else:
            c0, cn = be.inagg_logit_columns(x_local.size(1), cfg)
            # The cached table is valid for THIS x tensor object at THIS version only: a pointer
            # + shape key would also match a new tensor the allocator placed at a freed address
            # (the per-step dropout output PartitionedGAT feeds conv1), leaving stale x columns
            # under fresh logits.  A per-step temporary therefore never hits and takes the full
            # all-gather every step -- the same traffic as exchange="xs".
            owner = cache.get("owner")
            hit = (owner is not None and owner() is x_local and cache.get("version") == x_local._version
                   and cache.get("world") == part.world
                   and cache.get("bounds") == (part.bounds, part.uneven))
            # The two branches issue collectives of different sizes, so the ranks must agree on
            # which one runs: a rank whose x was rebuilt (or whose eval / train alternation differs)
            # would otherwise meet full-row gathers with logit-only ones -- a deadlock or a corrupt
            # table under RCCL instead of an error.  One 4-byte MIN all-reduce decides for everyone.
            if part.world > 1:
                flag = torch.tensor([1 if hit else 0], dtype=torch.int32, device=x_local.device)
                all_reduce_(flag, group, op=dist.ReduceOp.MIN)
                hit = bool(int(flag.item()))
            if not hit:                       # first call for this x: replicate the rows once
                cache["table"] = gather_rows(xs_local, part, group)
                cache["owner"], cache["version"] = weakref.ref(x_local), x_local._version
                cache["world"], cache["bounds"] = part.world, (part.bounds, part.uneven)
            else:                             # later calls: only the logits travel
                with _Timed("allgather_a_src", xs_local):
                    logits = gather_rows(xs_local[:, c0:c0 + cn].contiguous(), part, group)
                cache["table"][:, c0:c0 + cn] = logits
            table = cache["table"]
        # bf16 copy of the table for the edge kernels' gathers (cfg.rows16).  Exchange "as": its x
        # pieces are as static as the f32 table's, so it is made once per x tensor and only the
        # logits (f32, bytes 224.. of a 256-byte row = columns 56.. of the row seen as float32) are
        # rewritten per step; exchange "xs": converted per step from the gathered rows.
        xs16 = None
        if getattr(cfg, "rows16", False):
            if exchange == "as" and hit and cache.get("table16") is not None:
                xs16 = cache["table16"]
                xs16.view(torch.float32)[:, 56:56 + cn] = logits
            else:
                xs16 = be.inagg_table16(table, x_local.size(1), cfg)
                if exchange == "as":
                    cache["table16"] = xs16
        elif exchange == "as":
            cache["table16"] = None
        need_grad = any(ctx.needs_input_grad[1:5])
        kw = {} if xs16 is None else {"xs16": xs16}
        out, z, lse = be.inagg_fwd(layout, cfg, table, a_dst, weight, att_s, att_d, bias_c, need_grad, **kw)
        if need_grad:
            ctx.save_for_backward(table, a_dst, weight, att_s, att_d, bias_c, z, lse, xs16)
            ctx.misc = (layout, cfg, be, group, att_src.shape)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        table, a_dst, weight, att_s, att_d, bias_c, z, lse, xs16 = ctx.saved_tensors
        layout, cfg, be, group, att_shape = ctx.misc
        kw = {} if xs16 is None else {"xs16": xs16}
        d_w, d_as, d_ad, d_b = be.inagg_bwd(layout, cfg, table, a_dst, weight, att_s, att_d, bias_c,
                                            z, lse, grad_out.contiguous(), **kw)
        with _Timed("allreduce_params", d_w):
            all_reduce_many([d_w, d_as, d_ad, d_b], group)
        return (None, d_w, d_as.view(att_shape), d_ad.view(att_shape), d_b,
                None, None, None, None, None, None, None)


class PartitionedGATConv(nn.Module):
    """GATConv (model.py:26,28,30) whose forward takes this rank's rows of x.  Parameters
    are replicated (same seed / broadcast by the caller); their gradients come back
    all-reduced, so every rank steps identically."""

    def __init__(self, in_channels, out_channels, heads=1, negative_slope=0.2, dropout=0.0,
                 bias=True, backend=None, group=None, exchange="auto", balance="edges",
                 feature_dtype=None, chunks=None, padding="auto"):
        super().__init__()
        from .layer import _glorot_
        if balance not in ("edges", "nodes"):
            raise ValueError(f"balance must be 'edges' or 'nodes', got {balance!r}")
        self.balance = balance      # destination ranges of equal in-edge count / equal node count
        self.padding = padding      # make_partition: "auto" | "pad" | "exact"
        # exchange "h" in `chunks` pieces with the attention of piece k overlapping the collective of
        # piece k+1 (_PipelinedFn).  None: GAT_PIPELINE_CHUNKS, else 4 when there is anyone to talk to
        # (the first piece's transfer and the last piece's kernels are what stays exposed: ~1/4 each)
        if chunks is None:
            env = os.environ.get("GAT_PIPELINE_CHUNKS")
            chunks = int(env) if env else 0
        if not 0 <= int(chunks) <= _lib.MAX_PARTS:
            raise ValueError(f"chunks must be in [0, {_lib.MAX_PARTS}] (0: pick), got {chunks}")
        self.chunks = int(chunks)
        # torch.bfloat16: exchange "h" projects into a bf16 table, all-gathers THAT (half the xGMI
        # volume) and gathers bf16 rows; d_h travels back in f32.  Exchange "xs" / "as":
        # the edge kernels gather a bf16 copy of the packed table (what travels stays f32).  "x": f32.
        self.feature_dtype = feature_dtype
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.negative_slope, self.dropout = negative_slope, dropout
        self.lin = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(heads * out_channels)) if bias else None
        _glorot_(self.lin.weight), _glorot_(self.att_src), _glorot_(self.att_dst)
        self.backend = backend or HipBackend()
        self.group = group
        # "auto": input-side aggregation with the static rows replicated once ("as") when the
        # input carries no gradient and the kernels cover the shape, else all-gather h
        if exchange not in ("auto", "h", "x", "xs", "as"):
            raise ValueError(f"unknown exchange mode {exchange!r}")
        self.exchange = exchange
        self._layouts = None
        self._piece_layouts = None
        self._table_cache = {}

    def partition_for(self, edge_index: Tensor, num_nodes: int) -> Partition:
        """This rank's destination range for the graph (the caller slices x, y, grad_out by it)."""
        return make_partition(edge_index, num_nodes, dist.get_world_size(self.group),
                              dist.get_rank(self.group), self.balance, self.padding)

    def pieces(self, part: Partition) -> int:
        """how many pieces exchange "h" travels in"""
        if self.chunks:
            return self.chunks
        return 4 if part.world > 1 else 1

    def piece_layouts_for(self, edge_index: Tensor, part: Partition, tm: TableMap):
        """One rectangular layout per piece of the table (the edges whose sources the piece holds),
        built once per edge_index object like layout_for's."""
        owner, version, key, layouts = self._piece_layouts or (None, None, None, None)
        want = (part.rank, part.world, part.bounds, part.uneven, tm.chunks)
        if owner is None or owner() is not edge_index or version != edge_index._version or key != want:
            layouts = [self.backend.build_chunk_layout(edges, tm.rows[k], part.n_local)
                       if tm.rows[k] and part.n_local else None
                       for k, (edges, _) in enumerate(chunk_edge_lists(edge_index, part, tm))]
            self._piece_layouts = (weakref.ref(edge_index), edge_index._version, want, layouts)
        return layouts

    def layout_for(self, edge_index: Tensor, part: Partition):
        """The rank's CSR, built once per edge_index OBJECT (weak reference + version: an address
        reused by another tensor of the same shape does not match)."""
        owner, version, key, layout = self._layouts or (None, None, None, None)
        if (owner is None or owner() is not edge_index or version != edge_index._version
                or key != (part.rank, part.world, part.bounds, part.uneven)):
            layout = self.backend.build_layout(local_edges(edge_index, part), part)
            self._layouts = (weakref.ref(edge_index), edge_index._version,
                             (part.rank, part.world, part.bounds, part.uneven), layout)
        return layout

    def forward(self, x_local: Tensor, edge_index: Tensor, num_nodes: int) -> Tensor:
        from .ops import AttnConfig
        part = self.partition_for(edge_index, num_nodes)
        if x_local.size(0) != part.n_local:
            raise ValueError(f"rank {part.rank} owns nodes [{part.lo}, {part.hi}) but got "
                             f"{x_local.size(0)} rows of x")
        mask_mode, seed = _lib.MASK_NONE, 0
        if self.training and self.dropout > 0.0:
            mask_mode = _lib.MASK_HASH
            seed = int(torch.empty((), dtype=torch.int64).random_().item()) + part.rank
        cfg = AttnConfig(self.heads, self.out_channels, self.negative_slope, self.dropout,
                         mask_mode, seed, None)
        mode = self.resolve_exchange(x_local, cfg)
        pipelined = mode == "h" and self.pieces(part) > 1
        layout = None if pipelined else self.layout_for(edge_index, part)
        cfg.feat_bf16 = (mode == "h" and self.feature_dtype == torch.bfloat16
                         and getattr(self.backend, "bf16_supported", lambda f, c: False)(self.in_channels, cfg))
        cfg.rows16 = bool(mode in ("xs", "as") and self.feature_dtype == torch.bfloat16 and
                          getattr(self.backend, "inagg_rows16_supported", lambda f, c: False)(self.in_channels, cfg))
        if mode in ("xs", "as"):
            return _PartitionedInAggFn.apply(x_local, self.lin.weight, self.att_src, self.att_dst,
                                             self.bias, part, layout, cfg, self.backend,
                                             self.group, mode, self._table_cache)
        if pipelined:
            tm = part.table(self.pieces(part))
            return _PipelinedFn.apply(x_local, self.lin.weight, self.att_src, self.att_dst, self.bias,
                                      part, tm, self.piece_layouts_for(edge_index, part, tm), cfg,
                                      self.backend, self.group)
        return _PartitionedFn.apply(x_local, self.lin.weight, self.att_src, self.att_dst,
                                    self.bias, part, layout, cfg, self.backend, self.group, mode)

    def resolve_exchange(self, x_local: Tensor, cfg) -> str:
        x_grad = x_local.requires_grad and torch.is_grad_enabled()
        can = not x_grad and self.backend.inagg_supported(self.in_channels, cfg)
        if self.exchange == "auto":
            if can:
                return "as"
            dp = self.heads * ((self.out_channels + 3) // 4 * 4)
            return "x" if self.in_channels < dp else "h"   # fewer columns on the wire wins
        if self.exchange in ("xs", "as") and not can:
            raise RuntimeError(f"exchange={self.exchange!r} needs an input without gradient and a "
                               "shape the input-side aggregation kernels cover")
        return self.exchange


# ------------------------------------------------------------------ the whole network, partitioned
class _SumRanks(torch.autograd.Function):
    """y = sum over ranks of x (every rank gets y).  Each rank back-propagates its OWN share of
    the loss, so the gradient arriving at y on a rank is that rank's partial; the gradient of
    x is their sum: the backward is the same all-reduce."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return all_reduce_(x.contiguous().clone(), group)

    @staticmethod
    def backward(ctx, g):
        return all_reduce_(g.contiguous().clone(), ctx.group), None


def sync_elu_bn_dropout(x_local: Tensor, bn: nn.BatchNorm1d, p: float, training: bool,
                        num_nodes: int, group=None) -> Tensor:
    """dropout(bn(elu(x))) (/root/reference/gat/model.py:40-42) on a row partition: BatchNorm's
    batch statistics are those of ALL N rows (full-model caveat) -- two tiny
    all-reduces forward ([D] sums: mean, then centred squares, so no E[x^2] - mean^2
    cancellation) and their adjoints backward.  Running statistics update as nn.BatchNorm1d
    does (unbiased variance, momentum), identically on every rank.  Eval mode is local."""
    e = torch.nn.functional.elu(x_local)
    if training:
        mean = _SumRanks.apply(e.sum(0), group) / num_nodes
        cen = e - mean
        var = _SumRanks.apply((cen * cen).sum(0), group) / num_nodes
        with torch.no_grad():
            m = bn.momentum
            bn.running_mean.mul_(1 - m).add_(m * mean)
            bn.running_var.mul_(1 - m).add_(m * var * (num_nodes / max(num_nodes - 1, 1)))
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(1)
        y = cen * torch.rsqrt(var + bn.eps) * bn.weight + bn.bias
    else:
        y = (e - bn.running_mean) * torch.rsqrt(bn.running_var + bn.eps) * bn.weight + bn.bias
    return torch.nn.functional.dropout(y, p, training)


class _GlueExchange:
    """the two collectives gat_amd.glue.elu_bn_dropout_rows needs, on this module's group"""

    def __init__(self, group):
        self.group = group

    def gather(self, vec: Tensor) -> Tensor:
        world = dist.get_world_size(self.group)
        return all_gather_rows(vec.view(1, -1).contiguous(), world, self.group)

    def sum_(self, t: Tensor) -> Tensor:
        return all_reduce_(t, self.group)


class PartitionedGAT(nn.Module):
    """The reference's network (/root/reference/gat/model.py:26-31, 37-47) on one large graph
    split over the ranks by destination: rank r holds rows [lo, hi) of x and y.  The three
    layers are PartitionedGATConv (their parameter gradients come back all-reduced), the two
    BatchNorms use global batch statistics, the loss is the mean over all N nodes.  Every rank
    back-propagates its own rows' share of that mean; call `sync_grads()` before the optimizer
    step (BatchNorm's weight and bias gradients are partial sums until then).

    fused (default on the HIP backend): ELU -> BatchNorm -> dropout between the layers runs on the
    fused kernels of glue.hip -- two passes forward, two backward, with the per-column statistics
    (Chan's combination of every rank's rows / mean / M2) and the backward's two column sums
    exchanged between the passes -- and log_softmax + nll_loss on tail.hip (`forward_loss`).
    fused=False keeps the stock torch ops (what the CPU stand-in backend of the tests runs).
    feature_dtype=torch.bfloat16: bf16 feature tables in the layers (see PartitionedGATConv)."""

    def __init__(self, num_features, num_classes, hidden_dim=16, dropout=0.7, backend=None, group=None,
                 balance="edges", feature_dtype=None, chunks=None, padding="auto", fused=None):
        super().__init__()
        kw = dict(dropout=dropout, backend=backend, group=group, balance=balance,
                  feature_dtype=feature_dtype, chunks=chunks, padding=padding)
        self.conv1 = PartitionedGATConv(num_features, hidden_dim, heads=8, **kw)
        self.bn1 = nn.BatchNorm1d(hidden_dim * 8)
        self.conv2 = PartitionedGATConv(hidden_dim * 8, hidden_dim, heads=8, **kw)
        self.bn2 = nn.BatchNorm1d(hidden_dim * 8)
        self.conv3 = PartitionedGATConv(hidden_dim * 8, num_classes, heads=1, **kw)
        self.p, self.group = dropout, group
        self.fused = isinstance(self.conv1.backend, HipBackend) if fused is None else bool(fused)
        self._exchange = _GlueExchange(group)
        self._gens = {}

    def partition_for(self, edge_index: Tensor, num_nodes: int) -> Partition:
        return self.conv1.partition_for(edge_index, num_nodes)

    def _generator(self, device) -> torch.Generator:
        """A generator of this rank's own per device: ranks are seeded identically so that their
        parameters agree, and torch's global generator would then hand every rank the same mask
        pattern for its different rows."""
        gen = self._gens.get(device)
        if gen is None:
            gen = torch.Generator(device=device)
            gen.manual_seed(torch.initial_seed() + 7919 * (dist.get_rank(self.group) + 1))
            self._gens[device] = gen
        return gen

    def _drop(self, t: Tensor) -> Tensor:
        """Feature dropout (model.py:31,39,42,45) drawn from this rank's generator."""
        if not self.training or self.p <= 0.0:
            return t
        keep = torch.rand(t.shape, generator=self._generator(t.device), device=t.device) >= self.p
        return t * keep.to(t.dtype) / (1.0 - self.p)

    def _glue(self, x: Tensor, bn: nn.BatchNorm1d, num_nodes: int) -> Tensor:
        """dropout(bn(elu(x))) with the batch statistics of all N rows (model.py:40-42, 43-45)"""
        if self.fused:
            from .glue import elu_bn_dropout_rows
            seed = None
            if self.training and self.p > 0.0:      # a mask stream of this rank's own
                seed = int(torch.empty((), dtype=torch.int64).random_(
                    generator=self._generator(torch.device("cpu"))).item())
            return elu_bn_dropout_rows(x, bn, self.p, self.training, num_nodes, self._exchange, seed)
        return self._drop(sync_elu_bn_dropout(x, bn, 0.0, self.training, num_nodes, self.group))

    def logits(self, x_local: Tensor, edge_index: Tensor, num_nodes: int) -> Tensor:
        x = self.conv1(self._drop(x_local), edge_index, num_nodes)
        x = self.conv2(self._glue(x, self.bn1, num_nodes), edge_index, num_nodes)
        return self.conv3(self._glue(x, self.bn2, num_nodes), edge_index, num_nodes)

    def forward(self, x_local: Tensor, edge_index: Tensor, num_nodes: int) -> Tensor:
        return torch.log_softmax(self.logits(x_local, edge_index, num_nodes), dim=1)

    def forward_loss(self, x_local: Tensor, edge_index: Tensor, num_nodes: int, y_local: Tensor):
        """(log-probabilities of this rank's rows, this rank's share of the loss): model.py:47 and :53
        in one pass forward and one backward on the fused tail when `fused`."""
        z = self.logits(x_local, edge_index, num_nodes)
        if self.fused and z.size(0) > 0:
            from .glue import log_softmax_nll
            logp, mean_local = log_softmax_nll(z, y_local)       # mean over this rank's rows
            return logp, mean_local * (z.size(0) / num_nodes)
        logp = torch.log_softmax(z, dim=1)
        return logp, self.loss_share(logp, y_local, num_nodes)

    def loss_share(self, logp_local: Tensor, y_local: Tensor, num_nodes: int) -> Tensor:
        """This rank's share of F.nll_loss(out, y) over all N nodes (model.py:53): the shares of
        all ranks add up to the reference's loss; back-propagate the share."""
        return torch.nn.functional.nll_loss(logp_local, y_local, reduction="sum") / num_nodes

    def global_loss(self, share: Tensor) -> Tensor:
        return all_reduce_(share.detach().clone(), self.group)

    def sync_grads(self) -> None:
        all_reduce_many([p.grad for bn in (self.bn1, self.bn2) for p in (bn.weight, bn.bias)],
                        group=self.group)


# ------------------------------------------------------------------ bench.py --gpus N
EXCHANGE_TEXT = {
    "h": "all-gather(h) fwd, reduce-scatter(d_h) + all-reduce(param grads) bwd",
    "x": "all-gather(x) + full projection fwd, all-reduce(param grads) bwd",
    "xs": "input-side aggregation; all-gather(packed rows x|a_src) fwd every step, "
          "all-reduce(param grads) bwd",
    "as": "input-side aggregation; static x rows replicated once before the timed steps, "
          "all-gather(a_src [N,H]) fwd every step, all-reduce(param grads) bwd",
}


def _timed_steps(step, steps, warmup, dev):
    """bench.py's contract: warm-up, then exactly `steps` steps between barrier + synchronize on
    both sides, MAX over ranks.  -> (seconds, {kernel tag: mean ms on this rank},
    {collective: mean ms per step this rank's stream spent in / waiting for it})"""
    for _ in range(warmup):
        step()
    _lib.prof_read()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    _lib.prof_enable(True)          # per-kernel HIP events, as in the single-GPU run
    comm_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    _lib.prof_enable(False)
    comm = {k: round(v / steps, 4) for k, v in sorted(comm_ms_read().items())}
    comm_timing(False)
    all_reduce_(el, op=dist.ReduceOp.MAX)
    per = {}
    for tag, ms in _lib.prof_read():
        per.setdefault(tag, []).append(ms)
    # a kernel launched once per piece shows up `pieces` times per step: report its per-step total
    return float(el.item()), {k: sum(v) / steps for k, v in per.items()}, comm


def _recorded_traffic_ratio(workload: str, kernel: str, full_alg_bytes):
    """HBM traffic / algorithmic bytes of `kernel` from the committed one-GPU PMC passes
    (profiles/rN/pmc_traffic.json; full_alg_bytes = the kernel's algorithmic bytes on the whole
    graph, what those passes ran on) -> (ratio or None, source)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = workload[:-len("_hidden")] if workload.endswith("_hidden") else workload
    for rnd in ("r3", "r2", "r1"):
        path = os.path.join(root, "profiles", rnd, "pmc_traffic.json")
        try:
            with open(path) as f:
                t = json.load(f)
            if t.get("workload") == base and t["kernels"].get(kernel) and full_alg_bytes:
                return t["kernels"][kernel] / full_alg_bytes, f"profiles/{rnd}/pmc_traffic.json"
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def bench_partitioned(args, rank: int, world: int, dev) -> None:
    """One products-shaped layer, destination-partitioned over `world` GPUs; rank 0 prints
    bench.py's JSON line.  Every rank regenerates the same graph from the seed and keeps
    its own destination range.  `value` is the mode --exchange names (auto: what the layer picks
    by itself); the line also carries the north_star's own pattern -- all-gather of the projected
    features + all-reduce of the weight gradients, exchange="h" -- timed the same way: in pieces
    with the attention overlapping the collectives ("exchange_h") and as one blocking collective
    per direction ("exchange_h_one_piece"), each with the time spent in its collectives."""
    from . import synth
    from .ops import AttnConfig
    ei, n, f, h, c = synth.workload_graph(args.workload, args.seed, dev)
    balance = getattr(args, "partition", "edges")
    padding = getattr(args, "padding", "auto")
    chunks = int(getattr(args, "chunks", 0) or 0)
    part = make_partition(ei, n, world, rank, balance, padding)
    g = torch.Generator(device=dev).manual_seed(args.seed)
    x = torch.randn(n, f, generator=g, device=dev)[part.lo:part.hi].contiguous()
    go = torch.randn(n, h * c, generator=g, device=dev)[part.lo:part.hi].contiguous()
    hidden = args.workload.endswith("_hidden")
    x.requires_grad_(hidden)
    torch.manual_seed(args.seed)
    bf16 = getattr(args, "dtype", "f32") == "bf16"
    conv = PartitionedGATConv(f, c, heads=h, dropout=args.dropout, balance=balance, padding=padding,
                              exchange=getattr(args, "exchange", "auto"), chunks=chunks,
                              feature_dtype=torch.bfloat16 if bf16 else None).to(dev).train()
    mode = conv.resolve_exchange(x, AttnConfig(h, c))
    pieces = conv.pieces(part) if mode == "h" else 1
    torch.cuda.synchronize()
    t0 = time.time()
    if mode == "h" and pieces > 1:
        layouts = conv.piece_layouts_for(ei, part, part.table(pieces))
        n_edges = sum(l.num_edges for l in layouts if l is not None)
    else:
        n_edges = conv.layout_for(ei, part).num_edges
    torch.cuda.synchronize()
    t_layout = time.time() - t0
    # every rank's (edges, nodes): one all-reduce of a [world, 2] table with the own row filled in.
    # It is also the first collective of the process and runs on the main thread -- communicator
    # set-up must not happen inside the backward (autograd's thread), where the first all-reduce
    # of a gloo rehearsal was seen to hang.
    shares = torch.zeros(world, 2, device=dev, dtype=torch.int64)
    shares[rank, 0], shares[rank, 1] = n_edges, part.n_local
    all_reduce_(shares)
    shares = shares.tolist()
    ep = int(sum(s[0] for s in shares))

    trace = bool(os.environ.get("GAT_TRACE"))

    def step():
        conv.zero_grad(set_to_none=True)
        x.grad = None
        out = conv(x, ei, n)
        if trace:
            torch.cuda.synchronize()
            print(f"[rank {rank}] {time.time() % 1000:.2f} forward done", file=sys.stderr, flush=True)
        out.backward(go)
        if trace:
            torch.cuda.synchronize()
            print(f"[rank {rank}] {time.time() % 1000:.2f} backward done", file=sys.stderr, flush=True)

    def region(mode_, pieces_):
        """time one exchange mode -> dict for the line"""
        conv.exchange, conv.chunks = mode_, pieces_
        conv._table_cache.clear()
        el, avg_, comm_ = _timed_steps(step, args.steps, args.warmup, dev)
        fb = 2 if ("proj16_fwd_gemm" in avg_ or (bf16 and mode_ in ("xs", "as")
                   and conv.backend.inagg_rows16_supported(f, AttnConfig(h, c)))) else 4
        dp = h * ((c + 3) // 4 * 4)
        width = {"h": dp * fb, "x": f * 4, "xs": (-(-f // 4) * 4 + -(-h // 4) * 4) * 4, "as": -(-h // 4) * 4 * 4}[mode_]
        tm_ = part.table(pieces_)
        return {"exchange": mode_, "pieces": pieces_, "ms_per_step": round(el / args.steps * 1e3, 4),
                "value": ep / (el / args.steps), "unit": "edges/s", "parallelism": EXCHANGE_TEXT[mode_],
                "comm_ms": comm_, "comm_ms_total": round(sum(comm_.values()), 4),
                "comm_note": "milliseconds per step this rank's compute stream spent inside / waiting for each "
                             "collective (HIP events around it); with pieces > 1 the wait_* entries are what the "
                             "overlap did NOT hide",
                "gathered_table": {"rows": tm_.total, "rows_per_node": round(tm_.total / max(n, 1), 4),
                                   "bytes": tm_.total * width,
                                   "received_bytes_per_rank_fwd": (tm_.total - sum(tm_.cap[s][rank] for s in range(pieces_))) * width},
                "kernels": {k: {"avg_ms": round(v, 4)} for k, v in sorted(avg_.items())},
                "_avg": avg_, "_fb": fb}

    main_r = region(mode, pieces)
    avg = main_r["_avg"]
    # roofline of rank 0's dominant kernel on ITS share: local destination rows, local edges
    kb = synth.kernel_bytes(part.n_local, int(shares[rank][0]), h, h * c, f, feat_bytes=main_r["_fb"])
    roofline = None
    if any(k in avg for k in kb):
        dom = max(kb, key=lambda k: avg.get(k, 0.0))
        achieved = kb[dom] / (avg[dom] * 1e-3) / 1e9
        full = synth.kernel_bytes(n, ep, h, h * c, f, feat_bytes=main_r["_fb"]).get(dom)
        ratio, src = _recorded_traffic_ratio(args.workload, dom + ("_bf16" if main_r["_fb"] == 2 else ""), full)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": 8000.0,
                    "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
                    "traffic": None if ratio is None else int(kb[dom] * ratio),
                    "traffic_source": None if ratio is None else
                    f"{src}: traffic / algorithmic bytes of this kernel in the recorded one-GPU rocprofv3 --pmc "
                    f"passes ({ratio:.3f}) applied to this rank's algorithmic bytes; not measured in this run",
                    "avg_ms": round(avg[dom], 4), "algorithmic_bytes": kb[dom],
                    "scope": f"rank 0 of {world}: {part.n_local} destination rows, {int(shares[rank][0])} edges"
                             + (f"; {main_r['pieces']} launches per step (one per piece), avg_ms is their sum"
                                if main_r["pieces"] > 1 else "")}
    others = {}
    if not getattr(args, "no_exchange_h", False):
        hp = chunks or (4 if world > 1 else 1)
        if not (mode == "h" and pieces == hp):
            others["exchange_h"] = region("h", hp)
        if hp > 1 and not (mode == "h" and pieces == 1):
            others["exchange_h_one_piece"] = region("h", 1)
    if rank == 0:
        strip = lambda d: {k: v for k, v in d.items() if not k.startswith("_")}
        res = {"metric": "GAT layer fwd+bwd edges/sec", "value": main_r["value"],
               "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": main_r["ms_per_step"], "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None,
               "dtype": "bf16" if main_r["_fb"] == 2 else "f32", "data": "synthetic",
               "config": {"workload": f"{args.workload}-shaped synthetic graph: {n} nodes, "
                                      f"{ei.size(1)} edges ({ep} after self-loop normalisation), one GAT "
                                      f"layer {f}->{h}x{c}{' (hidden layer: d_x computed)' if hidden else ''}, "
                                      f"attention dropout {args.dropout} (train mode)",
                          "nodes": n, "edges": ep, "in_features": f, "heads": h, "channels": c,
                          "parallelism": f"destination partition x{world} (ranges balanced by "
                                         f"{'in-edge' if balance == 'edges' else 'node'} count), "
                                         f"exchange={mode}"
                                         + (f" in {pieces} pieces, attention overlapped with the collectives" if pieces > 1 else "")
                                         + ": " + EXCHANGE_TEXT[mode] + f", {dist.get_backend()}",
                          "partition": {"balance": balance,
                                        "tables": "exact (per-owner collectives)" if part.uneven else
                                                  "padded to the longest range (equal-shard collectives)",
                                        "padded_nodes_over_n": round(part.padding_ratio, 4),
                                        "edges_per_rank": [int(s[0]) for s in shares],
                                        "nodes_per_rank": [int(s[1]) for s in shares]},
                          "graph_layout_build_s": round(t_layout, 4)},
               "roofline": roofline,
               "pieces": main_r["pieces"],
               "comm_ms": main_r["comm_ms"], "comm_ms_total": main_r["comm_ms_total"],
               "gathered_table": main_r["gathered_table"],
               "kernels": main_r["kernels"]}
        res.update({k: strip(v) for k, v in others.items()})
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        # the reference's CPU path next to it (north_star): the oracle on this host's cores, after the
        # process group is gone so that no rank waits in a collective meanwhile
        res["cpu_baseline"] = None
        if not getattr(args, "no_cpu_baseline", False):
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            if root not in sys.path:
                sys.path.insert(0, root)
            import bench as _bench
            res["cpu_baseline"] = _bench.cpu_baseline(args.workload[:-len("_hidden")] if hidden else args.workload,
                                                      args.seed, budget_s=15.0)
        print(json.dumps(res))
