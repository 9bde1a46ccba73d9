"""CPU oracle for the GAT layer hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product package ``gat_amd`` never does.

PARITY UNPINNED.  The arithmetic of the path lives in the third-party
``torch_geometric.nn.GATConv`` (pinned 2.5.3, /root/reference/poetry.lock:1937-1938,
running on torch 2.4.1, poetry.lock:1883-1884, without torch-scatter/pyg-lib), which
is neither under /root/reference nor installable here, and the reference holds no
tests, fixtures or golden vectors for it.  What follows is
a restatement of that library's published algorithm, anchored on the reference's own
call sites:

* constructor arguments in force   /root/reference/gat/model.py:26,28,30
  (heads, concat=True, dropout=p; every other GATConv argument at its default:
  negative_slope=0.2, add_self_loops=True, bias=True, edge_dim=None, aggr='add')
* forward(x, edge_index)           /root/reference/gat/model.py:40,43,46
* backward through autograd        /root/reference/gat/model.py:54
* edge_index orientation/dtype     /root/reference/gat/converter.py:7-15
  (row 0 = source j, row 1 = destination i, int64)

Two independent statements of the same mathematics are kept and cross-checked by
``tests/test_oracle.py``:

(i)  ``gatconv_forward`` -- op for op the tensor program GATConv executes on the CPU
     (index_select / scatter amax / scatter_add), differentiated by autograd;
(ii) ``gatconv_backward_closed_form`` -- the explicit gradient formulae of SURVEY.md
     section 8a row a11, sharing no code with autograd.

The integer part (self-loop normalisation, destination-sorted CSR, transposed CSR) is
restated here with numpy and, independently, in plain C in ``oracle/csr_ref.c``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

NEG_SLOPE_DEFAULT = 0.2  # GATConv default negative_slope (model.py:26 passes none)


# ----------------------------------------------------------------------------------
# Integer part: edge list normalisation and CSR layouts (must be bit exact)
# ----------------------------------------------------------------------------------
def normalize_edges(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """remove_self_loops followed by add_self_loops, as GATConv.forward does when
    add_self_loops=True (the default in force at model.py:26,28,30).

    Kept edges stay in their original order; one (n, n) edge per node n = 0..N-1 is
    appended at the END.  Nothing is coalesced: parallel edges survive.
    """
    src, dst = edge_index[0], edge_index[1]
    keep = src != dst
    loops = torch.arange(num_nodes, dtype=edge_index.dtype)
    return torch.stack([torch.cat([src[keep], loops]), torch.cat([dst[keep], loops])])


@dataclass
class CSRLayout:
    """Destination-sorted CSR of the normalised edge list plus its transpose.

    rowptr[i]..rowptr[i+1]  in-edges of destination i, in normalised-list order
    col[e]                   source of CSR edge e
    perm[e]                  index of CSR edge e in the normalised edge list
    rowptr_t / col_t         the same edges grouped by source; col_t = destination;
                             inside a source row edges are ordered by CSR position
    pos_t[t]                 CSR position e of transposed edge t
    """
    num_nodes: int
    num_edges: int
    rowptr: np.ndarray
    col: np.ndarray
    perm: np.ndarray
    rowptr_t: np.ndarray
    col_t: np.ndarray
    pos_t: np.ndarray


def build_csr(edge_index: torch.Tensor, num_nodes: int, add_self_loops: bool = True) -> CSRLayout:
    """Stable sort of the normalised edge list by destination (then by source for the
    transpose).  numpy restatement; ``oracle/csr_ref.c`` is the plain-C twin."""
    ei = normalize_edges(edge_index, num_nodes) if add_self_loops else edge_index
    src = ei[0].numpy().astype(np.int64)
    dst = ei[1].numpy().astype(np.int64)
    e = src.shape[0]
    perm = np.argsort(dst, kind="stable").astype(np.int32)
    col = src[perm].astype(np.int32)
    rowptr = np.zeros(num_nodes + 1, dtype=np.int32)
    np.cumsum(np.bincount(dst, minlength=num_nodes), out=rowptr[1:])
    pos_t = np.argsort(col, kind="stable").astype(np.int32)
    col_t = dst[perm][pos_t].astype(np.int32)
    rowptr_t = np.zeros(num_nodes + 1, dtype=np.int32)
    np.cumsum(np.bincount(src, minlength=num_nodes), out=rowptr_t[1:])
    return CSRLayout(num_nodes, e, rowptr, col, perm, rowptr_t, col_t, pos_t)


# ----------------------------------------------------------------------------------
# Parameter initialisation (GATConv.reset_parameters)
# ----------------------------------------------------------------------------------
def glorot_(t: torch.Tensor, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Glorot-uniform over the last two dims: bound = sqrt(6 / (size(-2)+size(-1)))."""
    bound = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        return t.uniform_(-bound, bound, generator=generator)


def init_params(in_channels: int, heads: int, out_channels: int, seed: int = 0,
                dtype=torch.float32):
    """lin.weight [H*C, F] glorot, att_src/att_dst [1,H,C] glorot, bias [H*C] zeros."""
    g = torch.Generator().manual_seed(seed)
    w = glorot_(torch.empty(heads * out_channels, in_channels, dtype=dtype), g)
    a_s = glorot_(torch.empty(1, heads, out_channels, dtype=dtype), g)
    a_d = glorot_(torch.empty(1, heads, out_channels, dtype=dtype), g)
    b = torch.zeros(heads * out_channels, dtype=dtype)
    return w, a_s, a_d, b


# ----------------------------------------------------------------------------------
# (i) op-for-op forward, differentiable by autograd
# ----------------------------------------------------------------------------------
def segment_softmax(z: torch.Tensor, index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """torch_geometric.utils.softmax without torch-scatter: detached scatter-amax,
    gather, sub, exp, scatter-add (+1e-16), gather, div."""
    shape = (num_nodes,) + tuple(z.shape[1:])
    idx = index.view(-1, *([1] * (z.dim() - 1))).expand_as(z)
    z_max = torch.full(shape, float("-inf"), dtype=z.dtype).scatter_reduce_(
        0, idx, z.detach(), reduce="amax", include_self=True)
    out = (z - z_max.index_select(0, index)).exp()
    out_sum = torch.zeros(shape, dtype=z.dtype).scatter_add_(0, idx, out) + 1e-16
    return out / out_sum.index_select(0, index)


class _RoundBF16(torch.autograd.Function):
    """y = x rounded to bfloat16 (nearest even), gradient passed straight through."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGradBF16(torch.autograd.Function):
    """y = x; the gradient arriving at y is rounded to bfloat16 on its way to x."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def gatconv_forward(x, edge_index, weight, att_src, att_dst, bias, heads, out_channels,
                    negative_slope: float = NEG_SLOPE_DEFAULT, dropout: float = 0.0,
                    training: bool = False, mask: Optional[torch.Tensor] = None,
                    add_self_loops: bool = True, concat: bool = True,
                    return_intermediates: bool = False, bf16_tables: bool = False,
                    bf16_matmul: bool = False):
    """One GATConv.forward (model.py:40,43,46).

    ``mask`` -- optional {0,1} tensor [E', H] in normalised-edge-list order.  Given,
    attention dropout is alpha * mask / (1 - p) (what F.dropout computes for that
    Bernoulli draw); absent, F.dropout draws its own mask when ``training``.
    """
    n = x.size(0)
    h_, c_ = heads, out_channels
    if bf16_matmul:      # bf16 MFMA projection: operands rounded to bf16, products accumulated wider
        h = F.linear(_RoundBF16.apply(x), _RoundBF16.apply(weight))
    else:
        h = F.linear(x, weight)                                  # lin(x), no bias
    if bf16_tables:      # the feature rows are STORED as bf16 (the reference has no such mode: this
        h = _RoundBF16.apply(h)   # models libgat_mi355's bf16 feature tables, rounding points included)
    h = h.view(n, h_, c_)
    a_s = (h * att_src.view(1, h_, c_)).sum(dim=-1)              # [N, H]
    a_d = (h * att_dst.view(1, h_, c_)).sum(dim=-1)
    ei = normalize_edges(edge_index, n) if add_self_loops else edge_index
    src, dst = ei[0], ei[1]
    u = a_s.index_select(0, src) + a_d.index_select(0, dst)      # [E', H]
    z = F.leaky_relu(u, negative_slope)
    alpha = segment_softmax(z, dst, n)
    if mask is not None:
        alpha_t = alpha * (mask.to(alpha.dtype) / (1.0 - dropout))
    else:
        alpha_t = F.dropout(alpha, p=dropout, training=training)
    msg = alpha_t.unsqueeze(-1) * h.index_select(0, src)         # [E', H, C]
    out = torch.zeros(n, h_, c_, dtype=x.dtype).scatter_add_(
        0, dst.view(-1, 1, 1).expand_as(msg), msg)
    if bf16_tables:      # the backward gathers grad_out rows from a bf16 copy
        out = _RoundGradBF16.apply(out)
    out = out.view(n, h_ * c_) if concat else out.mean(dim=1)
    if bias is not None:
        out = out + bias
    if return_intermediates:
        return out, dict(h=h, a_src=a_s, a_dst=a_d, edge_index=ei, u=u, alpha=alpha,
                         alpha_tilde=alpha_t)
    return out


# ----------------------------------------------------------------------------------
# (ii) closed-form backward, no autograd
# ----------------------------------------------------------------------------------
def gatconv_backward_closed_form(x, edge_index, weight, att_src, att_dst, heads,
                                 out_channels, grad_out, negative_slope=NEG_SLOPE_DEFAULT,
                                 dropout: float = 0.0, mask: Optional[torch.Tensor] = None,
                                 add_self_loops: bool = True):
    """Gradients of L = <out, grad_out> wrt x, weight, att_src, att_dst, bias
    (concat=True).  Recomputes the forward internally."""
    n = x.size(0)
    h_, c_ = heads, out_channels
    with torch.no_grad():
        h = (x @ weight.t()).view(n, h_, c_)
        asrc = att_src.view(1, h_, c_)
        adst = att_dst.view(1, h_, c_)
        a_s = (h * asrc).sum(-1)
        a_d = (h * adst).sum(-1)
        ei = normalize_edges(edge_index, n) if add_self_loops else edge_index
        src, dst = ei[0], ei[1]
        u = a_s[src] + a_d[dst]
        z = torch.where(u > 0, u, negative_slope * u)
        alpha = segment_softmax(z, dst, n)
        keep = torch.ones_like(alpha) if mask is None else mask.to(alpha.dtype) / (1.0 - dropout)
        alpha_t = alpha * keep
        g = grad_out.view(n, h_, c_)
        d_bias = g.sum(0).reshape(-1)
        d_alpha_t = (g[dst] * h[src]).sum(-1)                       # [E', H]
        d_h = torch.zeros_like(h).index_add_(0, src, alpha_t.unsqueeze(-1) * g[dst])
        d_alpha = d_alpha_t * keep
        row_dot = torch.zeros(n, h_, dtype=x.dtype).index_add_(0, dst, alpha * d_alpha)
        d_z = alpha * (d_alpha - row_dot[dst])
        d_u = d_z * torch.where(u > 0, torch.ones_like(u), torch.full_like(u, negative_slope))
        d_a_d = torch.zeros(n, h_, dtype=x.dtype).index_add_(0, dst, d_u)
        d_a_s = torch.zeros(n, h_, dtype=x.dtype).index_add_(0, src, d_u)
        d_att_src = (d_a_s.unsqueeze(-1) * h).sum(0, keepdim=True)
        d_att_dst = (d_a_d.unsqueeze(-1) * h).sum(0, keepdim=True)
        d_h = d_h + d_a_s.unsqueeze(-1) * asrc + d_a_d.unsqueeze(-1) * adst
        d_hf = d_h.reshape(n, h_ * c_)
        d_w = d_hf.t() @ x
        d_x = d_hf @ weight
    return dict(x=d_x, weight=d_w, att_src=d_att_src, att_dst=d_att_dst, bias=d_bias,
                h=d_h, a_src=d_a_s, a_dst=d_a_d)


def gatconv_grads_autograd(x, edge_index, weight, att_src, att_dst, bias, heads,
                           out_channels, grad_out, **kw):
    """Forward + autograd backward of (i); returns (out, grads dict)."""
    xs = x.detach().clone().requires_grad_(True)
    ps = [t.detach().clone().requires_grad_(True) for t in (weight, att_src, att_dst, bias)]
    out = gatconv_forward(xs, edge_index, ps[0], ps[1], ps[2], ps[3], heads, out_channels, **kw)
    out.backward(grad_out)
    return out.detach(), dict(x=xs.grad, weight=ps[0].grad, att_src=ps[1].grad,
                              att_dst=ps[2].grad, bias=ps[3].grad)


# ----------------------------------------------------------------------------------
# nn.Module faces of the oracle: one layer, and the 3-layer model of model.py:18-47
# ----------------------------------------------------------------------------------
class OracleGATConv(torch.nn.Module):
    """GATConv as constructed at model.py:26,28,30 -- parameters named as in
    torch-geometric 2.5.x: att_src, att_dst, bias, lin.weight."""

    def __init__(self, in_channels, out_channels, heads=1, concat=True,
                 negative_slope=NEG_SLOPE_DEFAULT, dropout=0.0, add_self_loops=True, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.negative_slope, self.dropout = concat, negative_slope, dropout
        self.add_self_loops = add_self_loops
        self.lin = torch.nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_src = torch.nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = torch.nn.Parameter(torch.empty(1, heads, out_channels))
        if bias:
            self.bias = torch.nn.Parameter(torch.empty(heads * out_channels if concat else out_channels))
        else:
            self.register_parameter("bias", None)
        self.injected_mask = None  # test hook: [E', H] keep-mask in normalised edge order
        self.reset_parameters()

    def reset_parameters(self):
        glorot_(self.lin.weight)
        glorot_(self.att_src)
        glorot_(self.att_dst)
        if self.bias is not None:
            torch.nn.init.zeros_(self.bias)

    def forward(self, x, edge_index):
        mask = self.injected_mask if self.training else None
        return gatconv_forward(x, edge_index, self.lin.weight, self.att_src, self.att_dst,
                               self.bias, self.heads, self.out_channels, self.negative_slope,
                               self.dropout, self.training, mask, self.add_self_loops, self.concat)


class OracleGAT(torch.nn.Module):
    """The network of model.py:26-31 and its forward, model.py:37-47."""

    def __init__(self, num_features, num_classes, hidden_dim=16, dropout=0.7):
        super().__init__()
        self.conv1 = OracleGATConv(num_features, hidden_dim, heads=8, dropout=dropout)
        self.bn1 = torch.nn.BatchNorm1d(hidden_dim * 8)
        self.conv2 = OracleGATConv(hidden_dim * 8, hidden_dim, heads=8, dropout=dropout)
        self.bn2 = torch.nn.BatchNorm1d(hidden_dim * 8)
        self.conv3 = OracleGATConv(hidden_dim * 8, num_classes, heads=1, concat=True, dropout=dropout)
        self.dropout = torch.nn.Dropout(p=dropout)

    def forward(self, x, edge_index):
        x = self.dropout(x)
        x = F.elu(self.conv1(x, edge_index))
        x = self.bn1(x)
        x = self.dropout(x)
        x = F.elu(self.conv2(x, edge_index))
        x = self.bn2(x)
        x = self.dropout(x)
        x = self.conv3(x, edge_index)
        return F.log_softmax(x, dim=1)


# ----------------------------------------------------------------------------------
# Algorithmic traffic model -- shared by bench.py and DESIGN.md
# ----------------------------------------------------------------------------------
def algorithmic_bytes(num_nodes: int, num_edges: int, heads: int, channels: int) -> dict:
    """Bytes one layer must move (fp32 values, int32 indices), per SURVEY.md 8d."""
    d = heads * channels
    fwd = num_edges * (4 * d + 4 * heads + 4) + num_nodes * (4 * d + 12 * heads + 4)
    bwd = num_edges * (8 * d + 16 * heads + 12) + num_nodes * (8 * d + 16 * heads)
    return dict(fwd=fwd, bwd=bwd, total=fwd + bwd)


def elu_bn_dropout_ref(x, bn, mask=None, p=0.0):
    """The reference's glue between two layers, /root/reference/gat/model.py:40-42 (and :43-45):
    x = F.elu(x); x = bn(x); x = dropout(x) -- with the dropout's Bernoulli draw replaced by an
    explicit keep-mask (the same trick as the attention dropout).  `bn` is a torch
    nn.BatchNorm1d in the wanted mode; its running statistics are updated as the module does."""
    import torch.nn.functional as F
    y = bn(F.elu(x))
    if bn.training and p > 0.0 and mask is not None:
        y = y * mask.to(y.dtype) / (1.0 - p)
    return y


def log_softmax_nll_ref(logits, target):
    """(log-probabilities, mean nll loss) as /root/reference/gat/model.py:47 and :53 compute them
    (stock CPU torch ops; test infrastructure like the rest of this file)."""
    import torch.nn.functional as F
    logp = F.log_softmax(logits, dim=1)
    return logp, F.nll_loss(logp, target)
