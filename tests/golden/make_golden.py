"""Writes tests/golden/*.npz: frozen input/output vectors of the oracle.

PARITY UNPINNED: the reference's own arithmetic (torch_geometric.nn.GATConv 2.5.3) is
not importable here and the reference holds no fixtures, so these vectors
come from oracle/gat_oracle.py -- they freeze the oracle, they do not pin it to the
reference.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import cases  # noqa: E402
from oracle import gat_oracle as O  # noqa: E402

PAIRS = [("knn", "conv1_h30", 0.425), ("multigraph", "cora_l1", 0.0), ("star", "arxiv_h64", 0.4),
         ("multigraph", "conv3_cls2", 0.425), ("hub_source", "odd_c40", 0.0),
         ("chain", "cora_l2", 0.35)]


def main():
    torch.set_num_threads(1)
    for gname, sname, p in PAIRS:
        ei, n = cases.GRAPHS[gname]()
        f, h, c = cases.SHAPES[sname]
        x, w, a_s, a_d, b, go = cases.layer_inputs(n, f, h, c)
        csr = O.build_csr(ei, n)
        mask = cases.keep_mask(csr.num_edges, h, p) if p > 0 else None
        kw = dict(dropout=p, mask=mask, training=p > 0)
        out, grads = O.gatconv_grads_autograd(x, ei, w, a_s, a_d, b, h, c, go, **kw)
        _, inter = O.gatconv_forward(x, ei, w, a_s, a_d, b, h, c, return_intermediates=True, **kw)
        np.savez_compressed(
            os.path.join(HERE, f"{gname}__{sname}.npz"),
            edge_index=ei.numpy(), num_nodes=n, heads=h, channels=c, dropout=p,
            x=x.numpy(), weight=w.numpy(), att_src=a_s.numpy(), att_dst=a_d.numpy(),
            bias=b.numpy(), grad_out=go.numpy(),
            mask=(mask.numpy() if mask is not None else np.zeros((0, h), dtype=bool)),
            norm_edge_index=inter["edge_index"].numpy(), rowptr=csr.rowptr, col=csr.col,
            perm=csr.perm, rowptr_t=csr.rowptr_t, col_t=csr.col_t, pos_t=csr.pos_t,
            alpha=inter["alpha"].detach().numpy(), out=out.numpy(),
            d_x=grads["x"].numpy(), d_weight=grads["weight"].numpy(),
            d_att_src=grads["att_src"].numpy(), d_att_dst=grads["att_dst"].numpy(),
            d_bias=grads["bias"].numpy())
        print("wrote", gname, sname)


if __name__ == "__main__":
    main()
