"""Bisect the d_W mismatch at 12 -> 8x64 on a 5-neighbour graph."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import cases
from oracle import gat_oracle as O
from gat_amd import _lib, synth
from test_gpu_parity import run_layer
cuda = torch.device("cuda:0")

def trial(n, f, h, c, p, opts, graph="knn5"):
    ei = synth.out_degree_k_graph(n, 5, seed=n) if graph == "knn5" else torch.zeros(2, 0, dtype=torch.long)
    inputs = cases.layer_inputs(n, f, h, c, seed=n + 7)
    x, w, a_s, a_d, b, go = inputs
    ne = O.build_csr(ei, n).num_edges
    mask = cases.keep_mask(ne, h, p) if p > 0 else None
    ref_out, ref_g = O.gatconv_grads_autograd(x, ei, w, a_s, a_d, b, h, c, go, dropout=p, mask=mask, training=p > 0)
    for k, v in opts.items():
        _lib.set_option(k, v)
    try:
        out, g = run_layer(cuda, ei, n, f, h, c, p, mask, 64, inputs, inagg=True, stream=False)
    finally:
        for k in opts:
            _lib.set_option(k, -1)
    dw = g["weight"].cpu()
    err = (dw - ref_g["weight"]).abs()
    rel = float(err.max() / ref_g["weight"].abs().max())
    rows = (err.view(h, c, f).amax(dim=2) > 1e-4 * float(ref_g["weight"].abs().max())).nonzero()
    print(f"n={n} {f}->{h}x{c} p={p} {graph} {opts}: d_W max-norm err {rel:.2e}; bad (head, channel) pairs: {len(rows)}",
          rows[:6].tolist(), flush=True)

trial(1500, 12, 8, 64, 0.1, {"GATMI_GROUPROWS": 2, "GATMI_FUSED": 0})
trial(1500, 12, 8, 64, 0.0, {"GATMI_GROUPROWS": 2, "GATMI_FUSED": 0})
trial(1500, 12, 8, 64, 0.1, {"GATMI_GROUPROWS": 0, "GATMI_FUSED": 0})
trial(1500, 12, 8, 64, 0.1, {"GATMI_GROUPROWS": 2, "GATMI_FUSED": 0, "GATMI_GEMM_V2": 0})
trial(4200, 12, 8, 64, 0.1, {"GATMI_GROUPROWS": 2, "GATMI_FUSED": 0})
trial(1500, 12, 8, 48, 0.1, {"GATMI_GROUPROWS": 2, "GATMI_FUSED": 0})
trial(1500, 12, 4, 64, 0.1, {"GATMI_FUSED": 0})
trial(1500, 16, 8, 64, 0.1, {"GATMI_GROUPROWS": 2, "GATMI_FUSED": 0})
trial(1500, 12, 8, 64, 0.1, {"GATMI_GROUPROWS": 2, "GATMI_FUSED": 0}, graph="loops")
