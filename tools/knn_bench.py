"""Time the GPU kNN graph construction against scikit-learn's on the host (not part of the product)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from gat_amd.data import knn_graph_hip
from sklearn.neighbors import kneighbors_graph
for n in (10_000, 100_000, 400_000):
    x = np.random.default_rng(0).random((n, 25))
    xd = torch.as_tensor(x).cuda()
    knn_graph_hip(xd, 5); torch.cuda.synchronize()
    t0 = time.time(); ei = knn_graph_hip(xd, 5); torch.cuda.synchronize(); t_gpu = time.time() - t0
    line = f"N={n}: GPU {t_gpu*1e3:.1f} ms ({n*n*25*2/t_gpu/1e12:.1f} TFLOP/s fp64)"
    if n <= 100_000:
        t0 = time.time()
        a = kneighbors_graph(x, 5, mode="connectivity", include_self=True, n_jobs=16).tocoo()
        t_cpu = time.time() - t0
        same = np.array_equal(ei.cpu().numpy(), np.vstack([a.row, a.col]).astype(np.int64))
        line += f" | scikit-learn (16 threads) {t_cpu:.2f} s, identical: {same}"
    print(line, flush=True)
