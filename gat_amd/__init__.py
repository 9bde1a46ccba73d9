"""gat_amd -- the GAT layer hot path of gokcantali/gat on MI355X (gfx950).

Host side (PyTorch-ROCm for memory, streams and torch.distributed) over the C ABI of
libgat_mi355.so (include/gat_mi355.h): hand-written HIP kernels only, no fallback.
"""
from .layer import GATConv  # noqa: F401
from .graph import GraphLayout, build_graph, cached_graph  # noqa: F401
