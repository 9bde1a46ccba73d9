from gat_amd.model import GAT  # noqa: F401
