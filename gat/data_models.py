from gat_amd.data import Metrics  # noqa: F401
