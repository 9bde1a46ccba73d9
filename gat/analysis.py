"""The two reporting helpers the training loop calls (/root/reference/gat/analysis.py:112-157) under
the reference's import path.  The exploratory plots of that file (heat map, skew / kurtosis bars)
are reporting outside the hot path and are not reproduced."""
from gat_amd.model import plot_metrics, print_epoch_stats  # noqa: F401
