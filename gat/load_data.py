"""The reference's CSV loader under its own import path (/root/reference/gat/load_data.py:8):
`load_data(path)` is gat_amd.etl.read_traces."""
from gat_amd.etl import read_traces as load_data  # noqa: F401
