"""Import-path shim: `from gat.model import GAT`, `from gat.converter import
convert_to_graph`, `from gat.data_models import Metrics` resolve to gat_amd, so the
reference's drivers keep their imports (/root/reference/run.py:7-9)."""
