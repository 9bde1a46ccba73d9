"""The reference's column encoders under its own import path (/root/reference/gat/encoder.py:7-58),
from gat_amd/etl.py (same names, same arguments, same results)."""
from gat_amd.etl import boolean_string_to_int, ip_encoder, number_normalizer, string_encoder  # noqa: F401
