"""CPU oracle (test infrastructure).  See gat_oracle.py for the import rules."""
