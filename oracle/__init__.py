"""CPU oracle (test infrastructure only; see gse_oracle.py's header)."""
