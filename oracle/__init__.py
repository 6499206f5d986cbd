"""CPU restatement of the LIDF query path: test infrastructure only (see lidf_oracle.py)."""
