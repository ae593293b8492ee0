"""ORACLE package -- test infrastructure only (see nmrfit_oracle.py header)."""
