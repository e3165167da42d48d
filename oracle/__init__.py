"""CPU oracle (test infrastructure only) -- see oracle/coreg_oracle.py."""
