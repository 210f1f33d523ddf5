"""CPU oracle (test infrastructure only). See oracle/vorta_oracle.py."""
