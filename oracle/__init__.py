"""CPU oracle (test infrastructure only).  See oracle/shotvae_oracle.py."""
