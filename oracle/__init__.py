"""Test infrastructure only: CPU restatement of the reference hot path (see shineon_oracle.py)."""
