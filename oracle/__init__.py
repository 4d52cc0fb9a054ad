"""CPU oracle package — TEST INFRASTRUCTURE ONLY (see msi_oracle.c header)."""
