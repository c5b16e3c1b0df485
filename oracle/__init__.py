"""CPU oracle for the doppel-speller hot path -- TEST INFRASTRUCTURE ONLY (see oracle/doppel_oracle.c)."""
