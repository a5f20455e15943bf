"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see oracle/vo_oracle.h)."""
