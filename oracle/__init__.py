"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see blackstar_oracle.h).  Never imported by blackstar_amd/."""
