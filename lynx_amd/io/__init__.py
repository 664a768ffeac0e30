"""File formats around the hot path that need no third-party package: LatticeJSON and ASTRA."""
