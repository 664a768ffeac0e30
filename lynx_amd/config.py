"""Run-time switches of the package (plain module attributes)."""

import os

# Accumulate the outgoing beam's moments in the epilogue of the tracking kernel (no extra
# HBM traffic).  If False, the first moment property read costs one more pass over the beam.
fused_moments = os.environ.get("LYNX_FUSED_MOMENTS", "1") != "0"

# What the fused epilogue accumulates: by default the moments the reference's ParticleBeam exposes as
# properties (means, the six variances, sigma_xx', sigma_yy'); True adds the rest of the 6x6
# covariance (21 products per particle instead of 8).  `beam.covariance()` gets the whole matrix of
# any beam with one extra pass when it was not accumulated.
fused_covariance = os.environ.get("LYNX_FUSED_COVARIANCE", "0") == "1"

# Build+compose in its own launch instead of the fused prologue (A/B switch).
two_kernel = os.environ.get("LYNX_TWO_KERNEL", "0") == "1"

# Delta degrees of freedom of ParticleBeam.sigma_*.  The reference spells it
# `xs.std(dim=-1)` (lynx/particles/particle_beam.py:742), i.e. the torch default: unbiased.
std_ddof = int(os.environ.get("LYNX_STD_DDOF", "1"))

# `ParticleBeam.broadcast` of a single beam: keep one stored copy and let the tracking kernel
# read it once per sample (LYNX_TRACK_SHARED_INPUT) instead of repeating it physically like
# the reference does (particle_beam.py:838-843).  Same values and shapes either way.
lazy_broadcast = os.environ.get("LYNX_LAZY_BROADCAST", "1") != "0"

# A run that is followed by an active cavity is applied together with it (one 7x7 application with
# T_cav . T_run per particle instead of two; LYNX_TRACK_SEQUENTIAL_STEPS when False).
merge_steps = os.environ.get("LYNX_MERGE_STEPS", "1") != "0"
