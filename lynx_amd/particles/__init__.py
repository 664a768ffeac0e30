"""Beam containers: moments-only (`ParameterBeam`) and macro-particles in HBM (`ParticleBeam`)."""

from .beam import Beam
from .parameter_beam import ParameterBeam
from .particle_beam import ParticleBeam

__all__ = ["Beam", "ParameterBeam", "ParticleBeam"]
