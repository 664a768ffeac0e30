"""
Reverse-mode gradients of the outgoing beam's moments with respect to element parameters
(BASELINE config 5).  The reference only claims differentiability
(`setup.py:14-17`; its tests still assert torch `grad_fn`, `tests/test_differentiable.py`) --
there is no `jax.grad` call to mirror, so the API is an explicit vector-Jacobian product:

    vjp = lynx_amd.grad.track_vjp(segment, beam)     # forward pass, fused moments
    out = vjp.outgoing                                # the tracked ParticleBeam
    g = vjp(mu_bar=..., cov_bar=...)                  # cotangents of mean (.., 6) and cov (.., 6, 6)
    g[segment.Q1]["k1"], g.energy                     # dL/dk1 (shape of k1), dL/dE_in

All arithmetic runs in the HIP kernels of `lynx_amd/csrc/lynx_grad.hpp`
(`lynx_track_particles_backward`).
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi, engine
from .device import get_runtime
from .particles.particle_beam import ParticleBeam, _tri

PARAM_NAMES = {
    _ffi.KIND_DRIFT: ["length"],
    _ffi.KIND_QUADRUPOLE: ["length", "k1", "tilt", "misalignment_x", "misalignment_y"],
    _ffi.KIND_DIPOLE: ["length", "angle", "e1", "e2", "tilt", "fringe_integral", "fringe_integral_exit", "gap"],
    _ffi.KIND_HCOR: ["length", "angle"],
    _ffi.KIND_VCOR: ["length", "angle"],
    _ffi.KIND_CAVITY: ["length", "voltage", "phase", "frequency"],
}


def _unbroadcast(grad: np.ndarray, shape) -> np.ndarray:
    """Sum a batch-shaped gradient down to the shape of a parameter that was broadcast."""
    shape = tuple(shape)
    if grad.shape == shape:
        return grad
    while grad.ndim > len(shape):
        grad = grad.sum(axis=0)
    for ax, n in enumerate(shape):
        if n == 1 and grad.shape[ax] != 1:
            grad = grad.sum(axis=ax, keepdims=True)
    return grad.reshape(shape)


class Gradients:
    """dL/d(parameter) per element, `energy` = dL/d(incoming beam energy)."""

    def __init__(self, per_element: dict, energy: np.ndarray):
        self._per_element = per_element
        self.energy = energy

    def __getitem__(self, element) -> dict:
        return self._per_element[id(element)]

    def __contains__(self, element) -> bool:
        return id(element) in self._per_element


class TrackVJP:
    def __init__(self, segment, beam: ParticleBeam):
        if not isinstance(beam, ParticleBeam):
            raise TypeError("track_vjp needs a ParticleBeam")
        items = engine.partition(segment.elements if hasattr(segment, "elements") else [segment])
        if len(items) != 1 or not isinstance(items[0], engine.Program):
            raise NotImplementedError("track_vjp: lattices with active BPMs are not supported yet")
        self.program = items[0]
        self.beam = beam
        self.cache = segment.__dict__.setdefault("_lattice_cache", engine.LatticeCache())
        self.outgoing = engine.run_program_particles(self.cache, self.program, beam, moments=True)

    def __call__(self, mu_bar=None, cov_bar=None) -> Gradients:
        rt = get_runtime()
        beam, program = self.beam, self.program
        batch_shape, dtype = beam.batch_shape, beam.dtype
        B = int(np.prod(batch_shape, dtype=np.int64))
        rec = np.zeros((B, _ffi.MOMENT_STRIDE), dtype=np.float64)
        if mu_bar is not None:
            mu_bar = np.asarray(mu_bar, dtype=np.float64).reshape(B, -1)
            rec[:, : mu_bar.shape[1]] = mu_bar
        if cov_bar is not None:
            G = np.asarray(cov_bar, dtype=np.float64).reshape(B, 6, 6)
            for i in range(6):
                rec[:, _tri(i, i)] = G[:, i, i]
                for j in range(i + 1, 6):
                    rec[:, _tri(i, j)] = G[:, i, j] + G[:, j, i]
        lat = engine._ready(self.cache, program, batch_shape, dtype, beam._energy.host())
        E = lat.E
        g_rec = rt.to_device(rec)
        g_par = rt.empty((B, max(E, 1), 8), dtype)
        g_en = rt.empty((B,), dtype)
        fwd = self.outgoing._moments.device(rt)
        e_in = beam._energy.broadcast_device(rt, batch_shape)
        rt.check(rt.lib.lynx_track_particles_backward(
            rt.ctx, lat.handle, beam.num_particles, C.c_void_p(e_in.ptr), C.c_void_p(beam._particles.device(rt).ptr),
            C.c_void_p(fwd.ptr), C.c_void_p(g_rec.ptr), C.c_void_p(g_par.ptr), C.c_void_p(g_en.ptr)))
        raw = g_par.numpy().reshape(*batch_shape, max(E, 1), 8)
        per_element = {}
        for e, el in enumerate(program.leaves):
            names = PARAM_NAMES.get(el._kind)
            if not names:
                continue
            g = {n: raw[..., e, j] for j, n in enumerate(names)}
            out = {}
            if "misalignment_x" in g:
                mis = np.stack([g.pop("misalignment_x"), g.pop("misalignment_y")], axis=-1)
                out["misalignment"] = _unbroadcast(mis, np.asarray(el.misalignment).shape)
            from .accelerator.dipole import RBend

            if isinstance(el, RBend):  # e1 = e1_user + angle/2, e2 = e2_user + angle/2 (rbend.py:79-80)
                g["angle"] = g["angle"] + 0.5 * (g["e1"] + g["e2"])
            for n, v in g.items():
                out[n] = _unbroadcast(v, np.asarray(getattr(el, n)).shape)
            if id(el) in per_element:  # the same element object appears twice in the lattice
                for n in out:
                    per_element[id(el)][n] = per_element[id(el)][n] + out[n]
            else:
                per_element[id(el)] = out
        energy = _unbroadcast(g_en.numpy().reshape(batch_shape), np.asarray(beam.energy).shape)
        return Gradients(per_element, energy)


def track_vjp(segment, beam: ParticleBeam) -> TrackVJP:
    """Forward pass through `segment`; returns the callable vector-Jacobian product."""
    return TrackVJP(segment, beam)
