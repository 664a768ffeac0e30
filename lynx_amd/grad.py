"""
Reverse-mode gradients of the outgoing beam's moments with respect to element parameters
(BASELINE config 5).  The reference only claims differentiability
(`setup.py:14-17`; its tests still assert torch `grad_fn`, `tests/test_differentiable.py`) --
there is no `jax.grad` call to mirror, so the API is an explicit vector-Jacobian product:

    vjp = lynx_amd.grad.track_vjp(segment, beam)     # forward pass, fused moments
    out = vjp.outgoing                                # the tracked ParticleBeam
    g = vjp(mu_bar=..., cov_bar=...)                  # cotangents of mean (.., 6) and cov (.., 6, 6)
    g = vjp(mu_x=..., sigma_x=..., sigma_y=...)       # or of the beam properties a loss is written in
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

# kinds whose parameter row is differentiated; the slot names come from the element's schema
DIFFERENTIABLE_KINDS = (_ffi.KIND_DRIFT, _ffi.KIND_QUADRUPOLE, _ffi.KIND_DIPOLE, _ffi.KIND_HCOR, _ffi.KIND_VCOR,
                        _ffi.KIND_CAVITY, _ffi.KIND_SOLENOID, _ffi.KIND_UNDULATOR)


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
    """
    dL/d(parameter) per element (`g[element]["k1"]`), `energy` = dL/d(incoming beam energy).
    The raw `[B][E][8]` result stays in HBM; an element's gradients are copied and shaped on
    first access.
    """

    def __init__(self, program, raw_dev, energy_dev, batch_shape, energy_shape, particles_dev=None,
                 mu_dev=None, cov_dev=None):
        self._program, self._raw_dev, self._energy_dev = program, raw_dev, energy_dev
        self._particles_dev = particles_dev
        self._mu_dev, self._cov_dev = mu_dev, cov_dev
        self._batch_shape, self._energy_shape = tuple(batch_shape), tuple(energy_shape)
        self._raw = None
        self._cache = {}

    def _host(self):
        if self._raw is None:
            E = max(len(self._program.leaves), 1)
            self._raw = self._raw_dev.numpy().reshape(*self._batch_shape, E, 8)
        return self._raw

    @property
    def energy(self) -> np.ndarray:
        return _unbroadcast(self._energy_dev.numpy().reshape(self._batch_shape), self._energy_shape)

    @property
    def particles(self):
        """dL/d(incoming particles), (*batch, N, 7), device-resident; needs `wrt_particles=True`."""
        if self._particles_dev is None:
            raise KeyError("call the VJP with wrt_particles=True to get the gradient w.r.t. the incoming particles")
        return self._particles_dev

    @property
    def mu(self) -> np.ndarray:
        """dL/d(incoming mu), (*batch, 7) -- ParameterBeam VJPs only."""
        if self._mu_dev is None:
            raise KeyError("the gradient w.r.t. mu exists for ParameterBeam VJPs only")
        return self._mu_dev.numpy()

    @property
    def cov(self) -> np.ndarray:
        """dL/d(incoming cov), (*batch, 7, 7), entry by entry -- ParameterBeam VJPs only."""
        if self._cov_dev is None:
            raise KeyError("the gradient w.r.t. cov exists for ParameterBeam VJPs only")
        return self._cov_dev.numpy()

    def __contains__(self, element) -> bool:
        return any(el is element and el._kind in DIFFERENTIABLE_KINDS for el in self._program.leaves)

    def __getitem__(self, element) -> dict:
        if id(element) in self._cache:
            return self._cache[id(element)]
        from .accelerator.magnets import RBend

        raw, total = self._host(), None
        for e, el in enumerate(self._program.leaves):
            if el is not element or el._kind not in DIFFERENTIABLE_KINDS:
                continue
            g = {n: raw[..., e, j] for j, n in enumerate(el._row_names())}
            out = {}
            if "misalignment_x" in g:
                mis = np.stack([g.pop("misalignment_x"), g.pop("misalignment_y")], axis=-1)
                out["misalignment"] = _unbroadcast(mis, np.asarray(el.misalignment).shape)
            if isinstance(el, RBend):  # e1 = e1_user + angle/2, e2 = e2_user + angle/2 (rbend.py:79-80)
                g["angle"] = g["angle"] + 0.5 * (g["e1"] + g["e2"])
            for n, v in g.items():
                out[n] = _unbroadcast(v, np.asarray(getattr(el, n)).shape)
            # the same element object may appear several times in the lattice
            total = out if total is None else {n: total[n] + out[n] for n in out}
        if total is None:
            raise KeyError(f"{element!r} has no differentiable parameters in this lattice")
        self._cache[id(element)] = total
        return total


_COORDINATES = ("x", "xp", "y", "yp", "s", "p")


def property_cotangents(outgoing: ParticleBeam, named: dict):
    """
    Cotangents of named beam properties -> (mu_bar, cov_bar).  `mu_<c>`: the mean itself;
    `sigma_<c>` = sqrt(cov_cc n / (n - ddof)) (particle_beam.py:736-823, `config.std_ddof`), so
    d sigma / d cov_cc = n / (n - ddof) / (2 sigma); `sigma_xxp`, `sigma_yyp`: cov_01, cov_23.
    """
    from . import config

    batch = outgoing.batch_shape
    mu_bar, cov_bar = np.zeros((*batch, 6)), np.zeros((*batch, 6, 6))
    n = float(outgoing.num_particles)
    for name, bar in named.items():
        bar = np.broadcast_to(np.asarray(bar, dtype=np.float64), batch)
        kind, _, coord = name.partition("_")
        if kind == "mu" and coord in _COORDINATES:
            mu_bar[..., _COORDINATES.index(coord)] += bar
        elif kind == "sigma" and coord in _COORDINATES:
            c = _COORDINATES.index(coord)
            sigma = np.asarray(getattr(outgoing, name), dtype=np.float64)
            cov_bar[..., c, c] += bar * (n / (n - config.std_ddof)) / (2.0 * sigma)
        elif name in ("sigma_xxp", "sigma_yyp"):
            c = 0 if name == "sigma_xxp" else 2
            cov_bar[..., c, c + 1] += bar
        else:
            raise KeyError(f"no cotangent rule for beam property {name!r}")
    return mu_bar, cov_bar


def _reading_cotangents(program, readings: dict, batch_shape) -> np.ndarray | None:
    """
    `{bpm: cotangent of bpm.reading}` -> [B][observers][2] float64 in the order the program reads them.  A BPM's reading
    is `stack([mu_x, mu_y])` of the beam that enters it (bpm.py:48-54), shape (2, *batch): so is its cotangent.
    """
    if not readings:
        return None
    B = int(np.prod(batch_shape, dtype=np.int64))
    out = np.zeros((B, len(program.observers), 2), dtype=np.float64)
    known = {id(element): k for k, (_, element) in enumerate(program.observers)}
    for element, bar in readings.items():
        if id(element) not in known:
            raise KeyError(f"{element!r} is not an active BPM of this lattice")
        bar = np.broadcast_to(np.asarray(bar, dtype=np.float64), (2, *batch_shape))
        out[:, known[id(element)], 0] += bar[0].reshape(B)
        out[:, known[id(element)], 1] += bar[1].reshape(B)
    return out


class TrackVJP:
    def __init__(self, segment, beam: ParticleBeam):
        if not isinstance(beam, ParticleBeam):
            raise TypeError("track_vjp needs a ParticleBeam")
        # active BPMs are read inside the streaming pass (observer steps); their readings are differentiable outputs
        items = engine.partition(segment.elements if hasattr(segment, "elements") else [segment], fuse_observers=True)
        if len(items) != 1 or not isinstance(items[0], engine.Program):
            raise NotImplementedError(
                f"track_vjp: a lattice with an active Screen or Aperture, or with more than {_ffi.MAX_OBSERVERS} active "
                "BPMs, is tracked in several passes -- differentiate the stretches one by one")
        self.program = items[0]
        beam = beam.materialized()  # the reverse pass indexes the incoming particles per sample
        self.beam = beam
        self.cache = segment.__dict__.setdefault("_lattice_cache", engine.LatticeCache())
        self.outgoing = engine.run_program_particles(self.cache, self.program, beam, moments=True)

    def __call__(self, mu_bar=None, cov_bar=None, wrt_particles: bool = False, readings: dict | None = None,
                 **properties) -> Gradients:
        """
        Cotangents of the outgoing beam's mean (`mu_bar`, (*batch, 6)) and covariance (`cov_bar`, (*batch, 6, 6)), of named
        beam properties (`sigma_x=...`), and -- `readings={bpm: bar}` -- of the readings of the lattice's active BPMs
        (`bar` shaped like `bpm.reading`: (2, *batch)).
        """
        rt = get_runtime()
        beam, program = self.beam, self.program
        batch_shape, dtype = beam.batch_shape, beam.dtype
        B = int(np.prod(batch_shape, dtype=np.int64))
        if properties:
            extra_mu, extra_cov = property_cotangents(self.outgoing, properties)
            mu_bar = extra_mu if mu_bar is None else extra_mu + np.asarray(mu_bar, dtype=np.float64).reshape(extra_mu.shape)
            cov_bar = extra_cov if cov_bar is None else extra_cov + np.asarray(cov_bar, dtype=np.float64).reshape(extra_cov.shape)
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
        lat = engine._ready(self.cache, program, batch_shape, dtype, beam._energy._host)
        E = lat.E
        # (cotangents in and gradients out: small ones live in host memory the GPU reads and writes through)
        g_rec = rt.to_device_result(rec)
        g_par = rt.empty_result((B, max(E, 1), 8), dtype)
        g_en = rt.empty_result((B,), dtype)
        g_p = rt.empty((*batch_shape, beam.num_particles, 7), dtype) if wrt_particles else None
        fwd = self.outgoing._moments.device(rt)
        e_in = beam._energy.broadcast_device(rt, batch_shape)
        obs_bar = _reading_cotangents(program, readings, batch_shape)
        g_obs = None if obs_bar is None else rt.to_device_result(obs_bar)  # (named: alive until the call has been enqueued)
        rt.check(rt.lib.lynx_track_particles_backward(
            rt.ctx, lat.handle, beam.num_particles, C.c_void_p(e_in.ptr), C.c_void_p(beam._particles.device(rt).ptr),
            C.c_void_p(fwd.ptr), C.c_void_p(g_rec.ptr), C.c_void_p(g_par.ptr), C.c_void_p(g_en.ptr),
            None if g_p is None else C.c_void_p(g_p.ptr), None if g_obs is None else C.c_void_p(g_obs.ptr)))
        return Gradients(program, g_par, g_en, batch_shape, np.asarray(beam.energy).shape, g_p)


class ChainedGradients:
    """Gradients of a lattice that was differentiated stretch by stretch (a ParameterBeam through active BPMs)."""

    def __init__(self, parts: list):
        self._parts = parts  # in lattice order

    @property
    def energy(self) -> np.ndarray:
        return sum(part.energy for part in self._parts)

    @property
    def mu(self) -> np.ndarray:
        return self._parts[0].mu

    @property
    def cov(self) -> np.ndarray:
        return self._parts[0].cov

    def __contains__(self, element) -> bool:
        return any(element in part for part in self._parts)

    def __getitem__(self, element) -> dict:
        total = None
        for part in self._parts:
            if element in part:
                got = part[element]
                total = got if total is None else {n: total[n] + got[n] for n in got}
        if total is None:
            raise KeyError(f"{element!r} has no differentiable parameters in this lattice")
        return total


class MomentsVJP:
    """
    Vector-Jacobian product of `Segment.track` on a `ParameterBeam` (lynx_track_moments_backward).  An active BPM cuts
    the lattice in stretches (it is a host-side step of `track`, bpm.py:48-58): each stretch is differentiated by the
    kernel, and (mu_bar, cov_bar) -- plus the cotangent of the BPM's reading, which is (mu_x, mu_y) of the beam at that
    point -- is handed from a stretch to the one in front of it.
    """

    def __init__(self, segment, beam):
        items = engine.partition(segment.elements if hasattr(segment, "elements") else [segment])
        for item in items:
            if not isinstance(item, engine.Program) and not getattr(item, "_fusable_observer", False):
                raise NotImplementedError("track_vjp: lattices with an active Screen or Aperture are not differentiated")
        self.cache = segment.__dict__.setdefault("_lattice_cache", engine.LatticeCache())
        self.stretches = []  # (Program, beam entering it) / (BPM, None), in lattice order
        for k, item in enumerate(items):
            if isinstance(item, engine.Program):
                if any(el._kind == _ffi.KIND_CAVITY for el in item.leaves) and any(isinstance(i, engine.Program) for i in items[k + 1:]):
                    raise NotImplementedError(
                        "track_vjp: a cavity in front of an active BPM (the energy cotangent does not cross stretches yet)")
                self.stretches.append((item, beam))
                beam = engine.run_program_parameters(self.cache, item, beam)
            else:
                item._observe(beam)
                self.stretches.append((item, None))
                beam = beam._shallow_copy()
        self.program = items[0] if len(items) == 1 else None
        self.beam = self.stretches[0][1] if self.stretches else beam
        self.outgoing = beam

    def _property_cotangents(self, named: dict):
        """`mu_<c>`: mu[c]; `sigma_<c>` = sqrt(max(cov_cc, 1e-20)) (parameter_beam.py:371-417); `sigma_xxp/yyp`: cov_01, cov_23."""
        out = self.outgoing
        batch = out.batch_shape
        mu_bar, cov_bar = np.zeros((*batch, 7)), np.zeros((*batch, 7, 7))
        for name, bar in named.items():
            bar = np.broadcast_to(np.asarray(bar, dtype=np.float64), batch)
            kind, _, coord = name.partition("_")
            if kind == "mu" and coord in _COORDINATES:
                mu_bar[..., _COORDINATES.index(coord)] += bar
            elif kind == "sigma" and coord in _COORDINATES:
                c = _COORDINATES.index(coord)
                var = np.asarray(out._cov, dtype=np.float64)[..., c, c]
                cov_bar[..., c, c] += np.where(var > 1e-20, bar / (2.0 * np.sqrt(np.maximum(var, 1e-20))), 0.0)
            elif name in ("sigma_xxp", "sigma_yyp"):
                c = 0 if name == "sigma_xxp" else 2
                cov_bar[..., c, c + 1] += bar
            else:
                raise KeyError(f"no cotangent rule for beam property {name!r}")
        return mu_bar, cov_bar

    def _stretch(self, program, beam, mb, cb) -> Gradients:
        rt = get_runtime()
        batch_shape, dtype = beam.batch_shape, beam.dtype
        B = int(np.prod(batch_shape, dtype=np.int64))
        lat = engine._ready(self.cache, program, batch_shape, dtype, beam._energy._host)
        g_par = rt.empty((B, max(lat.E, 1), 8), dtype)
        g_en = rt.empty((B,), dtype)
        g_mu = rt.empty((*batch_shape, 7), dtype)
        g_cov = rt.empty((*batch_shape, 7, 7), dtype)
        e_in = beam._energy.broadcast_device(rt, batch_shape)
        p = lambda a: C.c_void_p(a.ptr)  # noqa: E731
        # named, so that both uploads stay allocated until the call has been enqueued
        mb_dev, cb_dev = rt.to_device(mb.astype(dtype)), rt.to_device(cb.astype(dtype))
        rt.check(rt.lib.lynx_track_moments_backward(
            rt.ctx, lat.handle, p(e_in), p(beam._mu_d.device(rt)), p(beam._cov_d.device(rt)),
            p(mb_dev), p(cb_dev), p(g_par), p(g_en), p(g_mu), p(g_cov)))
        return Gradients(program, g_par, g_en, batch_shape, np.asarray(beam.energy).shape, mu_dev=g_mu, cov_dev=g_cov)

    def __call__(self, mu_bar=None, cov_bar=None, readings: dict | None = None, **properties):
        """
        `mu_bar` (*batch, <=7), `cov_bar` (*batch, 6|7, 6|7): cotangents entry by entry; `readings={bpm: bar}`: cotangents
        of the readings of active BPMs, shaped like `bpm.reading` -- (2, *batch).
        """
        batch_shape = self.outgoing.batch_shape
        B = int(np.prod(batch_shape, dtype=np.int64))
        mb, cb = np.zeros((B, 7)), np.zeros((B, 7, 7))
        if properties:
            pm, pc = self._property_cotangents(properties)
            mb += pm.reshape(B, 7)
            cb += pc.reshape(B, 7, 7)
        if mu_bar is not None:
            mu_bar = np.asarray(mu_bar, dtype=np.float64).reshape(B, -1)
            mb[:, : mu_bar.shape[1]] += mu_bar
        if cov_bar is not None:
            cov_bar = np.asarray(cov_bar, dtype=np.float64)
            k = cov_bar.shape[-1]
            cb[:, :k, :k] += cov_bar.reshape(B, k, k)
        readings = dict(readings or {})
        known = {id(item): item for item, beam in self.stretches if beam is None}
        for element in readings:
            if id(element) not in known:
                raise KeyError(f"{element!r} is not an active BPM of this lattice")
        parts = []
        for item, beam in reversed(self.stretches):
            if beam is None:  # an active BPM: its reading is (mu_x, mu_y) of the beam passing here
                for element, bar in readings.items():
                    if element is item:
                        bar = np.broadcast_to(np.asarray(bar, dtype=np.float64), (2, *batch_shape))
                        mb[:, 0] += bar[0].reshape(B)
                        mb[:, 2] += bar[1].reshape(B)
                continue
            part = self._stretch(item, beam, mb, cb)
            parts.insert(0, part)
            mb, cb = np.asarray(part.mu, dtype=np.float64).reshape(B, 7), np.asarray(part.cov, dtype=np.float64).reshape(B, 7, 7)
        if not parts:
            raise ValueError("track_vjp: the lattice has no element to differentiate")
        return parts[0] if len(parts) == 1 else ChainedGradients(parts)


def track_vjp(segment, beam):
    """Forward pass through `segment`; returns the callable vector-Jacobian product."""
    from .particles.parameter_beam import ParameterBeam

    if isinstance(beam, ParameterBeam):
        return MomentsVJP(segment, beam)
    return TrackVJP(segment, beam)
