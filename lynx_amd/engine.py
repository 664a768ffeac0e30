"""
Host side of the hot path: turns a list of element objects into the flat lattice program
liblynxhip consumes (include/lynx_hip.h) and dispatches the kernels.

What stays on the host is exactly what the reference also decides in Python: the order of
elements, the partition of the lattice into maximal skippable runs and non-skippable
elements (`Segment.track`, lynx/accelerator/segment.py:340-356) and the whole-batch
`if any(...)` predicates (track_methods.py:101, quadrupole.py:75, dipole.py:119,
cavity.py:128,164,260,290).  All arithmetic on maps, moments and particles happens in the
HIP kernels; there is no CPU fallback.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _ffi, config
from .device import DeviceArray, Dual, dtype_code, get_runtime

ELECTRON_MASS_EV = 510998.95069  # cavity.py:20
_F64 = np.dtype(np.float64)

_LATE = None


def _late():
    """
    What the engine needs from modules that import the engine themselves, bound on first use.  (As `from ... import`
    statements inside `track` and friends they cost seven import-system look-ups per call -- a fifth of the host
    time of a ParameterBeam `track`, which is bound by the host's enqueue rate.)
    """
    global _LATE
    if _LATE is None:
        from types import SimpleNamespace

        from .accelerator.element import EPOCH, STRUCTURE
        from .accelerator.segment import Segment
        from .particles.beam import Beam
        from .particles.parameter_beam import ParameterBeam
        from .particles.particle_beam import ParticleBeam

        from .accelerator.segment import ElementList

        _LATE = SimpleNamespace(EPOCH=EPOCH, STRUCTURE=STRUCTURE, Segment=Segment, Beam=Beam,
                                ParameterBeam=ParameterBeam, ParticleBeam=ParticleBeam, ElementList=ElementList)
    return _LATE


# -------------------------------------------------------------------------------------------
# program = what Segment.track iterates over
# -------------------------------------------------------------------------------------------


@dataclass

class Program:
    """A maximal stretch of the lattice one kernel launch can run: runs + active cavities."""

    leaves: list = field(default_factory=list)  # leaf elements in lattice order
    steps: list = field(default_factory=list)   # (kind, first, last) over `leaves`
    raw: bool = False  # runs take their first map as is (Element.track), see LYNX_STEP_FLAG_RAW
    observers: list = field(default_factory=list)  # (step index, element): active BPMs read inside the pass

    def add_run_element(self, element, new_run: bool):
        idx = len(self.leaves)
        self.leaves.append(element)
        if new_run or not self.steps or self.steps[-1][0] != _ffi.STEP_RUN:
            self.steps.append([_ffi.STEP_RUN, idx, idx + 1])
        else:
            self.steps[-1][2] = idx + 1

    def add_cavity(self, element):
        idx = len(self.leaves)
        self.leaves.append(element)
        self.steps.append([_ffi.STEP_CAVITY, idx, idx + 1])

    def add_observer(self, element):
        """An active BPM as a one-element run step of its own (LYNX_STEP_FLAG_OBSERVE): the streaming kernel
        adds up x and y of the particles entering it -- `reading = stack([mu_x, mu_y])`, bpm.py:48-54 --
        instead of the pass being cut in two there."""
        idx = len(self.leaves)
        self.leaves.append(element)
        self.observers.append((len(self.steps), element))
        self.steps.append([_ffi.STEP_RUN, idx, idx + 1])


def partition(elements, fuse_observers: bool = False) -> list:
    """
    Mirror of the `todos` loop in segment.py:344-351, flattened over nested segments:
    returns a list of `Program` objects and host-side barrier elements (active screens and
    apertures; active BPMs too unless `fuse_observers`, the ParticleBeam case, where up to
    `_ffi.MAX_OBSERVERS` of them per program are read inside the streaming pass), in
    order.  A nested non-skippable Segment is tracked on its own by the reference, so it
    starts and ends a run; a nested skippable Segment's elements join the current run (the
    reference multiplies its pre-composed product instead: same map up to rounding).
    """
    Segment = _late().Segment

    out: list = []
    state = {"new_run": True}

    def current() -> Program:
        if not out or not isinstance(out[-1], Program):
            out.append(Program())
            state["new_run"] = True
        return out[-1]

    def room_for_observer() -> bool:
        """The program being filled reads fewer than MAX_OBSERVERS active BPMs so far (a new one reads none)."""
        return not (out and isinstance(out[-1], Program)) or len(out[-1].observers) < _ffi.MAX_OBSERVERS

    def walk(items):
        for el in items:
            if isinstance(el, Segment):
                if el.is_skippable:
                    walk(el.elements)
                else:
                    state["new_run"] = True
                    walk(el.elements)
                    state["new_run"] = True
            elif el.is_skippable:
                current().add_run_element(el, state["new_run"])
                state["new_run"] = False
            elif el._kind == _ffi.KIND_CAVITY:
                current().add_cavity(el)
                state["new_run"] = True
            elif fuse_observers and getattr(el, "_fusable_observer", False) and room_for_observer():
                current().add_observer(el)
                state["new_run"] = True
            elif getattr(el, "_host_barrier", False):
                out.append(el)
                state["new_run"] = True
            else:
                raise TypeError(f"element {el!r} cannot be tracked by lynx_amd")

    walk(elements)
    return out


def plan(owner, elements, raw: bool, fuse_observers: bool = False) -> list:
    """
    `partition(elements)` remembered on `owner` for as long as no attribute that can change the
    partition has been written (global STRUCTURE counter) and the element list holds the same
    objects: a tracking loop then costs one C-level pass over the list instead of Python work
    per element, also when magnet strengths are rewritten between the calls.
    """
    late = _late()
    STRUCTURE, Segment = late.STRUCTURE, late.Segment

    def identities(items):
        ids = tuple(map(id, items))
        if any(issubclass(t, Segment) for t in set(map(type, items))):  # nested lists can change too
            ids = (ids, tuple(identities(el.elements) for el in items if isinstance(el, Segment)))
        return ids

    # a Segment's own list says when it is changed in place (ElementList bumps STRUCTURE, and so do the lists of nested
    # segments): the list object itself stands for its contents; any other sequence is compared element by element
    ids = id(elements) if type(elements) is late.ElementList else identities(elements)
    token = (STRUCTURE[0], raw, ids)
    remembered = owner.__dict__.setdefault("_plan", {}).get(fuse_observers)
    if remembered is not None and remembered[0] == token:
        return remembered[1]
    items = partition(elements, fuse_observers)
    for item in items:
        if isinstance(item, Program):
            item.raw = raw
    owner.__dict__["_plan"][fuse_observers] = (token, items)
    return items


# -------------------------------------------------------------------------------------------
# packing
# -------------------------------------------------------------------------------------------


def _broadcast_param(value, batch_shape, dtype, what):
    arr = np.asarray(value, dtype=dtype)
    try:
        return np.broadcast_to(arr, batch_shape)
    except ValueError:
        raise AssertionError(
            f"Beam shape {tuple(batch_shape)} does not match element shape {arr.shape} ({what})"
        ) from None


class PackedLattice:
    """Device-resident lattice program + the host arrays it was packed from."""

    def __init__(self, program: Program, batch_shape, dtype):
        self.program = program
        self.batch_shape = tuple(batch_shape)
        self.dtype = np.dtype(dtype)
        self.B = int(np.prod(self.batch_shape, dtype=np.int64))
        leaves = program.leaves
        E, S = len(leaves), len(program.steps)
        self.elems = (_ffi.Elem * max(E, 1))()
        self.steps = (_ffi.Step * max(S, 1))()
        parts, offset = [], 0
        self.layout = []  # per element: (offset, scalars, batch stride) in the parameter pool
        for e, el in enumerate(leaves):
            block, stride = self._pack_element(el)
            self.elems[e] = _ffi.Elem(el._kind, 0, offset, stride)
            self.layout.append((offset, block.size, stride))
            if block.size:
                parts.append(block)
                offset += block.size
        self.pool = (np.concatenate(parts) if parts else np.zeros(1, dtype=self.dtype)).astype(self.dtype)
        for s, (kind, first, last) in enumerate(program.steps):
            self.steps[s] = _ffi.Step(kind, first, last, 0)
        self.E, self.S = E, S
        self.elem_flags = [0] * E
        self.step_flags = [0] * S
        self.has_cavity_step = any(k == _ffi.STEP_CAVITY for k, _, _ in program.steps)
        self.handle = None
        self.rt = None
        self._static = None
        self._has_cavity = False
        self._ready_epoch = None
        self.versions = tuple(el._version for el in leaves)

    def _pack_element(self, el):
        """The element's parameter rows as one pool block: (flat array, batch stride)."""
        rows = el._param_rows(self.dtype)
        n = _ffi.PARAMS_OF_KIND[el._kind]
        assert len(rows) == n, (el, len(rows), n)
        if n == 0:
            return np.zeros(0, dtype=self.dtype), 0
        if all(np.asarray(r).size == 1 for r in rows):  # shared by the whole batch
            return np.array([np.asarray(r, dtype=self.dtype).reshape(()) for r in rows], dtype=self.dtype), 0
        mat = np.stack([_broadcast_param(r, self.batch_shape, self.dtype, type(el).__name__).reshape(self.B)
                        for r in rows], axis=1)
        return np.ascontiguousarray(mat).reshape(-1), n

    def refresh(self, versions) -> bool:
        """
        Element parameters changed (`quad.k1 = ...`) but not the structure: rewrite the pool
        blocks of the changed elements in place (`lynx_lattice_update_params`).  False if a
        block changed its size or stride -- the caller then packs from scratch.
        """
        for e, (el, old, new) in enumerate(zip(self.program.leaves, self.versions, versions)):
            if old == new:
                continue
            block, stride = self._pack_element(el)
            offset, size, old_stride = self.layout[e]
            if block.size != size or stride != old_stride:
                return False
            if size:
                self.pool[offset:offset + size] = block
                if self.handle is not None:
                    self.rt.check(self.rt.lib.lynx_lattice_update_params(
                        self.handle, offset, size, self.pool[offset:offset + size].ctypes.data))
            if self._static is not None:  # whole-batch predicates of this element (tilt, misalignment, ...)
                self._static = list(self._static)
                self._static[e] = el._static_flags()
        self.versions = tuple(versions)
        return True

    # whole-batch predicates -----------------------------------------------------------------
    def evaluate_flags(self, energy_host=None):
        """
        Element and step flags the host decides: the parameter-only predicates of magnets (tilt,
        misalignment, thick dipole: `_static_flags`), `LYNX_STEP_FLAG_RAW` and the observer steps.  The
        cavity predicates depend on the beam energy on its way through the lattice and are evaluated on
        the device before every build (`k_cavity_flags`), so the energy never has to come back to the
        host for them.  What stays here of cavity.py:260 (`assert Ei > 0`) is the check of the INCOMING
        energy, when the host already holds it; a cavity that decelerates the beam through zero is not
        caught.  Returns (elem_flags, step_flags).
        """
        leaves = self.program.leaves
        if self._static is None:  # element versions are part of the cache key: these cannot change
            self._static = [el._static_flags() for el in leaves]
            self._has_cavity = any(el._kind == _ffi.KIND_CAVITY for el in leaves)
        base = _ffi.STEP_FLAG_RAW if self.program.raw else 0
        observed = {s for s, _ in self.program.observers}
        if self._has_cavity and energy_host is not None:
            assert np.all(np.asarray(energy_host) > 0), "Initial energy must be larger than 0"
        return self._static, [base | (_ffi.STEP_FLAG_OBSERVE if s in observed else 0) for s in range(self.S)]

    # device side ----------------------------------------------------------------------------
    def upload(self, rt, elem_flags, step_flags):
        for e, f in enumerate(elem_flags):
            self.elems[e].flags = f
        for s, f in enumerate(step_flags):
            self.steps[s].flags = f
        self.elem_flags, self.step_flags = list(elem_flags), list(step_flags)
        handle = C.c_void_p()
        rt.check(rt.lib.lynx_lattice_create(
            rt.ctx, dtype_code(self.dtype), self.B, self.E, self.elems, self.S, self.steps,
            self.pool.ctypes.data, self.pool.size, C.byref(handle)))
        self.handle, self.rt = handle, rt

    def set_flags(self, elem_flags, step_flags):
        if list(elem_flags) == self.elem_flags and list(step_flags) == self.step_flags:
            return
        ef = (C.c_int32 * max(self.E, 1))(*elem_flags)
        sf = (C.c_int32 * max(self.S, 1))(*step_flags)
        self.rt.check(self.rt.lib.lynx_lattice_set_flags(self.handle, ef, sf))
        self.elem_flags, self.step_flags = list(elem_flags), list(step_flags)

    def release(self):
        if self.handle is not None:
            if not self.rt.closed:
                self.rt.lib.lynx_lattice_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class LatticeCache:
    """
    Packed programs of one owner (a Segment or an Element).  Keyed by structure (which element
    objects, which steps, batch shape, dtype); a parameter change only rewrites the changed
    elements' blocks of the device-resident pool.
    """

    def __init__(self, capacity: int = 4):
        self.capacity = capacity
        self.entries: dict = {}
        self._last = None

    def get(self, program: Program, batch_shape, dtype) -> PackedLattice:
        EPOCH = _late().EPOCH
        shape_key = (tuple(batch_shape), np.dtype(dtype).str, program.raw)
        last = self._last
        if last is not None and last[0] is program and last[1] == EPOCH[0] and last[2] == shape_key:
            return last[3]
        leaves = program.leaves
        key = (tuple(map(id, leaves)), tuple(tuple(s) for s in program.steps), shape_key)
        versions = tuple(el._version for el in leaves)
        hit = self.entries.get(key)
        if hit is not None and hit.versions != versions and not hit.refresh(versions):
            self.entries.pop(key).release()
            hit = None
        if hit is None:
            if len(self.entries) >= self.capacity:
                self.entries.pop(next(iter(self.entries))).release()
            hit = PackedLattice(program, batch_shape, dtype)
            self.entries[key] = hit
        self._last = (program, EPOCH[0], shape_key, hit)
        return hit


def _ready(cache: LatticeCache, program: Program, batch_shape, dtype, energy_host) -> PackedLattice:
    lat = cache.get(program, batch_shape, dtype)
    # nothing an element holds has been written since this lattice's flags were last evaluated (EPOCH), and it has no
    # cavity whose `assert Ei > 0` would have to look at THIS beam's energy: it is ready as it stands
    epoch = _late().EPOCH[0]
    if lat._ready_epoch == epoch and not lat._has_cavity and lat.handle is not None:
        return lat
    elem_flags, step_flags = lat.evaluate_flags(energy_host)
    if lat.handle is None:
        lat.upload(get_runtime(), elem_flags, step_flags)
    else:
        lat.set_flags(elem_flags, step_flags)
    lat._ready_epoch = epoch
    return lat


# -------------------------------------------------------------------------------------------
# dispatch
# -------------------------------------------------------------------------------------------


def _ptr(x):
    return None if x is None else C.c_void_p(x.ptr)


def run_program_particles(cache, program: Program, beam, moments: bool | None = None):
    """One pass of the streaming kernel: ParticleBeam -> ParticleBeam (`lynx_track_particles_new`)."""
    ParticleBeam = _late().ParticleBeam
    rt = get_runtime()
    dtype = beam.dtype
    batch_shape = beam.batch_shape
    lat = _ready(cache, program, batch_shape, dtype, beam._energy._host)  # no read-back: None if it lives in HBM only
    p_in = beam._particles.device(rt)  # (N, 7)-like storage when the beam is shared by the batch
    e_in = beam._energy.broadcast_device(rt, batch_shape)
    want_moments = config.fused_moments if moments is None else moments
    flags = ((0 if not want_moments else _ffi.TRACK_COVARIANCE if config.fused_covariance else _ffi.TRACK_MOMENTS)
             | (_ffi.TRACK_TWO_KERNEL if config.two_kernel else 0)
             | (_ffi.TRACK_SHARED_INPUT if beam.is_shared else 0)
             | (0 if config.merge_steps else _ffi.TRACK_SEQUENTIAL_STEPS))
    n = beam.num_particles
    # the outgoing beam's blocks come from the library's pool inside the call: one crossing of the C boundary per
    # `track` instead of one per array (a small call is bound by the host's enqueue rate)
    blocks = (C.c_void_p * 4)()
    status = rt.lib.lynx_track_particles_new(rt.ctx, lat.handle, n, e_in.ptr, p_in.ptr, flags,
                                             1 if lat.has_cavity_step else 0, blocks)
    if status:
        rt.check(status)
    adopt = DeviceArray.adopt
    p_out = adopt(rt, blocks[0], (*batch_shape, n, 7), dtype)
    e_out = adopt(rt, blocks[1], batch_shape, dtype) if blocks[1] else None
    mom = adopt(rt, blocks[2], (*batch_shape, _ffi.MOMENT_STRIDE), _F64) if blocks[2] else None
    if blocks[3]:
        obs = adopt(rt, blocks[3], (lat.B, len(program.observers), 2), _F64)
        for k, (_, element) in enumerate(program.observers):
            element._reading_from(obs, k, batch_shape, dtype)  # read back only if somebody looks at it
    out = ParticleBeam.__new__(ParticleBeam)
    charges = beam._charges
    if charges is not None and beam.is_shared:
        charges = np.ascontiguousarray(beam.particle_charges)
    out._init_raw(Dual(dev=p_out), Dual(dev=e_out) if e_out is not None else beam._energy,
                  charges, dtype, moments=Dual(dev=mom) if mom is not None else None)
    return out


def run_program_parameters(cache, program: Program, beam):
    """ParameterBeam -> ParameterBeam (lynx_track_moments)."""
    ParameterBeam = _late().ParameterBeam
    rt = get_runtime()
    dtype = beam.dtype
    batch_shape = beam.batch_shape
    lat = _ready(cache, program, batch_shape, dtype, beam._energy._host)
    mu_in = beam._mu_d.device(rt)
    cov_in = beam._cov_d.device(rt)
    mu_out = rt.empty_result(mu_in.shape, dtype)
    cov_out = rt.empty_result(cov_in.shape, dtype)
    e_in = beam._energy.broadcast_device(rt, batch_shape)
    e_out = rt.empty(batch_shape, dtype) if lat.has_cavity_step else None
    rt.check(rt.lib.lynx_track_moments(rt.ctx, lat.handle, _ptr(e_in), _ptr(mu_in), _ptr(cov_in),
                                       _ptr(mu_out), _ptr(cov_out), _ptr(e_out)))
    out = ParameterBeam.__new__(ParameterBeam)
    out._init_raw(Dual(dev=mu_out), Dual(dev=cov_out),
                  Dual(dev=e_out) if e_out is not None else beam._energy, beam.total_charge, dtype)
    return out


def track(owner, elements, incoming, raw: bool = False):
    """`Segment.track` (raw=False) / `Element.track` (raw=True) for both beam types."""
    late = _late()
    Beam, ParameterBeam, ParticleBeam = late.Beam, late.ParameterBeam, late.ParticleBeam
    if incoming is Beam.empty:
        for el in elements:
            if getattr(el, "_host_barrier", False):
                el._observe(incoming)
        return incoming
    if not isinstance(incoming, (ParameterBeam, ParticleBeam)):
        raise TypeError(f"Parameter incoming is of invalid type {type(incoming)}")
    cache = owner.__dict__.setdefault("_lattice_cache", LatticeCache())
    beam = incoming
    for item in plan(owner, elements, raw, fuse_observers=isinstance(incoming, ParticleBeam)):
        if isinstance(item, Program):
            if isinstance(beam, ParticleBeam):
                beam = run_program_particles(cache, item, beam)
            else:
                beam = run_program_parameters(cache, item, beam)
        else:  # host-side barrier: active BPM (bpm.py:48-58) or active Screen (screen.py:126-141)
            item._observe(beam)
            if getattr(item, "_swallows_beam", False):
                beam = Beam.empty
            elif hasattr(item, "_transform"):  # active Aperture (aperture.py:69-108)
                beam = item._transform(beam)
            elif beam is not Beam.empty:
                beam = beam._shallow_copy()
        if beam is Beam.empty:  # everything behind an active screen sees the empty beam
            seen = False
            for later in elements:
                if later is item:
                    seen = True
                elif seen and getattr(later, "_host_barrier", False):
                    later._observe(beam)
            return beam
    return beam


def transfer_map(owner, elements, energy, dtype, raw: bool = False) -> np.ndarray:
    """`transfer_map(energy)` of a skippable element list -> host array (*batch, 7, 7)."""
    rt = get_runtime()
    dtype = np.dtype(dtype)
    energy = np.asarray(energy, dtype=dtype)
    batch_shape = energy.shape
    items = plan(owner, elements, raw)
    if not items:
        out = np.zeros((*batch_shape, 7, 7), dtype=dtype)
        out[..., range(7), range(7)] = 1
        return out
    assert len(items) == 1 and isinstance(items[0], Program) and len(items[0].steps) == 1, (
        "transfer_map needs a skippable element list")
    program = items[0]
    cache = owner.__dict__.setdefault("_lattice_cache", LatticeCache())
    lat = _ready(cache, program, batch_shape, dtype, energy)
    e_in = rt.to_device(np.ascontiguousarray(energy))
    steps = rt.empty((lat.B, 1, _ffi.STEP_STRIDE), dtype)
    rt.check(rt.lib.lynx_build_compose(rt.ctx, lat.handle, _ptr(e_in), _ptr(steps), None))
    table = steps.numpy()
    return table[:, 0, :49].reshape(*batch_shape, 7, 7)


def cavity_rmatrix(cavity, energy, dtype) -> np.ndarray:
    """`Cavity.transfer_map` = `_cavity_rmatrix` (cavity.py:248-325) whether or not it is on."""
    rt = get_runtime()
    dtype = np.dtype(dtype)
    energy = np.asarray(energy, dtype=dtype)
    program = Program(raw=True)
    program.add_run_element(cavity, True)
    cache = cavity.__dict__.setdefault("_lattice_cache", LatticeCache())
    lat = _ready(cache, program, energy.shape, dtype, energy)
    e_in = rt.to_device(np.ascontiguousarray(energy))
    steps = rt.empty((lat.B, 1, _ffi.STEP_STRIDE), dtype)
    rt.check(rt.lib.lynx_build_compose(rt.ctx, lat.handle, _ptr(e_in), _ptr(steps), None))
    return steps.numpy()[:, 0, :49].reshape(*energy.shape, 7, 7)
