"""
CPU-side checks of the product's host layer: the C ABI library loads and exports every
symbol include/lynx_hip.h declares, the lattice packer / partition / flag evaluation, the
reference's error conventions that are decided on the host, and that compute fails LOUDLY
(no silent CPU fallback) when no GPU is present.
"""

import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

import lynx_amd as lx
from lynx_amd import _ffi, engine
from oracle import lynx_oracle as o

ROOT = Path(__file__).resolve().parent.parent


def _declared_functions():
    header = (ROOT / "include" / "lynx_hip.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    return sorted(set(re.findall(r"\b(lynx_[a-z0-9_]+)\s*\(", header)))


def test_library_exports_every_declared_symbol(built_library):
    lib = ctypes.CDLL(str(built_library))
    names = _declared_functions()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), f"liblynxhip.so does not export {name}"
    assert set(names) == set(_ffi.SIGNATURES), set(names) ^ set(_ffi.SIGNATURES)
    assert _ffi.load().lynx_version().startswith(b"lynxhip")


def test_struct_layouts_match_the_header():
    assert ctypes.sizeof(_ffi.Elem) == 16 and ctypes.sizeof(_ffi.Step) == 16
    header = (ROOT / "include" / "lynx_hip.h").read_text()
    for name, value in (("LYNX_KIND_CAVITY", _ffi.KIND_CAVITY), ("LYNX_KIND_CUSTOM", _ffi.KIND_CUSTOM),
                        ("LYNX_KIND_DIPOLE", _ffi.KIND_DIPOLE)):
        assert re.search(rf"{name}\s*=\s*{value}\b", header)
    for name, value in (("LYNX_FLAG_TILT", 1), ("LYNX_FLAG_MISALIGNED", 2), ("LYNX_FLAG_THICK", 4),
                        ("LYNX_FLAG_CAV_BETA", 8), ("LYNX_FLAG_CAV_GAIN", 16), ("LYNX_FLAG_CAV_T5XX", 32),
                        ("LYNX_STEP_FLAG_RAW", 64), ("LYNX_MOMENT_STRIDE", 36), ("LYNX_UNIQUE_ID_BYTES", 128)):
        assert re.search(rf"#define {name} {value}\b", header), name


def _gpu_present():
    n = ctypes.c_int(0)
    _ffi.load().lynx_device_count(ctypes.byref(n))
    return n.value > 0


def test_compute_fails_loudly_without_a_gpu(built_library):
    if _gpu_present():
        pytest.skip("a GPU is present; the loud-failure path is exercised on the CPU-only box")
    f = lambda v: np.array([v], dtype=np.float32)  # noqa: E731
    with pytest.raises(_ffi.LynxError, match="no HIP device|no ROCm"):
        lx.Drift(f(1.0)).track(lx.ParameterBeam.from_parameters())
    with pytest.raises(_ffi.LynxError):
        lx.Segment([lx.Drift(f(1.0))]).transfer_map(f(1e8))
    with pytest.raises(_ffi.LynxError):
        lx.ParticleBeam(np.ones((1, 4, 7), np.float32), f(1e8)).sigma_x


def test_missing_library_is_an_error_not_a_fallback(monkeypatch, tmp_path):
    monkeypatch.setenv("LYNX_HIP_LIBRARY", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_ffi, "_lib", None)
    with pytest.raises(_ffi.LynxError, match="not built"):
        _ffi.load()


def test_no_product_module_imports_the_oracle():
    for path in (ROOT / "lynx_amd").rglob("*.py"):
        text = path.read_text()
        assert "oracle" not in re.sub(r"#.*", "", text).replace("lynx_oracle.py", ""), path


def test_partition_matches_segment_track_todos():
    f = lambda v: np.array([v], dtype=np.float32)  # noqa: E731
    cav_on = lx.Cavity(f(1.0), voltage=f(1e6), name="on")
    cav_off = lx.Cavity(f(1.0), voltage=f(0.0), name="off")
    bpm = lx.BPM(name="bpm", is_active=True)
    els = [lx.Drift(f(1)), lx.BPM(), cav_on, lx.Marker(), lx.Drift(f(1)), bpm, cav_off, lx.Quadrupole(f(1), k1=f(1))]
    items = engine.partition(els)
    assert [type(i).__name__ for i in items] == ["Program", "BPM", "Program"]
    assert [s[0] for s in items[0].steps] == [_ffi.STEP_RUN, _ffi.STEP_CAVITY, _ffi.STEP_RUN]
    assert [(s[1], s[2]) for s in items[0].steps] == [(0, 2), (2, 3), (3, 5)]
    assert items[2].steps == [[_ffi.STEP_RUN, 0, 2]]
    specs = [o.Drift(f(1)), o.BPM(), o.Cavity(f(1), voltage=f(1e6)), o.Marker(), o.Drift(f(1)), o.BPM(is_active=True),
             o.Cavity(f(1), voltage=f(0.0)), o.Quadrupole(f(1), k1=f(1))]
    assert [k for k, _ in o.partition(specs)] == ["run", "single", "run", "single", "run"]
    # a nested non-skippable segment starts and ends a run; a skippable one joins the current run
    inner_ns = lx.Segment([lx.Drift(f(1)), lx.Cavity(f(1), voltage=f(1e6)), lx.Drift(f(1))])
    inner_s = lx.Segment([lx.Drift(f(1)), lx.Drift(f(2))])
    items = engine.partition([lx.Drift(f(1)), inner_s, inner_ns, lx.Drift(f(1))])
    assert [(s[0], s[1], s[2]) for s in items[0].steps] == [(0, 0, 3), (0, 3, 4), (1, 4, 5), (0, 5, 6), (0, 6, 7)]


def test_packed_lattice_layout_and_flags():
    B = 3
    f = lambda v: np.full(B, v, dtype=np.float32)  # noqa: E731
    quad = lx.Quadrupole(f(0.2), k1=np.array([1.0, 2.0, 3.0], np.float32), tilt=np.array([0, 0.1, 0], np.float32))
    drift1 = lx.Drift(np.array([0.5], np.float32))  # broadcastable: stored once
    dip = lx.Dipole(np.array([0.0, 0.5, 0.0], np.float32), angle=f(0.1))
    cav = lx.Cavity(f(1.0), voltage=np.array([0.0, 1e7, 0.0], np.float32), phase=f(10.0), frequency=f(1.3e9))
    prog = engine.partition([quad, drift1, lx.Marker(), dip, cav])[0]
    lat = engine.PackedLattice(prog, (B,), np.float32)
    kinds = [lat.elems[i].kind for i in range(lat.E)]
    assert kinds == [_ffi.KIND_QUADRUPOLE, _ffi.KIND_DRIFT, _ffi.KIND_IDENTITY, _ffi.KIND_DIPOLE, _ffi.KIND_CAVITY]
    assert [lat.elems[i].batch_stride for i in range(lat.E)] == [5, 0, 0, 8, 4]
    assert [lat.elems[i].param_offset for i in range(lat.E)] == [0, 15, 16, 16, 40]
    assert lat.pool.size == 15 + 1 + 24 + 12 and lat.pool.dtype == np.float32
    row1 = lat.pool[5:10]
    assert np.allclose(row1, [0.2, 2.0, 0.1, 0.0, 0.0])  # sample 1 of the quadrupole: L, k1, tilt, mx, my
    ef, sf = lat.evaluate_flags(f(1e8))
    assert ef[0] == _ffi.FLAG_TILT and ef[3] == _ffi.FLAG_THICK
    # cavity bits are whole-batch predicates of the beam ENERGY: evaluated on the device before every build
    # (k_cavity_flags; GPU test test_cavity_predicates_are_evaluated_on_the_device), never on the host
    assert ef[4] == 0 and sf == [0, 0]
    lat2 = engine.PackedLattice(engine.partition([cav])[0], (B,), np.float32)
    with pytest.raises(AssertionError, match="Initial energy must be larger than 0"):  # cavity.py:260
        lat2.evaluate_flags(f(0.0))
    assert lat2.evaluate_flags(None) == ([0], [0])  # energy in HBM only: no read-back for the sake of a check


def test_shape_mismatch_is_an_assertion_error():
    prog = engine.partition([lx.Drift(np.ones(2, np.float32))])[0]
    with pytest.raises(AssertionError, match="does not match element shape"):  # drift.py:45-47
        engine.PackedLattice(prog, (3,), np.float32)


def test_parameter_changes_refresh_the_packed_lattice():
    q = lx.Quadrupole(np.array([0.2], np.float32), k1=np.array([1.0], np.float32), name="Q")
    seg = lx.Segment([lx.Drift(np.array([1.0], np.float32)), q])
    v0 = q._version
    seg.Q.k1 = np.array([4.2], np.float32)  # README.md:60 style mutation
    assert q._version == v0 + 1
    cache = engine.LatticeCache()
    a = cache.get(engine.partition(seg.elements)[0], (1,), np.float32)
    assert cache.get(engine.partition(seg.elements)[0], (1,), np.float32) is a
    # same structure, new value: the changed element's pool block is rewritten in place
    q.k1 = np.array([5.0], np.float32)
    b = cache.get(engine.partition(seg.elements)[0], (1,), np.float32)
    assert b is a and b.pool[2] == 5.0 and b.versions[1] == q._version
    q.tilt = np.array([0.3], np.float32)  # static flags depend on the parameters: re-evaluated
    c = cache.get(engine.partition(seg.elements)[0], (1,), np.float32)
    assert c is a and c.evaluate_flags(np.array([1e8], np.float32))[0][1] & _ffi.FLAG_TILT
    # a parameter that stops being shared by the batch changes the block size: packed afresh
    cache2 = engine.LatticeCache()
    a2 = cache2.get(engine.partition(seg.elements)[0], (3,), np.float32)
    q.k1 = np.array([1.0, 2.0, 3.0], np.float32)
    seg.elements[0].length = np.ones(3, np.float32)
    q.length, q.tilt, q.misalignment = np.full(3, 0.2, np.float32), np.zeros(3, np.float32), np.zeros((3, 2), np.float32)
    b2 = cache2.get(engine.partition(seg.elements)[0], (3,), np.float32)
    assert b2 is not a2 and b2.layout[1][2] == 5 and np.array_equal(b2.pool[b2.layout[1][0] + 1::5][:3], [1.0, 2.0, 3.0])


def test_plan_is_reused_until_an_attribute_is_written():
    f = lambda v: np.array([v], dtype=np.float32)  # noqa: E731
    bpm = lx.BPM(name="B")
    seg = lx.Segment([lx.Drift(f(1.0)), lx.Quadrupole(f(0.2), k1=f(1.0), name="Q"), bpm, lx.Drift(f(1.0))])
    first = engine.plan(seg, seg.elements, False)
    assert engine.plan(seg, seg.elements, False) is first and len(first) == 1
    cache = engine.LatticeCache()
    lat = cache.get(first[0], (1,), np.float32)
    assert cache.get(first[0], (1,), np.float32) is lat  # O(1) path: same program object, same epoch
    seg.Q.k1 = f(2.0)  # a magnet strength cannot change the partition: same plan, refreshed pool block
    second = engine.plan(seg, seg.elements, False)
    assert second is first and cache.get(second[0], (1,), np.float32) is lat and lat.pool[lat.layout[1][0] + 1] == 2.0
    cav = lx.Cavity(f(1.0), voltage=f(0.0), name="C")
    with_cavity = lx.Segment([lx.Drift(f(1.0)), cav])
    off = engine.plan(with_cavity, with_cavity.elements, False)
    assert off[0].steps == [[_ffi.STEP_RUN, 0, 2]]
    cav.voltage = f(1e6)  # ... a cavity voltage can: the cavity becomes a step of its own
    assert engine.plan(with_cavity, with_cavity.elements, False)[0].steps == [[_ffi.STEP_RUN, 0, 1], [_ffi.STEP_CAVITY, 1, 2]]
    bpm.is_active = True  # structure changes: the BPM becomes a host-side barrier
    third = engine.plan(seg, seg.elements, False)
    assert [type(i).__name__ for i in third] == ["Program", "BPM", "Program"]
    seg.elements.append(lx.Drift(f(0.5)))  # in-place list mutation, no attribute write
    assert len(engine.plan(seg, seg.elements, False)[-1].leaves) == 2
    inner = lx.Segment([lx.Drift(f(0.1))])
    outer = lx.Segment([inner, lx.Drift(f(0.2))])
    assert len(engine.plan(outer, outer.elements, False)[0].leaves) == 2
    inner.elements.append(lx.Drift(f(0.3)))  # nested list mutated in place
    assert len(engine.plan(outer, outer.elements, False)[0].leaves) == 3


def test_segment_container_api():
    f = lambda v: np.array([v], dtype=np.float32)  # noqa: E731
    seg = lx.Segment([lx.BPM(name="BPM1"), lx.Drift(f(1.0), name="D"), lx.Drift(f(2.0), name="D"),
                      lx.Quadrupole(f(0.2), k1=f(0.0), name="Q"), lx.Marker(name="M"),
                      lx.HorizontalCorrector(f(0.3), name="H")])
    assert isinstance(seg.D, list) and len(seg.D) == 2 and seg.Q.name == "Q"  # segment.py:45-54
    assert np.isclose(seg.length, 3.5) and seg.is_skippable
    assert [e.name for e in seg.subcell("D", "Q").elements] == ["D", "D", "Q"]
    assert not any(isinstance(e, lx.Marker) for e in seg.without_inactive_markers().elements)
    drifts = seg.inactive_elements_as_drifts().elements
    assert isinstance(drifts[3], lx.Drift) and isinstance(drifts[5], lx.Drift) and isinstance(drifts[0], lx.BPM)
    assert [e.name for e in seg.without_inactive_zero_length_elements().elements] == ["D", "D", "Q", "H"]
    nested = lx.Segment([seg, lx.Drift(f(1.0))])
    assert len(nested.flattened().elements) == 7
    b = seg.broadcast((3, 2))
    assert b.length.shape == (3, 2) and b.Q.k1.shape == (3, 2) and b.Q.misalignment.shape == (3, 2, 2)
    assert b.H.angle.shape == (1,)  # horizontal_corrector.py:69-72 does not repeat `angle`
    seg.BPM1.is_active = True
    assert not seg.is_skippable


def test_beam_constructors_follow_the_reference():
    b = lx.ParameterBeam.from_parameters()
    assert b._mu.shape == (1, 7) and b._cov.shape == (1, 7, 7) and b.energy[0] == 1e8
    assert np.isclose(b.sigma_x, 175e-9) and np.isclose(b.sigma_xp, 2e-7) and np.isclose(b.sigma_s, 1e-6)
    with pytest.raises(AssertionError, match="Arguments must have the same shape"):  # parameter_beam.py:91-94
        lx.ParameterBeam.from_parameters(mu_x=np.zeros(2), sigma_x=np.ones(3))
    t = lx.ParameterBeam.from_twiss(beta_x=np.array([5.91253676811640894]), alpha_x=np.array([3.55631307633660354]),
                                    emittance_x=np.array([3.494768647122823e-09]), energy=np.array([6e6]))
    assert np.isclose(t.beta_x, 5.91253676811640894) and np.isclose(t.alpha_x, 3.55631307633660354)
    assert np.isclose(t.emittance_x, 3.494768647122823e-09)
    assert np.isclose(t.relativistic_gamma, 6e6 / 510998.95069)
    bb = b.broadcast((3, 10))
    assert bb._mu.shape == (3, 10, 7) and bb._cov.shape == (3, 10, 7, 7) and bb.energy.shape == (3, 10)
    tr = b.transformed_to(mu_x=np.array([1e-5], np.float32), sigma_x=np.array([1.75e-7], np.float32))
    assert np.isclose(tr.mu_x, 1e-5) and np.isclose(tr.sigma_x, 1.75e-7)
    with pytest.raises(AssertionError, match="7-dimensional"):  # particle_beam.py:35-37
        lx.ParticleBeam(np.ones((1, 10, 6), np.float32), np.array([1e8]))
    p = lx.ParticleBeam.from_parameters(num_particles=1000, sigma_x=np.array([1e-5, 2e-5]), seed=0)
    assert p.particles.shape == (2, 1000, 7) and np.all(p.particles[..., 6] == 1) and p.num_particles == 1000
    assert p.particle_charges.shape == (2, 1000) and p.total_charge.shape == (2,)
    assert np.allclose(np.std(p.xs, axis=-1), [1e-5, 2e-5], rtol=0.1)
    lin = lx.ParticleBeam.make_linspaced(num_particles=11, sigma_x=np.array([1e-3], np.float32))
    assert np.allclose(lin.xs[0], np.linspace(-1e-3, 1e-3, 11))
    bp = p.broadcast((3,))
    assert bp.particles.shape == (6, 1000, 7)  # Tensor.repeat semantics on a (2,)-batch


def test_uniform_3d_ellipsoid_sampler():
    """reference tests/test_particle_beam.py:108-146"""
    radius_x, radius_y, radius_s = np.array([1e-3, 2e-3]), np.array([1e-4, 2e-4]), np.array([1e-5, 2e-5])
    beam = lx.ParticleBeam.uniform_3d_ellipsoid(
        num_particles=20_000, radius_x=radius_x, radius_y=radius_y, radius_s=radius_s, sigma_xp=np.array([2e-7, 1e-7]),
        sigma_yp=np.array([3e-7, 2e-7]), sigma_p=np.array([1e-6, 2e-6]), energy=np.array([1e7, 2e7]),
        total_charge=np.array([1e-9, 3e-9]), seed=1)
    assert beam.num_particles == 20_000
    assert np.all(np.abs(beam.xs).T <= radius_x) and np.all(np.abs(beam.ys).T <= radius_y)
    assert np.all(np.abs(beam.ss).T <= radius_s)
    assert np.allclose(np.std(beam.xps, axis=-1), [2e-7, 1e-7], rtol=0.05)
    assert np.allclose(beam.energy, [1e7, 2e7]) and np.allclose(beam.total_charge, [1e-9, 3e-9], rtol=1e-5)


def test_gradient_host_helpers():
    """Host side of lynx_amd.grad: broadcast parameters receive summed gradients; a full
    symmetric cov cotangent maps to the upper-triangle record of include/lynx_hip.h."""
    from lynx_amd.grad import _unbroadcast
    from lynx_amd.particles.particle_beam import _tri

    g = np.arange(6.0).reshape(3, 2)
    assert np.array_equal(_unbroadcast(g, (3, 2)), g)
    assert np.array_equal(_unbroadcast(g, (1,)), [15.0])
    assert np.array_equal(_unbroadcast(g, (1, 2)), [[6.0, 9.0]])
    assert np.array_equal(_unbroadcast(g, (2,)), [6.0, 9.0])
    idx = sorted(_tri(i, j) for i in range(6) for j in range(i, 6))
    assert idx == list(range(7, 28))  # the 21 upper-triangle slots of a moment record
    assert _tri(0, 0) == 7 and _tri(0, 5) == 12 and _tri(1, 1) == 13 and _tri(5, 5) == 27


def test_header_is_plain_c(tmp_path, built_library):
    """The drop-in boundary is a C ABI: include/lynx_hip.h must compile as C99 and link from a C client."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    header = ROOT / "include" / "lynx_hip.h"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", str(header)], check=True)
    client = tmp_path / "client.c"
    client.write_text(
        '#include "lynx_hip.h"\n#include <stdio.h>\n'
        'int main(void) { int n = -1; printf("%s %d\\n", lynx_version(), lynx_device_count(&n)); '
        'lynx_elem e = {LYNX_KIND_QUADRUPOLE, LYNX_FLAG_TILT, 0, 5}; lynx_step s = {LYNX_STEP_RUN, 0, 1, 0}; '
        'return (int)(sizeof e + sizeof s) == 32 ? 0 : 1; }\n')
    lib = ROOT / "lynx_amd" / "_lib"
    exe = tmp_path / "client"
    subprocess.run([gcc, "-std=c99", f"-I{ROOT / 'include'}", str(client), f"-L{lib}", "-llynxhip", f"-Wl,-rpath,{lib}",
                    "-o", str(exe)], check=True)
    done = subprocess.run([str(exe)], capture_output=True, text=True)
    assert done.returncode == 0 and done.stdout.startswith("lynx"), (done.returncode, done.stdout, done.stderr)


# ---------------------------------------------------------------------------------------------
# cached device state cannot go stale behind the caller's back
# ---------------------------------------------------------------------------------------------


def test_element_parameters_are_private_read_only_copies():
    """`quad.k1[0] = 5` cannot be seen by any version counter: it must raise, and the caller's
    own array must not be aliased (the reference recomputes from live tensors on every call)."""
    k1 = np.array([4.2, -4.2], dtype=np.float32)
    quad = lx.Quadrupole(np.array([0.2, 0.2], dtype=np.float32), k1=k1)
    with pytest.raises(ValueError, match="read-only"):
        quad.k1[0] = 5.0
    with pytest.raises(ValueError, match="read-only"):
        quad.misalignment[..., 0] += 1e-3
    k1[0] = 9.0  # the caller's array is theirs
    assert quad.k1[0] == np.float32(4.2)
    version = quad._version
    quad.k1 = k1  # assignment is the supported way and is seen
    assert quad._version == version + 1 and quad.k1[0] == np.float32(9.0)
    assert not quad.broadcast((3,)).k1.flags.writeable


def test_beam_arrays_are_private_read_only_copies():
    P = np.zeros((1, 4, 7), dtype=np.float32)
    P[..., 6] = 1
    beam = lx.ParticleBeam(P, np.array([1e8], dtype=np.float32))
    with pytest.raises(ValueError, match="read-only"):
        beam.particles[..., 0] -= 1e-3
    P[0, 0, 0] = 7.0
    assert np.asarray(beam.particles)[0, 0, 0] == 0.0
    pb = lx.ParameterBeam.from_parameters()
    with pytest.raises(ValueError, match="read-only"):
        pb._mu[..., 0] = 1.0
    with pytest.raises(ValueError, match="read-only"):
        pb.energy[0] = 1.0


def test_rank_local_device_beyond_the_visible_gpus_is_an_error(built_library, monkeypatch):
    """Two ranks on one GPU is how round 1's N=2 rehearsal ended in `ncclCommInitRank: invalid usage`."""
    from lynx_amd import device

    n = ctypes.c_int(0)
    _ffi.load().lynx_device_count(ctypes.byref(n))
    monkeypatch.setenv("LYNX_DEVICE", str(max(n.value, 1) + 3))
    monkeypatch.delenv("LYNX_ALLOW_GPU_SHARING", raising=False)
    with pytest.raises(_ffi.LynxError, match="GPU|device"):
        device.Runtime()


def test_objects_that_outlive_a_closed_runtime_are_dropped_quietly(built_library):
    """
    Round 1's `pytest exit 139` (gpurun_out/pytest1.log): `PackedLattice.__del__` and array
    finalizers ran at interpreter shutdown AFTER the context had been destroyed and called
    `lynx_lattice_destroy` / `lynx_buf_free` on freed memory.  Since then a closed runtime turns
    every later release into a no-op; this pins it without a GPU (no C entry point may be
    reached once `closed` is set -- the stand-in library raises if one is).
    """
    from lynx_amd import device, parallel

    class Tripwire:
        def __getattr__(self, name):
            raise AssertionError(f"{name} called on a closed runtime")

    rt = device.Runtime.__new__(device.Runtime)
    rt.lib, rt.ctx, rt.device, rt.closed = Tripwire(), ctypes.c_void_p(1), 0, True
    arr = device.DeviceArray.__new__(device.DeviceArray)
    arr.rt, arr.ptr = rt, 1234
    rt.free(arr.ptr)  # what the array's weakref finalizer calls
    lat = engine.PackedLattice.__new__(engine.PackedLattice)
    lat.handle, lat.rt = ctypes.c_void_p(99), rt
    lat.release()
    assert lat.handle is None
    comm = parallel.RcclCommunicator.__new__(parallel.RcclCommunicator)
    comm.rt = rt
    comm.close()
    rt.close()  # idempotent
