"""
Gradients of the outgoing moments w.r.t. element parameters (lynx_track_particles_backward)
against central finite differences of the ORACLE's forward pass in float64.  The reference
has no gradient implementation to compare with (parity unpinned).
"""

import numpy as np
import pytest

from oracle import lynx_oracle as o

from .helpers import make_lattice, singular_entry_voltage

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lx(built_library):
    import lynx_amd
    import lynx_amd.grad  # noqa: F401

    lynx_amd.device.get_runtime()
    return lynx_amd


def _loss(specs, P, energy, w_mu, w_cov):
    out = o.segment_track(specs, o.particle_beam(P, energy, np.float64), np.float64)
    Q = out["particles"][..., :6]
    mu = Q.mean(axis=-2)
    d = Q - mu[..., None, :]
    cov = np.einsum("...ni,...nj->...ij", d, d) / Q.shape[-2]
    return np.sum(w_mu * mu, axis=-1) + np.sum(w_cov * cov, axis=(-1, -2))  # (B,)


def _desc(B, rng):
    f = lambda v: np.full(B, v)  # noqa: E731
    return [("drift", dict(length=f(0.6))),
            ("quadrupole", dict(length=f(0.2), k1=rng.uniform(-5, 5, B), tilt=rng.uniform(-0.5, 0.5, B),
                                misalignment=rng.normal(0, 1e-3, (B, 2)))),
            ("hcor", dict(length=f(0.1), angle=rng.normal(0, 1e-3, B))),
            ("cavity", dict(length=f(1.0377), voltage=rng.uniform(5e6, 2e7, B), phase=rng.uniform(-10, 10, B),
                            frequency=f(1.3e9))),
            ("dipole", dict(length=f(0.5), angle=rng.uniform(0.05, 0.2, B), e1=f(0.05), e2=f(0.02),
                            fringe_integral=f(0.4), fringe_integral_exit=f(0.3), gap=f(0.02), tilt=f(0.1))),
            ("vcor", dict(length=f(0.1), angle=rng.normal(0, 1e-3, B))),
            ("cavity", dict(length=f(1.0), voltage=rng.uniform(5e6, 2e7, B), phase=f(3.0), frequency=f(1.3e9))),
            ("quadrupole", dict(length=f(0.3), k1=rng.uniform(-5, 5, B))),
            ("drift", dict(length=f(0.4)))]


PARAMS_TO_CHECK = {"drift": ["length"], "quadrupole": ["length", "k1", "tilt", "misalignment"],
                   "hcor": ["length", "angle"], "vcor": ["angle"], "cavity": ["length", "voltage", "phase", "frequency"],
                   "dipole": ["length", "angle", "e1", "e2", "tilt", "fringe_integral", "fringe_integral_exit", "gap"]}


def test_gradients_match_finite_differences_fp64(lx):
    rng = np.random.default_rng(42)
    B, N = 2, 400
    desc = _desc(B, rng)
    elements, specs = make_lattice(desc, np.float64, lx)
    seg = lx.Segment(elements)
    P = o.gaussian_particles((B,), N, seed=9, dtype=np.float64, sigma=[1e-3, 1e-4, 1e-3, 1e-4, 1e-3, 1e-3],
                             mu=[1e-3, -1e-4, 5e-4, 2e-4, 1e-4, 1e-3])
    energy = np.array([6e6, 8e6])
    w_mu = rng.normal(size=(B, 6))
    w_cov = rng.normal(size=(B, 6, 6)) * 1e3
    vjp = lx.grad.track_vjp(seg, lx.ParticleBeam(P, energy, dtype=np.float64))
    g = vjp(mu_bar=w_mu, cov_bar=w_cov)

    def fd(apply, x0):
        h = 1e-6 * max(abs(x0), 1e-2)
        apply(x0 + h)
        lp = _loss(specs, P, energy_fd[0], w_mu, w_cov)
        apply(x0 - h)
        lm = _loss(specs, P, energy_fd[0], w_mu, w_cov)
        apply(x0)
        return (lp - lm) / (2 * h)

    energy_fd = [energy.copy()]
    checked = 0
    for e, (kind, _) in enumerate(desc):
        for name in PARAMS_TO_CHECK.get(kind, []):
            arr = specs[e][name]
            if arr is None:  # parameter left at its default in this element
                continue
            got = g[elements[e]][name]
            assert got.shape == arr.shape
            for idx in np.ndindex(arr.shape):
                def apply(x, arr=arr, idx=idx):
                    arr[idx] = x
                ref = fd(apply, arr[idx])[idx[0]]  # sample idx[0] only depends on its own parameters
                scale = max(abs(ref), 1e-9 * np.max(np.abs(w_cov)))
                assert abs(got[idx] - ref) <= 2e-4 * scale + 1e-7 * np.max(np.abs(got)), (kind, e, name, idx, got[idx], ref)
                checked += 1
    for bidx in range(B):
        def apply(x, bidx=bidx):
            energy_fd[0] = energy.copy()
            energy_fd[0][bidx] = x
        ref = fd(apply, energy[bidx])[bidx]
        energy_fd[0] = energy.copy()
        assert abs(g.energy[bidx] - ref) <= 2e-4 * abs(ref) + 1e-12, (bidx, g.energy[bidx], ref)
    assert checked > 50


@pytest.mark.parametrize("B,N,ill", [(2, 3000, False), (300, 256, False), (2, 3000, True), (300, 256, True)],
                         ids=["workgroup-build", "lanes-build", "workgroup-build-ill", "lanes-build-ill"])
def test_merged_pairs_reverse_pass_fp32(lx, monkeypatch, B, N, ill):
    """
    float32 reverse pass over [run, cavity] pairs in the merged form of the forward kernel (one 7x7 application
    per pair, kick driven by two entry rows; k_build_bwd takes M_bar apart into T_run_bar and T_cav_bar):
    every parameter gradient against (a) the step-by-step float32 reverse pass (LYNX_BWD_MERGE=0) and (b) the
    float64 reverse pass, which the test above pins to finite differences of the oracle.  `ill`: a 1 MeV beam whose
    first cavity (30 degrees off crest) sits next to the singular point of its map's (s, delta) block (LYNX_DESC_ILL: the forward sweep of
    the reverse pass then drives the kick from the run's rows, like the forward kernels).
    """
    rng = np.random.default_rng(42)
    desc = _desc(B, rng)
    P = o.gaussian_particles((B,), N, seed=9, dtype=np.float64, sigma=[1e-3, 1e-4, 1e-3, 1e-4, 1e-3, 1e-3],
                             mu=[1e-3, -1e-4, 5e-4, 2e-4, 1e-4, 1e-3])
    energy = rng.uniform(6e6, 8e6, B)
    if ill:
        energy = np.full(B, 1e6)
        kind, params = desc[3]
        desc[3] = (kind, dict(params, length=np.full(B, 1.0), phase=np.full(B, -30.0),
                              voltage=singular_entry_voltage(1e6, -30.0, 1.0, 1.3e9, 4e5, 6.5e5) * (1 + rng.uniform(1e-4, 1e-2, B))))
        desc[6] = (desc[6][0], dict(desc[6][1], voltage=rng.uniform(2e5, 5e5, B)))
    w_mu = rng.normal(size=(B, 6))
    w_cov = rng.normal(size=(B, 6, 6)) * 1e3

    def gradients(dtype):
        elements, _ = make_lattice(desc, dtype, lx)
        g = lx.grad.track_vjp(lx.Segment(elements), lx.ParticleBeam(P.astype(dtype), energy.astype(dtype), dtype=dtype))(
            mu_bar=w_mu, cov_bar=w_cov)
        out = {(e, name): np.asarray(g[elements[e]][name], dtype=np.float64)
               for e, (kind, _) in enumerate(desc) for name in PARAMS_TO_CHECK.get(kind, [])
               if getattr(elements[e], name, None) is not None}
        out["energy"] = np.asarray(g.energy, dtype=np.float64)
        return out

    g64 = gradients(np.float64)
    merged = gradients(np.float32)
    monkeypatch.setenv("LYNX_BWD_MERGE", "0")
    stepwise = gradients(np.float32)
    monkeypatch.delenv("LYNX_BWD_MERGE")
    assert sum(v.size for v in g64.values()) > 50
    for key, ref in g64.items():
        floor = 1e-4 * max(np.max(np.abs(ref)), 1e-9 * np.max(np.abs(w_cov)))
        for name, got in (("merged", merged[key]), ("stepwise", stepwise[key])):
            assert got.shape == ref.shape
            assert np.all(np.abs(got - ref) <= 3e-3 * np.abs(ref) + floor), (name, key, got, ref)
        assert np.all(np.abs(merged[key] - stepwise[key]) <= 1e-3 * np.abs(ref) + floor), (key, merged[key], stepwise[key])
    # the two float32 passes are different computations (they round differently): the switch did switch
    assert any(np.any(merged[key] != stepwise[key]) for key in g64)


@pytest.mark.parametrize("B,N", [(3, 4000), (300, 512)], ids=["workgroup-build", "lanes-build"])
def test_structured_reverse_pass_fp32(lx, monkeypatch, B, N):
    """
    Samples whose units all have class U (runs of drifts, correctors, misaligned quadrupoles; cavities) take the
    structured reverse kernel (k_track_bwd_units: 16 particle products per unit instead of 49, unit records instead of
    step-table rows), the others the dense one.  Every parameter gradient and the gradient w.r.t. the incoming
    particles against (a) the dense float32 pass (LYNX_BWD_UNITS=0) and (b) the float64 pass, which
    test_gradients_match_finite_differences_fp64 pins to finite differences of the oracle.  With a tilted quadrupole in
    the lattice (coupled map: no unit of that run has the structure) both settings run the dense kernel: identical.
    """
    rng = np.random.default_rng(43)
    f = lambda v: np.full(B, v)  # noqa: E731

    def lattice(tilted):
        desc = []
        for cell in range(3):
            desc += [("drift", dict(length=f(0.3))),
                     ("quadrupole", dict(length=f(0.1), k1=rng.uniform(-5, 5, B), misalignment=rng.normal(0, 1e-3, (B, 2)),
                                         **({"tilt": rng.uniform(-0.3, 0.3, B)} if tilted and cell == 1 else {}))),
                     ("hcor", dict(length=f(0.1), angle=rng.normal(0, 1e-3, B))),
                     ("drift", dict(length=f(0.3))),
                     ("cavity", dict(length=f(1.0377), voltage=rng.uniform(5e6, 2e7, B), phase=rng.uniform(-10, 10, B),
                                     frequency=f(1.3e9)))]
        return desc + [("quadrupole", dict(length=f(0.2), k1=rng.uniform(-5, 5, B))), ("drift", dict(length=f(0.4)))]

    P = o.gaussian_particles((B,), N, seed=9, dtype=np.float64, sigma=[1e-3, 1e-4, 1e-3, 1e-4, 1e-3, 1e-3],
                             mu=[1e-3, -1e-4, 5e-4, 2e-4, 1e-4, 1e-3])
    energy = rng.uniform(6e6, 8e6, B)
    w_mu = rng.normal(size=(B, 6))
    w_cov = rng.normal(size=(B, 6, 6)) * 1e3

    def gradients(desc, dtype):
        elements, _ = make_lattice(desc, dtype, lx)
        g = lx.grad.track_vjp(lx.Segment(elements), lx.ParticleBeam(P.astype(dtype), energy.astype(dtype), dtype=dtype))(
            mu_bar=w_mu, cov_bar=w_cov, wrt_particles=True)
        out = {(e, name): np.asarray(g[elements[e]][name], dtype=np.float64)
               for e, (kind, _) in enumerate(desc) for name in PARAMS_TO_CHECK.get(kind, [])
               if getattr(elements[e], name, None) is not None}
        out["energy"] = np.asarray(g.energy, dtype=np.float64)
        out["particles"] = np.asarray(g.particles, dtype=np.float64)[..., :6]
        return out

    for tilted in (False, True):
        desc = lattice(tilted)
        g64 = gradients(desc, np.float64)
        structured = gradients(desc, np.float32)
        monkeypatch.setenv("LYNX_BWD_UNITS", "0")
        dense = gradients(desc, np.float32)
        monkeypatch.delenv("LYNX_BWD_UNITS")
        for key, ref in g64.items():
            floor = 1e-4 * max(np.max(np.abs(ref)), 1e-9 * np.max(np.abs(w_cov)))
            for name, got in (("structured", structured[key]), ("dense", dense[key])):
                assert got.shape == ref.shape
                assert np.all(np.abs(got - ref) <= 3e-3 * np.abs(ref) + floor), (tilted, name, key, got, ref)
            assert np.all(np.abs(structured[key] - dense[key]) <= 1e-3 * np.abs(ref) + floor), (tilted, key)
        if tilted:  # the middle run is coupled: ... for every sample (the tilt flag is a whole-batch predicate)
            assert all(np.array_equal(structured[key], dense[key]) for key in g64)
        else:       # different kernels: the sums over particles are associated differently
            assert any(np.any(structured[key] != dense[key]) for key in g64)


def test_samples_the_structured_reverse_kernel_leaves_to_the_dense_one(lx, monkeypatch):
    """
    Next to k_track_bwd_units the dense k_track_bwd is launched for the samples that kernel leaves.  When the lattice's
    plan proposes class U for every unit, such a sample can only be one whose map is not finite, and the dense kernel
    gets ONE workgroup per sample (all of the sample's tiles, row 0 of its partial sums, the other rows cleared) instead
    of `chunks` of them that return at once.  (a) LYNX_BWD_UNITS=2 leaves every odd sample to that launch whatever its
    class: every gradient as from the structured pass alone (default) and from the dense pass alone (LYNX_BWD_UNITS=0),
    to the sums' re-association, and the float64 pass as in test_structured_reverse_pass_fp32.  (b) a cavity without
    voltage in ONE sample (NaN map, cavity.py:269): the other samples' gradients are what they are without it.
    """
    B, N = 4, 6000  # 12 tiles of 512 per sample: several workgroups per sample in the structured kernel
    rng = np.random.default_rng(47)
    f = lambda v: np.full(B, v)  # noqa: E731
    volts = [rng.uniform(5e6, 2e7, B) for _ in range(3)]

    def lattice(dead=None):
        r = np.random.default_rng(48)
        desc = []
        for cell in range(3):
            v = volts[cell].copy()
            if dead is not None and cell == 1:
                v[dead] = 0.0
            desc += [("drift", dict(length=f(0.3))),
                     ("quadrupole", dict(length=f(0.1), k1=r.uniform(-5, 5, B), misalignment=r.normal(0, 1e-3, (B, 2)))),
                     ("drift", dict(length=f(0.3))),
                     ("cavity", dict(length=f(1.0377), voltage=v, phase=r.uniform(-10, 10, B), frequency=f(1.3e9)))]
        return desc

    P = o.gaussian_particles((B,), N, seed=10, dtype=np.float64, sigma=[1e-3, 1e-4, 1e-3, 1e-4, 1e-3, 1e-3],
                             mu=[1e-3, -1e-4, 5e-4, 2e-4, 1e-4, 1e-3])
    energy = rng.uniform(6e6, 8e6, B)
    w_mu = rng.normal(size=(B, 6))
    w_cov = rng.normal(size=(B, 6, 6)) * 1e3

    def gradients(desc, dtype):
        elements, _ = make_lattice(desc, dtype, lx)
        g = lx.grad.track_vjp(lx.Segment(elements), lx.ParticleBeam(P.astype(dtype), energy.astype(dtype), dtype=dtype))(
            mu_bar=w_mu, cov_bar=w_cov, wrt_particles=True)
        out = {(e, name): np.asarray(g[elements[e]][name], dtype=np.float64)
               for e, (kind, _) in enumerate(desc) for name in PARAMS_TO_CHECK.get(kind, [])
               if getattr(elements[e], name, None) is not None}
        out["energy"] = np.asarray(g.energy, dtype=np.float64)
        out["particles"] = np.asarray(g.particles, dtype=np.float64)[..., :6]
        return out

    def with_units(value, desc):
        monkeypatch.setenv("LYNX_BWD_UNITS", value)
        try:
            return gradients(desc, np.float32)
        finally:
            monkeypatch.delenv("LYNX_BWD_UNITS")

    # (a)
    desc = lattice()
    g64 = gradients(desc, np.float64)
    structured, mixed, dense = with_units("1", desc), with_units("2", desc), with_units("0", desc)
    for key, ref in g64.items():
        floor = 1e-4 * max(np.max(np.abs(ref)), 1e-9 * np.max(np.abs(w_cov)))
        assert np.all(np.abs(mixed[key] - ref) <= 3e-3 * np.abs(ref) + floor), key
        assert np.all(np.abs(mixed[key] - structured[key]) <= 1e-3 * np.abs(ref) + floor), key
        assert np.all(np.abs(mixed[key] - dense[key]) <= 1e-3 * np.abs(ref) + floor), key
        # even samples went through the structured kernel as in the default pass: the same bits
        assert np.array_equal(mixed[key][0::2], structured[key][0::2]), key
    assert any(np.any(mixed[key][1::2] != structured[key][1::2]) for key in g64)  # (the odd ones through another kernel)
    # (b)
    dead = 2
    with_dead = gradients(lattice(dead), np.float32)
    alive = [b for b in range(B) if b != dead]
    for key, ref in structured.items():
        assert np.array_equal(with_dead[key][alive], ref[alive]), key  # sample by sample: nothing of the dead one leaks
    assert not np.all(np.isfinite(with_dead["particles"][dead]))


@pytest.mark.parametrize("dtype,B,N", [(np.float32, 3, 4000), (np.float32, 300, 512), (np.float64, 2, 3000)],
                         ids=["fp32-workgroup-build", "fp32-lanes-build", "fp64"])
def test_the_reverse_pass_reads_the_table_of_its_forward_call(lx, monkeypatch, dtype, B, N):
    """
    A reverse pass that directly follows its forward call (same lattice, incoming energy, merge form; nothing written
    in between) reads the step table and unit records that call built instead of building its own
    (lynx_ctx::FwdTable; LYNX_BWD_REUSE_TABLE=0: always its own).  The same numbers go into the same kernels: every
    gradient bit for bit.  What must break the hand-over does: another forward call in between (the slots move on), a
    parameter written in between (the gradient is the one at the NEW value, as it always was), and -- many steps in a
    row, each of them a forward and a reverse pass -- the table slot's next build waits for the reverse pass that
    still reads it.
    """
    rng = np.random.default_rng(47)
    f = lambda v: np.full(B, v)  # noqa: E731
    desc = []
    for cell in range(3):
        desc += [("drift", dict(length=f(0.3))),
                 ("quadrupole", dict(length=f(0.1), k1=rng.uniform(-5, 5, B), misalignment=rng.normal(0, 1e-3, (B, 2)))),
                 ("hcor", dict(length=f(0.1), angle=rng.normal(0, 1e-3, B))), ("drift", dict(length=f(0.3))),
                 ("cavity", dict(length=f(1.0377), voltage=rng.uniform(5e6, 2e7, B), phase=rng.uniform(-10, 10, B), frequency=f(1.3e9)))]
    P = o.gaussian_particles((B,), N, seed=9, dtype=np.float64, sigma=[1e-3, 1e-4, 1e-3, 1e-4, 1e-3, 1e-3]).astype(dtype)
    energy = rng.uniform(6e6, 8e6, B).astype(dtype)
    w_cov = rng.normal(size=(B, 6, 6)) * 1e3

    def flat(g, elements):
        out = [np.asarray(g.energy)]
        for e, (kind, _) in enumerate(desc):
            out += [np.asarray(g[elements[e]][name]) for name in PARAMS_TO_CHECK.get(kind, []) if getattr(elements[e], name, None) is not None]
        return out

    def same(a, b):
        return all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b))

    elements, _ = make_lattice(desc, dtype, lx)
    seg = lx.Segment(elements)
    beam = lx.ParticleBeam(P, energy, dtype=dtype)
    handed_over = flat(lx.grad.track_vjp(seg, beam)(cov_bar=w_cov), elements)
    monkeypatch.setenv("LYNX_BWD_REUSE_TABLE", "0")
    own_table = flat(lx.grad.track_vjp(seg, beam)(cov_bar=w_cov), elements)
    monkeypatch.delenv("LYNX_BWD_REUSE_TABLE")
    assert same(handed_over, own_table)
    # another forward call between the two halves
    vjp = lx.grad.track_vjp(seg, beam)
    other, _ = make_lattice(desc[:5], dtype, lx)
    lx.Segment(other).track(beam)
    assert same(flat(vjp(cov_bar=w_cov), elements), own_table)
    # a parameter written between the two halves: the reverse pass builds at the new value
    quad = elements[1]
    k1_old, k1_new = np.asarray(quad.k1).copy(), (np.asarray(quad.k1) * 0.5).astype(dtype)
    vjp = lx.grad.track_vjp(seg, beam)
    quad.k1 = k1_new
    moved = flat(vjp(cov_bar=w_cov), elements)
    monkeypatch.setenv("LYNX_BWD_REUSE_TABLE", "0")
    quad.k1 = k1_old
    vjp2 = lx.grad.track_vjp(seg, beam)
    quad.k1 = k1_new
    assert same(moved, flat(vjp2(cov_bar=w_cov), elements))
    monkeypatch.delenv("LYNX_BWD_REUSE_TABLE")
    quad.k1 = k1_old
    # many steps in a row without a wait in between
    runs = [lx.grad.track_vjp(seg, beam)(cov_bar=w_cov) for _ in range(8)]
    for g in runs:
        assert same(flat(g, elements), own_table)


def test_gradient_of_linear_lattice_and_broadcast_parameters(lx):
    """All-skippable lattice (one composed map), fp32; a parameter shared by the whole batch
    receives the sum of the per-sample gradients."""
    rng = np.random.default_rng(1)
    B, N = 3, 2000
    k1 = np.array([4.2], dtype=np.float32)  # shared by all samples
    elements = [lx.Drift(np.array([0.5], np.float32)), lx.Quadrupole(np.array([0.2], np.float32), k1=k1, name="Q"),
                lx.Drift(np.full(B, 0.7, np.float32))]
    seg = lx.Segment(elements)
    P = o.gaussian_particles((B,), N, seed=2, dtype=np.float32, sigma=[1e-3, 1e-4, 1e-3, 1e-4, 1e-3, 1e-3])
    w_cov = np.zeros((B, 6, 6))
    w_cov[:, 0, 0] = 1.0  # L = sum_b var(x)_b
    g = lx.grad.track_vjp(seg, lx.ParticleBeam(P, np.full(B, 1e8, np.float32)))(cov_bar=w_cov)
    assert g[seg.Q]["k1"].shape == (1,)

    def loss(kv):
        specs = [o.Drift(np.full(B, 0.5)), o.Quadrupole(np.full(B, 0.2), k1=np.full(B, kv)), o.Drift(np.full(B, 0.7))]
        return _loss(specs, P.astype(np.float64), np.full(B, 1e8), np.zeros((B, 6)), w_cov).sum()

    ref = (loss(4.2 + 1e-4) - loss(4.2 - 1e-4)) / 2e-4
    assert np.isclose(g[seg.Q]["k1"][0], ref, rtol=2e-3), (g[seg.Q]["k1"], ref)


def test_gradient_wrt_incoming_particles(lx):
    """
    reference tests/test_differentiable.py:75-91 makes the incoming particles the leaf.  Here:
    dL/d(particle coordinates) of a cavity lattice against finite differences of the oracle for a
    handful of particles and coordinates, and the total against a directional derivative.
    """
    rng = np.random.default_rng(7)
    B, N = 2, 300
    desc = _desc(B, rng)
    elements, specs = make_lattice(desc, np.float64, lx)
    P = o.gaussian_particles((B,), N, seed=3, dtype=np.float64, sigma=[1e-3, 1e-4, 1e-3, 1e-4, 1e-3, 1e-3])
    energy = np.array([6e6, 8e6])
    w_mu, w_cov = rng.normal(size=(B, 6)), rng.normal(size=(B, 6, 6)) * 1e3
    g = lx.grad.track_vjp(lx.Segment(elements), lx.ParticleBeam(P, energy, dtype=np.float64))(
        mu_bar=w_mu, cov_bar=w_cov, wrt_particles=True)
    got = np.asarray(g.particles)
    assert got.shape == (B, N, 7)
    for b, n, c in [(0, 0, 0), (0, 17, 1), (1, 299, 2), (1, 5, 3), (0, 123, 4), (1, 200, 5)]:
        h = 1e-7
        Pp, Pm = P.copy(), P.copy()
        Pp[b, n, c] += h
        Pm[b, n, c] -= h
        ref = (_loss(specs, Pp, energy, w_mu, w_cov)[b] - _loss(specs, Pm, energy, w_mu, w_cov)[b]) / (2 * h)
        assert abs(got[b, n, c] - ref) <= 1e-4 * abs(ref) + 1e-6 * np.max(np.abs(got[b])), (b, n, c, got[b, n, c], ref)
    # directional derivative along a random direction of all particles at once
    D = rng.normal(size=P.shape) * [1e-3, 1e-4, 1e-3, 1e-4, 1e-3, 1e-3, 0]
    h = 1e-5
    ref = (_loss(specs, P + h * D, energy, w_mu, w_cov) - _loss(specs, P - h * D, energy, w_mu, w_cov)) / (2 * h)
    assert np.allclose(np.sum(got * D, axis=(1, 2)), ref, rtol=1e-5)
    with pytest.raises(KeyError):
        lx.grad.track_vjp(lx.Segment(elements), lx.ParticleBeam(P, energy, dtype=np.float64))(mu_bar=w_mu).particles


def test_ea_magnets_gradients_fp32(lx):
    """
    reference tests/test_differentiable.py:31-51: the five magnets of the ARES experimental area
    are the leaves.  Gradient of the beam size on the screen w.r.t. the three quadrupole strengths
    and two corrector angles, float32 kernels, against finite differences of the oracle.
    """
    f = lambda v: np.array([v], dtype=np.float32)  # noqa: E731
    settings = dict(q1=10.0, q2=-9.0, cv=1e-3, q3=7.0, ch=-2e-3)

    def lattice(ns, s, dtype):
        a = lambda v: np.array([v], dtype=dtype)  # noqa: E731
        return [ns.Drift(a(0.17504)), ns.Quadrupole(a(0.122), k1=a(s["q1"])), ns.Drift(a(0.428)),
                ns.Quadrupole(a(0.122), k1=a(s["q2"])), ns.Drift(a(0.204)), ns.VerticalCorrector(a(0.02), angle=a(s["cv"])),
                ns.Drift(a(0.204)), ns.Quadrupole(a(0.122), k1=a(s["q3"])), ns.Drift(a(0.179)),
                ns.HorizontalCorrector(a(0.02), angle=a(s["ch"])), ns.Drift(a(0.45))]

    P = o.gaussian_particles((1,), 20_000, seed=11, dtype=np.float32,
                             sigma=[1.75e-4, 3.7e-6, 1.75e-4, 3.7e-6, 8e-6, 2.3e-3])
    elements = lattice(lx, settings, np.float32)
    w_mu = np.array([[1.0, 0, 1.0, 0, 0, 0]])
    w_cov = np.zeros((1, 6, 6))
    w_cov[0, 0, 0] = w_cov[0, 2, 2] = 1e4  # L = mu_x + mu_y + 1e4 (var x + var y)
    g = lx.grad.track_vjp(lx.Segment(elements), lx.ParticleBeam(P, f(1.07e8)))(mu_bar=w_mu, cov_bar=w_cov)

    def loss(s):
        return _loss(lattice(o, s, np.float64), P.astype(np.float64), np.array([1.07e8]), w_mu, w_cov)[0]

    for key, index, name in (("q1", 1, "k1"), ("q2", 3, "k1"), ("cv", 5, "angle"), ("q3", 7, "k1"), ("ch", 9, "angle")):
        h = 1e-4 * max(abs(settings[key]), 1.0)
        ref = (loss({**settings, key: settings[key] + h}) - loss({**settings, key: settings[key] - h})) / (2 * h)
        got = g[elements[index]][name][0]
        assert abs(got - ref) <= 1e-3 * abs(ref) + 1e-7, (key, got, ref)  # float32 kernels: 1e-3


def test_gradient_based_tuning_example_converges(lx):
    """
    The scenario of the reference's docs/examples/gradientbased.ipynb (examples/
    gradient_based_tuning.py): Adam on three quadrupoles and two correctors of the ARES EA
    section, loss = mse of (mu_x, sigma_x, mu_y, sigma_y) on the screen.  Also checks the
    property cotangents (sigma = sqrt of the unbiased variance) against finite differences.
    """
    import importlib.util
    import pathlib

    spec = importlib.util.spec_from_file_location(
        "gradient_based_tuning", pathlib.Path(__file__).resolve().parents[1] / "examples" / "gradient_based_tuning.py")
    example = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(example)
    f = example.f
    segment = example.ares_ea()
    segment.AREAMQZM1.k1, segment.AREAMQZM2.k1, segment.AREAMQZM3.k1 = f(5.0), f(-5.0), f(5.0)
    segment.AREAMCVM1.angle, segment.AREAMCHM1.angle = f(1e-3), f(-1e-3)
    beam = lx.ParticleBeam.from_parameters(num_particles=20_000, sigma_x=f(1.75e-4), sigma_xp=f(3.7e-6),
                                           sigma_y=f(1.75e-4), sigma_yp=f(3.7e-6), energy=f(1.07e8), seed=0)

    # d sigma_x / d k1 of the first quadrupole: VJP vs central difference of the tracked sigma_x
    g = lx.grad.track_vjp(segment, beam)(sigma_x=1.0)
    got = float(g[segment.AREAMQZM1]["k1"][0])
    h = 0.05
    segment.AREAMQZM1.k1 = f(5.0 + h)
    up = float(segment.track(beam).sigma_x[0])
    segment.AREAMQZM1.k1 = f(5.0 - h)
    down = float(segment.track(beam).sigma_x[0])
    segment.AREAMQZM1.k1 = f(5.0)
    assert np.isclose(got, (up - down) / (2 * h), rtol=2e-2), (got, (up - down) / (2 * h))

    history = example.tune(segment, beam, steps=60)
    assert history[-1] < 0.05 * history[0], (history[0], history[-1])
    assert abs(float(segment.track(beam).mu_x[0])) < 2e-4


def test_parameter_beam_gradients_match_finite_differences_fp64(lx):
    """
    Reverse pass of the ParameterBeam path (lynx_track_moments_backward): every element
    parameter of the cavity lattice, the incoming energy, and the incoming mu and cov (the leaves
    of the reference's tests/test_differentiable.py:54-72) against central differences of the
    oracle's ParameterBeam tracking in float64.
    """
    rng = np.random.default_rng(11)
    B = 2
    desc = _desc(B, rng)
    elements, specs = make_lattice(desc, np.float64, lx)
    A = rng.normal(size=(B, 6, 6)) * [1e-4, 1e-5, 1e-4, 1e-5, 1e-4, 1e-3]
    cov = np.zeros((B, 7, 7))
    cov[:, :6, :6] = A @ np.swapaxes(A, -1, -2)
    mu = np.concatenate([rng.normal(size=(B, 6)) * [1e-3, 1e-4, 1e-3, 1e-4, 1e-4, 1e-3], np.ones((B, 1))], axis=-1)
    energy = np.array([6e6, 8e6])
    w_mu = rng.normal(size=(B, 7)) * [1, 1, 1, 1, 1, 1, 0]
    w_cov = np.zeros((B, 7, 7))
    w_cov[:, :6, :6] = rng.normal(size=(B, 6, 6)) * 1e3

    def loss(mu_, cov_, energy_):
        out = o.segment_track(specs, o.parameter_beam(mu_, cov_, energy_, np.float64), np.float64)
        return np.sum(w_mu * out["mu"], axis=-1) + np.sum(w_cov * out["cov"], axis=(-1, -2))

    beam = lx.ParameterBeam(mu, cov, energy, dtype=np.float64)
    g = lx.grad.track_vjp(lx.Segment(elements), beam)(mu_bar=w_mu, cov_bar=w_cov)

    def central(apply, x0):
        h = 1e-6 * max(abs(x0), 1e-2)
        apply(x0 + h)
        up = loss(mu, cov, energy)
        apply(x0 - h)
        down = loss(mu, cov, energy)
        apply(x0)
        return (up - down) / (2 * h)

    checked = 0
    for e, (kind, _) in enumerate(desc):
        for name in PARAMS_TO_CHECK.get(kind, []):
            arr = specs[e][name]
            if arr is None:
                continue
            got = g[elements[e]][name]
            for idx in np.ndindex(arr.shape):
                def apply(x, arr=arr, idx=idx):
                    arr[idx] = x
                ref = central(apply, arr[idx])[idx[0]]
                scale = max(abs(ref), 1e-9 * np.max(np.abs(w_cov)))
                assert abs(got[idx] - ref) <= 2e-4 * scale + 1e-7 * np.max(np.abs(got)), (kind, e, name, idx, got[idx], ref)
                checked += 1
    assert checked > 50
    # incoming energy, mu and cov
    for bidx in range(B):
        h = 1e-6 * energy[bidx]
        ep, em = energy.copy(), energy.copy()
        ep[bidx] += h
        em[bidx] -= h
        ref = (loss(mu, cov, ep)[bidx] - loss(mu, cov, em)[bidx]) / (2 * h)
        assert abs(g.energy[bidx] - ref) <= 2e-4 * abs(ref) + 1e-12, (bidx, g.energy[bidx], ref)
        for c in range(6):
            h = 1e-7
            mp, mm = mu.copy(), mu.copy()
            mp[bidx, c] += h
            mm[bidx, c] -= h
            ref = (loss(mp, cov, energy)[bidx] - loss(mm, cov, energy)[bidx]) / (2 * h)
            assert abs(g.mu[bidx, c] - ref) <= 1e-5 * abs(ref) + 1e-9 * np.max(np.abs(g.mu[bidx])), (bidx, c, g.mu[bidx, c], ref)
        for (r, c) in [(0, 0), (0, 1), (1, 0), (2, 3), (4, 4), (4, 5), (5, 4), (5, 5), (3, 5)]:
            h = 1e-12
            cp, cm = cov.copy(), cov.copy()
            cp[bidx, r, c] += h
            cm[bidx, r, c] -= h
            ref = (loss(mu, cp, energy)[bidx] - loss(mu, cm, energy)[bidx]) / (2 * h)
            assert abs(g.cov[bidx, r, c] - ref) <= 1e-4 * abs(ref) + 1e-7 * np.max(np.abs(g.cov[bidx])), (bidx, r, c, g.cov[bidx, r, c], ref)


def test_parameter_beam_property_cotangents_and_batches(lx):
    """sigma_x cotangent of a ParameterBeam against a central difference of the tracked sigma_x, fp32, batch of 1000."""
    f = lambda v: np.full(1000, v, np.float32)  # noqa: E731
    k1 = np.linspace(-8, 8, 1000).astype(np.float32)
    seg = lx.Segment([lx.Drift(f(0.2)), lx.Quadrupole(f(0.122), k1=k1, name="Q"), lx.Drift(f(1.0))])
    beam = lx.ParameterBeam.from_parameters(sigma_x=f(1.75e-4), sigma_xp=f(3.7e-6), energy=f(1.07e8))
    g = lx.grad.track_vjp(seg, beam)(sigma_x=np.ones(1000))
    got = g[seg.Q]["k1"]
    assert got.shape == (1000,)
    h = 0.05
    seg.Q.k1 = k1 + np.float32(h)
    up = seg.track(beam).sigma_x.astype(np.float64)
    seg.Q.k1 = k1 - np.float32(h)
    down = seg.track(beam).sigma_x.astype(np.float64)
    ref = (up - down) / (2 * h)
    assert np.allclose(got, ref, rtol=5e-2, atol=2e-3 * np.max(np.abs(ref)))


def test_config_5_gradients_at_the_full_shape(lx, monkeypatch):
    """
    BASELINE config 5's gradient half at ITS shape -- what `bench.py --workload c5 --grad` times: 4096 environments x
    10 000 particles x [Drift, misaligned Quadrupole, Drift, Cavity] x 8, float32, the bench's lattice and beam
    (`bench.describe("c5")`, `bench.BEAM_SIGMA`) -- i.e. k_track_bwd_units' launch geometry, exchange buffer and
    k_reduce_tbar over 4096 samples, which the small-shape tests above do not reach.  Cotangent: random weights on all
    six means and the whole covariance.  EVERY parameter of EVERY environment (k1, misalignment, lengths, voltage,
    phase, frequency, incoming energy) of
    (1) the structured reverse pass (default) against the dense one (LYNX_BWD_UNITS=0), and
    (2) both against the float64 reverse pass on the same particles and parameters (cast up), which
        test_gradients_match_finite_differences_fp64 pins to finite differences of the oracle.
    Distance of a gradient array g from its reference r: max over the batch of |g - r| / (|r| + 1e-3 max |r|).

    Measured on MI355X (this test prints the table; round 4), asserted at twice the measured value:

        (the table sits next to TOL_C5_GRAD below, and in DESIGN.md section 2)

    History of the cavity rows: as the round began phase stood at 1.2e-2, voltage 2.7e-3, frequency 4.6e-3 from the
    float64 pass and phase 1.3e-2 between the two float32 kernels.  Three cancellations were removed: the kick's
    sin(a) - sin(phi) and cos(a) - cos(phi) in the derivatives, summed over the particles as separate large numbers
    (lynx_grad.hpp: kick_cotangents); the bracket of r55_cor (cavity.py:296-305) in the dual-number builders
    (lynx_dual.hpp: cavity_r55_bracket_dual); and last cos(a) - cos(phi) in the FORWARD kick itself
    (lynx_device.hpp: cos_difference), whose rounding the recomputed forward states had carried into every gradient.
    """
    import bench

    dtype = np.float32
    batch, particles, cells, _, _ = bench.WORKLOADS["c5"]
    ids = np.arange(batch)
    desc = bench.describe("c5", ids, cells, dtype, seed=3)
    rng = np.random.default_rng(17)
    w_mu = rng.normal(size=(batch, 6))
    w_cov = rng.normal(size=(batch, 6, 6)) * 1e3
    names = {"drift": ["length"], "quadrupole": ["length", "k1", "misalignment"], "cavity": ["length", "voltage", "phase", "frequency"]}

    def gradients(desc_, beam):
        elements, _ = make_lattice(desc_, beam.dtype.type, lx)
        g = lx.grad.track_vjp(lx.Segment(elements), beam)(mu_bar=w_mu, cov_bar=w_cov)
        got = {(e, name): np.asarray(g[elements[e]][name], dtype=np.float64)
               for e, (kind, _) in enumerate(desc_) for name in names[kind]}
        got["energy"] = np.asarray(g.energy, dtype=np.float64)
        return got

    def distance(g, r):
        return float(np.max(np.abs(g - r) / (np.abs(r) + 1e-3 * np.max(np.abs(r)) + 1e-300)))

    beam = lx.ParticleBeam.synthetic((batch,), particles, sigma=bench.BEAM_SIGMA, energy=bench.beam_energy("c5"), seed=2, dtype=dtype)
    structured = gradients(desc, beam)
    monkeypatch.setenv("LYNX_BWD_UNITS", "0")
    dense = gradients(desc, beam)
    monkeypatch.delenv("LYNX_BWD_UNITS")
    assert any(np.any(structured[key] != dense[key]) for key in structured)  # two different kernels did run
    desc64 = [(kind, {k: np.asarray(v).astype(np.float64) for k, v in kw.items()}) for kind, kw in desc]
    beam64 = lx.ParticleBeam(np.asarray(beam.particles).astype(np.float64), np.full(batch, bench.beam_energy("c5")), dtype=np.float64)
    g64 = gradients(desc64, beam64)
    del beam64
    worst = {}
    for key, ref in g64.items():
        name = key if isinstance(key, str) else key[1]
        assert structured[key].shape == dense[key].shape == ref.shape and np.all(np.isfinite(structured[key])), key
        row = worst.setdefault(name, [0.0, 0.0, 0.0])
        row[0] = max(row[0], distance(structured[key], dense[key]))
        row[1] = max(row[1], distance(structured[key], ref))
        row[2] = max(row[2], distance(dense[key], ref))
    print(f"{'parameter':>14} {'structured-dense':>18} {'structured-float64':>20} {'dense-float64':>16}")
    for name, row in worst.items():
        print(f"{name:>14} {row[0]:18.1e} {row[1]:20.1e} {row[2]:16.1e}")
    for name, row in worst.items():
        assert row[0] <= TOL_C5_GRAD[name][0], (name, row)
        assert max(row[1], row[2]) <= TOL_C5_GRAD[name][1], (name, row)


# (float32 kernels against each other, float32 against the float64 pass): twice what the test above measured on MI355X,
# over all 4096 environments (round 4):          structured - dense    structured - float64    dense - float64
#                                  length              2.7e-05               3.4e-05               3.6e-05
#                                  k1                  5.3e-05               6.1e-05               4.9e-05
#                                  misalignment        3.4e-05               9.9e-05               1.0e-04
#                                  voltage             2.0e-05               1.0e-04               1.0e-04
#                                  phase               1.7e-05               7.6e-05               7.6e-05
#                                  frequency           2.2e-05               2.9e-04               2.7e-04
#                                  energy              9.7e-06               1.8e-05               1.8e-05
# (with the forward kick still subtracting two float32 cosines -- until late in round 4 -- the float64 columns read
# voltage 8.9e-4, phase 2.0e-3, frequency 8.4e-3, energy 9.6e-5: the reverse pass differentiates the forward states it
# recomputes, and those carried the cosines' rounding)
# (structured - dense: the structured kernel keeps a lane's share of the transverse sums S_x, S_y in float32 over all of
# its tiles before the workgroup adds them up -- with round 3's exchange-buffer form, which added every tile's products
# over the wave at once, that column read 4e-06 .. 2e-05)
TOL_C5_GRAD = {"length": (6e-5, 8e-5), "k1": (1.1e-4, 1.3e-4), "misalignment": (7e-5, 2e-4), "voltage": (4e-5, 2.1e-4),
               "phase": (3.5e-5, 1.6e-4), "frequency": (4.5e-5, 6e-4), "energy": (2e-5, 4e-5)}


def _ares_with_active_bpms(ns, dtype, values, active=True):
    """The README's ARES segment (BASELINE configs 1 and 2) with its second and last BPM switched on."""
    a = lambda v: np.array([v], dtype=dtype)  # noqa: E731
    bpm = (lambda on: ns.BPM(is_active=on)) if ns is not o else (lambda on: o.BPM(is_active=on))
    return [bpm(False), ns.Drift(a(values["d1"])), bpm(active), ns.Drift(a(1.0)),
            ns.VerticalCorrector(a(0.3), angle=a(values["v7"])), ns.Drift(a(0.2)),
            ns.HorizontalCorrector(a(values["lh"]), angle=a(values["h10"])), ns.Drift(a(7.0)),
            ns.HorizontalCorrector(a(0.3), angle=a(values["h12"])), ns.Drift(a(0.05)), bpm(active)]


@pytest.mark.parametrize("beam_type", ["particles-fp64", "particles-fp32", "parameters-fp64"])
def test_gradients_through_active_bpms(lx, beam_type):
    """
    BASELINE config 2's lattice with two ACTIVE BPMs (bpm.py:48-58: an active BPM is a step of `Segment.track` of its
    own and records `stack([mu_x, mu_y])` of the beam that enters it).  The loss sees the outgoing beam's moments AND
    both readings; its gradient w.r.t. corrector angles and lengths against central finite differences of the oracle
    (float64), for a ParticleBeam -- the BPMs are observer steps inside the one streaming pass, the cotangent of a
    reading is one more term of every particle's cotangent at that point of the reverse sweep -- and for a
    ParameterBeam, where `track` cuts the lattice at the BPMs and the reverse pass is chained stretch by stretch.
    """
    dtype = np.float32 if beam_type.endswith("fp32") else np.float64
    values = dict(d1=1.0, v7=3.142e-3, lh=0.3, h10=1e-4, h12=-1e-4)
    rng = np.random.default_rng(5)
    w_mu = rng.normal(size=(1, 6))
    w_cov = np.zeros((1, 6, 6))
    w_cov[0, :4, :4] = rng.normal(size=(4, 4)) * 1e3
    r_a, r_b = rng.normal(size=(2, 1)), rng.normal(size=(2, 1))
    N = 4000
    P = o.gaussian_particles((1,), N, seed=21, dtype=np.float64, sigma=[1e-4, 2e-5, 1e-4, 2e-5, 1e-5, 1e-3],
                             mu=[2e-4, 3e-5, -1e-4, 2e-5, 0.0, 0.0])
    energy = np.array([1e8])
    if beam_type.startswith("parameters"):
        Q = P[0, :, :6]
        mu0 = np.concatenate([Q.mean(axis=0), [1.0]])[None]
        cov0 = np.zeros((1, 7, 7))
        cov0[0, :6, :6] = np.cov(Q.T, bias=True)

    def loss(v):
        readings = []
        specs = _ares_with_active_bpms(o, np.float64, v)
        if beam_type.startswith("parameters"):
            out = o.segment_track(specs, o.parameter_beam(mu0, cov0, energy, np.float64), np.float64, bpm_readings=readings)
            mu, cov = out["mu"][..., :6], out["cov"][..., :6, :6]
        else:
            out = o.segment_track(specs, o.particle_beam(P, energy, np.float64), np.float64, bpm_readings=readings)
            Q = out["particles"][..., :6]
            mu = Q.mean(axis=-2)
            d = Q - mu[..., None, :]
            cov = np.einsum("...ni,...nj->...ij", d, d) / N
        (_, ra), (_, rb) = readings  # in lattice order (the oracle finds an element's index by equality: both BPMs "at 2")
        return float(np.sum(w_mu * mu) + np.sum(w_cov * cov) + np.sum(r_a * ra) + np.sum(r_b * rb))

    elements = _ares_with_active_bpms(lx, dtype, values)
    segment = lx.Segment(elements)
    if beam_type.startswith("parameters"):
        beam = lx.ParameterBeam(mu0, cov0, energy, dtype=dtype)
    else:
        beam = lx.ParticleBeam(P.astype(dtype), energy.astype(dtype), dtype=dtype)
    vjp = lx.grad.track_vjp(segment, beam)
    # the forward pass recorded the readings the loss is written in
    readings = []
    o.segment_track(_ares_with_active_bpms(o, np.float64, values), o.particle_beam(P, energy, np.float64), np.float64, bpm_readings=readings)
    tol_read = 1e-4 if dtype == np.float32 else 1e-7
    assert np.allclose(elements[2].reading, readings[0][1], rtol=tol_read, atol=1e-12)
    assert np.allclose(elements[10].reading, readings[1][1], rtol=tol_read, atol=1e-12)
    wc7 = np.zeros((1, 7, 7))
    wc7[:, :6, :6] = w_cov
    g = vjp(mu_bar=w_mu, cov_bar=wc7 if beam_type.startswith("parameters") else w_cov, readings={elements[2]: r_a, elements[10]: r_b})
    rtol = 2e-3 if dtype == np.float32 else 2e-5
    for key, index, name in (("d1", 1, "length"), ("v7", 4, "angle"), ("lh", 6, "length"), ("h10", 6, "angle"), ("h12", 8, "angle")):
        h = 1e-5 * max(abs(values[key]), 1e-2)
        ref = (loss({**values, key: values[key] + h}) - loss({**values, key: values[key] - h})) / (2 * h)
        got = float(np.asarray(g[elements[index]][name]).reshape(-1)[0])
        assert abs(got - ref) <= rtol * abs(ref) + 1e-9, (key, got, ref)
    # without the readings' cotangents the gradient is a different one: the BPM terms did go in
    plain = vjp(mu_bar=w_mu, cov_bar=wc7 if beam_type.startswith("parameters") else w_cov)
    assert not np.isclose(float(np.asarray(plain[elements[1]]["length"]).reshape(-1)[0]),
                          float(np.asarray(g[elements[1]]["length"]).reshape(-1)[0]), rtol=1e-3)
    with pytest.raises(KeyError):
        vjp(mu_bar=w_mu, readings={elements[0]: r_a})  # an inactive BPM reads nothing
