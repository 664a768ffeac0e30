"""
CPU oracle for the lynx `Segment.track` hot path -- TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement, function by function, of the reference algorithm
in /root/reference (jank324/lynx).  It is the *checker* for the HIP kernels in
`lynx_amd/csrc/`.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import it.  Nothing under `lynx_amd/` imports it, and the
product path has no CPU fallback.

Why a restatement and not the reference itself: the reference is an unfinished
torch->jax port (`README.md:13`) that imports `jax`/`equinox` (absent here, no
network) and is not executable JAX even with them (`nn.ModuleList` with no `nn`
import at `lynx/accelerator/segment.py:43`, in-place scatter into immutable arrays at
`lynx/track_methods.py:25-32`, ...).  Its arithmetic is fully legible, so every
function below cites the reference lines it follows and keeps the reference's
operation order.  torch idioms are read with torch semantics (`Tensor.repeat`,
in-place scatter, `std` unbiased, `any()` over the whole batch).

Pinning: the restatement is checked in `tests/test_oracle_kat.py` against the
known answers the reference holds for this path:
KAT-1 `try_batched.ipynb` (ParameterBeam x Drift), KAT-2
`docs/examples/optimize_speed.ipynb` (1051-map composition), KAT-3
`tests/test_vectorized.py:371-392` (CustomTransferMap), KAT-4
`tests/test_compare_ocelot.py:627-654` (Cavity, Bmad-confirmed Twiss), KAT-5
`tests/test_parameter_beam.py`.  Tracked-particle values have no absolute fixture in
the reference (its particle tests compare against Ocelot, which is absent, on an ASTRA
file that is missing: `.MISSING_LARGE_BLOBS:1`), so particle-level parity is
"pinned through the maps": the 7x7 maps are pinned by the KATs and the particle
update is the single matmul of `lynx/accelerator/element.py:85`.

Conventions
-----------
* State vector (x, x', y, y', s, delta, 1) (`lynx/accelerator/element.py:41-54`).
* A lattice is a list of element *specs*: dicts with key "kind" and array-valued
  parameters of shape `batch_shape` (or broadcastable to it).
* All arithmetic is carried out in `dtype` (np.float32 or np.float64).
"""

from __future__ import annotations

import numpy as np
from scipy import constants
from scipy.constants import physical_constants

# lynx/track_methods.py:9-11
REST_ENERGY = (
    constants.electron_mass * constants.speed_of_light**2 / constants.elementary_charge
)
# lynx/accelerator/cavity.py:20
ELECTRON_MASS_EV = physical_constants["electron mass energy equivalent in MeV"][0] * 1e6
SPEED_OF_LIGHT = constants.speed_of_light

SKIPPABLE_ALWAYS = (
    "solenoid",
    "undulator",
    "drift",
    "quadrupole",
    "dipole",
    "rbend",
    "hcor",
    "vcor",
    "custom",
    "marker",
)


def _a(x, dtype):
    return np.asarray(x, dtype=dtype)


def _eye(batch_shape, dtype):
    """`jnp.eye(7).repeat(*shape, 1, 1)` with torch semantics: one I7 per sample."""
    tm = np.zeros((*batch_shape, 7, 7), dtype=dtype)
    for i in range(7):
        tm[..., i, i] = 1
    return tm


def _matmul(a, b):
    """Batched 7x7 product in the operands' dtype (no silent upcast)."""
    return np.matmul(a, b)


# ---------------------------------------------------------------------------------
# lynx/track_methods.py
# ---------------------------------------------------------------------------------


def rotation_matrix(angle):
    """lynx/track_methods.py:14-34."""
    angle = np.asarray(angle)
    cs = np.cos(angle)
    sn = np.sin(angle)
    tm = _eye(angle.shape, angle.dtype)
    tm[..., 0, 0] = cs
    tm[..., 0, 2] = sn
    tm[..., 1, 1] = cs
    tm[..., 1, 3] = sn
    tm[..., 2, 0] = -sn
    tm[..., 2, 2] = cs
    tm[..., 3, 1] = -sn
    tm[..., 3, 3] = cs
    return tm


def base_rmatrix(length, k1, hx, tilt=None, energy=None):
    """lynx/track_methods.py:37-105."""
    length = np.asarray(length)
    dtype = length.dtype
    cdtype = np.complex64 if dtype == np.float32 else np.complex128
    k1 = np.asarray(k1, dtype=dtype)
    hx = np.asarray(hx, dtype=dtype)
    tilt = np.zeros_like(length) if tilt is None else np.asarray(tilt, dtype=dtype)
    energy = np.zeros_like(length) if energy is None else np.asarray(energy, dtype=dtype)
    shape = np.broadcast_shapes(length.shape, k1.shape, hx.shape, tilt.shape, energy.shape)
    length, k1, hx, tilt, energy = (
        np.broadcast_to(v, shape) for v in (length, k1, hx, tilt, energy)
    )

    with np.errstate(all="ignore"):
        gamma = energy / dtype.type(REST_ENERGY)  # :60
        igamma2 = np.ones(shape, dtype=dtype)  # :61  (NB: ones here, zeros in Drift)
        nz = gamma != 0
        igamma2[nz] = 1 / gamma[nz] ** 2  # :62
        beta = np.sqrt(1 - igamma2)  # :64

        k1 = k1.copy()  # :67
        k1[k1 == 0] = 1e-12  # :68

        kx2 = k1 + hx**2  # :70
        ky2 = -k1  # :71
        kx = np.sqrt(kx2.astype(cdtype))  # :72
        ky = np.sqrt(ky2.astype(cdtype))  # :73
        cx = np.cos(kx * length).real  # :74
        cy = np.cos(ky * length).real  # :75
        sy = length.copy()  # :76
        m = ky != 0
        sy[m] = (np.sin(ky[m] * length[m]) / ky[m]).real  # :77

        sx = (np.sin(kx * length) / kx).real  # :79
        dx = hx / kx2 * (1.0 - cx)  # :80
        r56 = hx**2 * (length - sx) / kx2 / beta**2  # :81
        r56 = r56 - length / beta**2 * igamma2  # :83

        R = _eye(shape, dtype)  # :85
        R[..., 0, 0] = cx
        R[..., 0, 1] = sx
        R[..., 0, 5] = dx / beta
        R[..., 1, 0] = -kx2 * sx
        R[..., 1, 1] = cx
        R[..., 1, 5] = sx * hx / beta
        R[..., 2, 2] = cy
        R[..., 2, 3] = sy
        R[..., 3, 2] = -ky2 * sy
        R[..., 3, 3] = cy
        R[..., 4, 0] = sx * hx / beta
        R[..., 4, 1] = dx / beta
        R[..., 4, 5] = r56

        if np.any(tilt != 0):  # :101  -- decided on the WHOLE batch
            # einsum("...ij,...jk,...kl->...il", A, B, C) read as (A.B).C
            R = _matmul(_matmul(rotation_matrix(-tilt), R), rotation_matrix(tilt))
    return R.astype(dtype)


def misalignment_matrix(misalignment):
    """lynx/track_methods.py:108-122.  Returns (R_entry, R_exit)."""
    misalignment = np.asarray(misalignment)
    dtype = misalignment.dtype
    batch_shape = misalignment.shape[:-1]
    R_exit = _eye(batch_shape, dtype)
    R_exit[..., 0, 6] = misalignment[..., 0]
    R_exit[..., 2, 6] = misalignment[..., 1]
    R_entry = _eye(batch_shape, dtype)
    R_entry[..., 0, 6] = -misalignment[..., 0]
    R_entry[..., 2, 6] = -misalignment[..., 1]
    return R_entry, R_exit


# ---------------------------------------------------------------------------------
# Element specs
# ---------------------------------------------------------------------------------


def Drift(length):
    return {"kind": "drift", "length": length}


def Quadrupole(length, k1=None, misalignment=None, tilt=None):
    return {"kind": "quadrupole", "length": length, "k1": k1, "misalignment": misalignment, "tilt": tilt}


def Dipole(length, angle=None, e1=None, e2=None, tilt=None, fringe_integral=None,
           fringe_integral_exit=None, gap=None):
    return {"kind": "dipole", "length": length, "angle": angle, "e1": e1, "e2": e2, "tilt": tilt,
            "fringe_integral": fringe_integral, "fringe_integral_exit": fringe_integral_exit,
            "gap": gap}


def RBend(length, angle=None, e1=None, e2=None, **kw):
    """lynx/accelerator/rbend.py:79-80: e1 += angle/2, e2 += angle/2."""
    spec = Dipole(length, angle=angle, e1=e1, e2=e2, **kw)
    spec["kind"] = "rbend"
    return spec


def HorizontalCorrector(length, angle=None):
    return {"kind": "hcor", "length": length, "angle": angle}


def VerticalCorrector(length, angle=None):
    return {"kind": "vcor", "length": length, "angle": angle}


def Cavity(length, voltage=None, phase=None, frequency=None):
    return {"kind": "cavity", "length": length, "voltage": voltage, "phase": phase,
            "frequency": frequency}


def CustomTransferMap(transfer_map, length=None):
    return {"kind": "custom", "transfer_map": transfer_map, "length": length}


def Solenoid(length, k=None, misalignment=None):
    return {"kind": "solenoid", "length": length, "k": k, "misalignment": misalignment}


def Undulator(length):
    return {"kind": "undulator", "length": length}


def BPM(is_active=False):
    return {"kind": "bpm", "is_active": is_active}


def Marker():
    return {"kind": "marker"}


def _p(spec, key, like, dtype):
    """Parameter `key` of `spec` as dtype array; missing -> zeros_like(like)."""
    v = spec.get(key)
    if v is None:
        return np.zeros_like(like)
    return np.asarray(v, dtype=dtype)


def is_skippable(spec) -> bool:
    """`Element.is_skippable` of each kind."""
    kind = spec["kind"]
    if kind in SKIPPABLE_ALWAYS:
        return True
    if kind == "cavity":  # cavity.py:64-70: not any(voltage != 0)
        v = spec.get("voltage")
        return not (v is not None and np.any(np.asarray(v) != 0))
    if kind == "bpm":  # bpm.py:39-41
        return not spec.get("is_active", False)
    raise ValueError(kind)


# ---------------------------------------------------------------------------------
# Per-element transfer maps
# ---------------------------------------------------------------------------------


def _drift_like_map(length, energy, dtype):
    """lynx/accelerator/drift.py:44-62 (igamma2 defaults to ZERO where gamma == 0)."""
    with np.errstate(all="ignore"):
        gamma = energy / dtype.type(REST_ENERGY)
        igamma2 = np.zeros_like(gamma)
        nz = gamma != 0
        igamma2[nz] = 1 / gamma[nz] ** 2
        beta = np.sqrt(1 - igamma2)
        shape = np.broadcast_shapes(length.shape, energy.shape)
        tm = _eye(shape, dtype)
        tm[..., 0, 1] = length
        tm[..., 2, 3] = length
        tm[..., 4, 5] = -length / beta**2 * igamma2
    return tm


def _dipole_edge(hx, e, fint, gap, dtype):
    """lynx/accelerator/dipole.py:143-181 (_transfer_map_enter/_exit share one form)."""
    with np.errstate(all="ignore"):
        sec_e = 1.0 / np.cos(e)
        phi = fint * hx * gap * sec_e * (1 + np.sin(e) ** 2)
        tm = _eye(phi.shape, dtype)
        tm[..., 1, 0] = hx * np.tan(e)
        tm[..., 3, 2] = -hx * np.tan(e - phi)
    return tm


def cavity_rmatrix(spec, energy, dtype):
    """lynx/accelerator/cavity.py:248-325."""
    dtype = np.dtype(dtype)
    length = np.asarray(spec["length"], dtype=dtype)
    voltage = _p(spec, "voltage", length, dtype)
    phase = _p(spec, "phase", length, dtype)
    frequency = _p(spec, "frequency", length, dtype)
    energy = np.asarray(energy, dtype=dtype)
    shape = np.broadcast_shapes(length.shape, voltage.shape, phase.shape, frequency.shape,
                                energy.shape)
    length, voltage, phase, frequency, energy = (
        np.broadcast_to(v, shape) for v in (length, voltage, phase, frequency, energy)
    )
    me = dtype.type(ELECTRON_MASS_EV)
    with np.errstate(all="ignore"):
        phi = np.deg2rad(phase)
        delta_energy = voltage * np.cos(phi)
        eta = dtype.type(1.0)
        Ei = energy / me
        Ef = (energy + delta_energy) / me
        Ep = (Ef - Ei) / length
        assert np.all(Ei > 0), "Initial energy must be larger than 0"  # :260

        alpha = np.sqrt(eta / 8) / np.cos(phi) * np.log(Ef / Ei)
        r11 = np.cos(alpha) - np.sqrt(2 / eta) * np.cos(phi) * np.sin(alpha)
        r12 = np.sqrt(8 / eta) * Ei / Ep * np.cos(phi) * np.sin(alpha)
        r21 = (
            -Ep / Ef
            * (np.cos(phi) / np.sqrt(2 * eta) + np.sqrt(eta / 8) / np.cos(phi))
            * np.sin(alpha)
        )
        r22 = Ei / Ef * (np.cos(alpha) + np.sqrt(2 / eta) * np.cos(phi) * np.sin(alpha))

        r56 = dtype.type(0.0)
        beta0 = dtype.type(1.0)
        beta1 = dtype.type(1.0)
        k = 2 * dtype.type(np.pi) * frequency / dtype.type(SPEED_OF_LIGHT)
        r55_cor = dtype.type(0.0)
        if np.any((voltage != 0) & (energy != 0)):  # :290 -- whole batch
            beta0 = np.sqrt(1 - 1 / Ei**2)
            beta1 = np.sqrt(1 - 1 / Ef**2)
            r56 = -length / (Ef**2 * Ei * beta1) * (Ef + Ei) / (beta1 + beta0)
            g0 = Ei
            g1 = Ef
            r55_cor = (
                k * length * beta0 * voltage / me * np.sin(phi)
                * (g0 * g1 * (beta0 * beta1 - 1) + 1)
                / (beta1 * g1 * (g0 - g1) ** 2)
            )
        r66 = Ei / Ef * beta0 / beta1
        r65 = k * np.sin(phi) * voltage / (Ef * beta1 * me)

        R = _eye(shape, dtype)
        R[..., 0, 0] = r11
        R[..., 0, 1] = r12
        R[..., 1, 0] = r21
        R[..., 1, 1] = r22
        R[..., 2, 2] = r11
        R[..., 2, 3] = r12
        R[..., 3, 2] = r21
        R[..., 3, 3] = r22
        R[..., 4, 4] = 1 + r55_cor
        R[..., 4, 5] = r56
        R[..., 5, 4] = r65
        R[..., 5, 5] = r66
    return R.astype(dtype)


def element_transfer_map(spec, energy, dtype=np.float32):
    """`Element.transfer_map(energy)` for every kind on the path -> (*batch, 7, 7)."""
    dtype = np.dtype(dtype)
    energy = np.asarray(energy, dtype=dtype)
    kind = spec["kind"]

    if kind in ("marker", "bpm"):  # marker.py:32-35, bpm.py:43-46
        return _eye(energy.shape, dtype)

    if kind == "custom":  # custom_transfer_map.py:87-88
        return np.asarray(spec["transfer_map"], dtype=dtype)

    length = np.asarray(spec["length"], dtype=dtype)

    if kind == "drift":
        assert energy.shape == length.shape, (  # drift.py:45-47
            f"Beam shape {energy.shape} does not match element shape {length.shape}"
        )
        return _drift_like_map(length, energy, dtype)

    if kind in ("hcor", "vcor"):  # horizontal_corrector.py:52-67, vertical_corrector.py:52-66
        angle = _p(spec, "angle", length, dtype)
        tm = _drift_like_map(length, energy, dtype)
        tm[..., 1 if kind == "hcor" else 3, 6] = angle
        return tm

    if kind == "quadrupole":  # quadrupole.py:66-80
        k1 = _p(spec, "k1", length, dtype)
        tilt = _p(spec, "tilt", length, dtype)
        mis = spec.get("misalignment")
        mis = (np.zeros((*length.shape, 2), dtype=dtype) if mis is None
               else np.asarray(mis, dtype=dtype))
        R = base_rmatrix(length, k1, np.zeros_like(length), tilt, energy)
        if np.all(mis == 0):
            return R
        R_entry, R_exit = misalignment_matrix(mis)
        return _matmul(_matmul(R_exit, R), R_entry)

    if kind in ("dipole", "rbend"):  # dipole.py:96-181, rbend.py:79-80
        angle = _p(spec, "angle", length, dtype)
        e1 = _p(spec, "e1", length, dtype)
        e2 = _p(spec, "e2", length, dtype)
        tilt = _p(spec, "tilt", length, dtype)
        fint = _p(spec, "fringe_integral", length, dtype)
        fintx = spec.get("fringe_integral_exit")
        fintx = fint if fintx is None else np.asarray(fintx, dtype=dtype)
        gap = _p(spec, "gap", length, dtype)
        if kind == "rbend":
            e1 = e1 + angle / 2
            e2 = e2 + angle / 2
        with np.errstate(all="ignore"):
            hx = np.zeros_like(length)  # dipole.py:96-102
            nz = length != 0
            hx[nz] = (np.broadcast_to(angle, length.shape)[nz] / length[nz])
            R_enter = _dipole_edge(hx, e1, fint, gap, dtype)
            R_exit = _dipole_edge(hx, e2, fintx, gap, dtype)
            if np.any(length != 0.0):  # dipole.py:119 -- whole batch
                R = base_rmatrix(length, np.zeros_like(length), hx, np.zeros_like(length), energy)
            else:  # thin corrector, dipole.py:127-133
                R = _eye(length.shape, dtype)
                R[..., 0, 1] = length
                R[..., 2, 6] = angle
                R[..., 2, 3] = length
            R = _matmul(R_exit, _matmul(R, R_enter))  # :136
            R = _matmul(rotation_matrix(-tilt), _matmul(R, rotation_matrix(tilt)))  # :138-140
        return R.astype(dtype)

    if kind == "cavity":  # cavity.py:72-79
        return cavity_rmatrix(spec, energy, dtype)

    if kind == "undulator":  # undulator.py:48-60
        with np.errstate(all="ignore"):
            gamma = energy / dtype.type(REST_ENERGY)
            igamma2 = np.zeros_like(gamma)
            nz = gamma != 0
            igamma2[nz] = 1 / gamma[nz] ** 2
            tm = _eye(np.broadcast_shapes(length.shape, energy.shape), dtype)
            tm[..., 0, 1] = length
            tm[..., 2, 3] = length
            tm[..., 4, 5] = length * igamma2
        return tm

    if kind == "solenoid":  # solenoid.py:61-105 (`if gamma != 0` read per sample)
        k = _p(spec, "k", length, dtype)
        mis = spec.get("misalignment")
        mis = (np.zeros((*length.shape, 2), dtype=dtype) if mis is None else np.asarray(mis, dtype=dtype))
        with np.errstate(all="ignore"):
            gamma = np.broadcast_to(energy / dtype.type(REST_ENERGY), length.shape)
            c = np.cos(length * k)
            s = np.sin(length * k)
            s_k = length.copy()
            nz = k != 0
            s_k[nz] = s[nz] / k[nz]
            r56 = np.zeros_like(length)
            g = gamma != 0
            gamma2 = gamma[g] * gamma[g]
            beta = np.sqrt(1.0 - 1.0 / gamma2)
            r56[g] -= length[g] / (beta * beta * gamma2)
            R = _eye(length.shape, dtype)
            R[..., 0, 0] = c**2
            R[..., 0, 1] = c * s_k
            R[..., 0, 2] = s * c
            R[..., 0, 3] = s * s_k
            R[..., 1, 0] = -k * s * c
            R[..., 1, 1] = c**2
            R[..., 1, 2] = -k * s**2
            R[..., 1, 3] = s * c
            R[..., 2, 0] = -s * c
            R[..., 2, 1] = -s * s_k
            R[..., 2, 2] = c**2
            R[..., 2, 3] = c * s_k
            R[..., 3, 0] = k * s**2
            R[..., 3, 1] = -s * c
            R[..., 3, 2] = -k * s * c
            R[..., 3, 3] = c**2
            R[..., 4, 5] = r56
        if np.all(mis == 0):
            return R
        R_entry, R_exit = misalignment_matrix(mis)
        return _matmul(_matmul(R_exit, R), R_entry)

    raise ValueError(f"unknown element kind {kind!r}")


# ---------------------------------------------------------------------------------
# Beams (plain dict containers)
# ---------------------------------------------------------------------------------


def parameter_beam(mu, cov, energy, dtype=np.float32):
    return {"type": "parameter", "mu": _a(mu, dtype), "cov": _a(cov, dtype),
            "energy": _a(energy, dtype)}


def particle_beam(particles, energy, dtype=np.float32):
    particles = _a(particles, dtype)
    assert particles.shape[-2] > 0 and particles.shape[-1] == 7, (
        "Particle vectors must be 7-dimensional."  # particle_beam.py:35-37
    )
    return {"type": "particle", "particles": particles, "energy": _a(energy, dtype)}


def parameter_beam_from_parameters(dtype=np.float32, **kw):
    """lynx/particles/parameter_beam.py:47-144."""
    dtype = np.dtype(dtype)
    given = [np.asarray(v) for v in kw.values() if v is not None]
    shape = given[0].shape if given else (1,)
    assert all(g.shape == shape for g in given), "Arguments must have the same shape."

    def g(name, default):
        v = kw.get(name)
        return np.full(shape, default, dtype=dtype) if v is None else np.asarray(v, dtype=dtype)

    mu_x, mu_xp, mu_y, mu_yp = g("mu_x", 0.0), g("mu_xp", 0.0), g("mu_y", 0.0), g("mu_yp", 0.0)
    sigma_x, sigma_xp = g("sigma_x", 175e-9), g("sigma_xp", 2e-7)
    sigma_y, sigma_yp = g("sigma_y", 175e-9), g("sigma_yp", 2e-7)
    sigma_s, sigma_p = g("sigma_s", 1e-6), g("sigma_p", 1e-6)
    cor_x, cor_y, cor_s = g("cor_x", 0.0), g("cor_y", 0.0), g("cor_s", 0.0)
    energy = g("energy", 1e8)

    mu = np.stack([mu_x, mu_xp, mu_y, mu_yp, np.zeros(shape, dtype), np.zeros(shape, dtype),
                   np.ones(shape, dtype)], axis=-1)
    cov = np.zeros((*shape, 7, 7), dtype=dtype)
    cov[..., 0, 0] = sigma_x**2
    cov[..., 0, 1] = cor_x
    cov[..., 1, 0] = cor_x
    cov[..., 1, 1] = sigma_xp**2
    cov[..., 2, 2] = sigma_y**2
    cov[..., 2, 3] = cor_y
    cov[..., 3, 2] = cor_y
    cov[..., 3, 3] = sigma_yp**2
    cov[..., 4, 4] = sigma_s**2
    cov[..., 4, 5] = cor_s
    cov[..., 5, 4] = cor_s
    cov[..., 5, 5] = sigma_p**2
    return parameter_beam(mu, cov, energy, dtype)


def parameter_beam_from_twiss(dtype=np.float32, **kw):
    """lynx/particles/parameter_beam.py:146-232."""
    dtype = np.dtype(dtype)
    given = [np.asarray(v) for v in kw.values() if v is not None]
    shape = given[0].shape if given else (1,)
    assert all(g.shape == shape for g in given), "Arguments must have the same shape."

    def g(name, default):
        v = kw.get(name)
        return np.full(shape, default, dtype=dtype) if v is None else np.asarray(v, dtype=dtype)

    beta_x, alpha_x, emittance_x = g("beta_x", 1.0), g("alpha_x", 0.0), g("emittance_x", 7.1971891e-13)
    beta_y, alpha_y, emittance_y = g("beta_y", 1.0), g("alpha_y", 0.0), g("emittance_y", 7.1971891e-13)
    assert np.all(beta_x > 0), "Beta function in x direction must be larger than 0 everywhere."
    assert np.all(beta_y > 0), "Beta function in y direction must be larger than 0 everywhere."
    sigma_x = np.sqrt(emittance_x * beta_x)
    sigma_xp = np.sqrt(emittance_x * (1 + alpha_x**2) / beta_x)
    sigma_y = np.sqrt(emittance_y * beta_y)
    sigma_yp = np.sqrt(emittance_y * (1 + alpha_y**2) / beta_y)
    cor_x = -emittance_x * alpha_x
    cor_y = -emittance_y * alpha_y
    return parameter_beam_from_parameters(
        dtype=dtype, sigma_x=sigma_x, sigma_xp=sigma_xp, sigma_y=sigma_y, sigma_yp=sigma_yp,
        sigma_s=g("sigma_s", 1e-6), sigma_p=g("sigma_p", 1e-6), energy=g("energy", 1e8),
        cor_s=g("cor_s", 0.0), cor_x=cor_x, cor_y=cor_y,
    )


def beam_moments(beam, ddof=1):
    """
    Read-out of every moment property.

    ParameterBeam: lynx/particles/parameter_beam.py:371-425.
    ParticleBeam:  lynx/particles/particle_beam.py:736-836 -- `mean`, `std(dim=-1)`
    (torch spelling => unbiased, ddof=1; pass ddof=0 for the jnp reading), and
    `sigma_xxp/yyp` = biased mean of centred products (:825-836).
    Derived Twiss: lynx/particles/beam.py:241-310.
    Particle reductions are carried out in float64 from the stored values: this is the
    exact value the reference's float32 reductions approximate.
    """
    out = {}
    names = ["x", "xp", "y", "yp", "s", "p"]
    if beam["type"] == "parameter":
        mu, cov = beam["mu"], beam["cov"]
        dtype = mu.dtype
        for i, n in enumerate(names):
            out["mu_" + n] = mu[..., i]
            out["sigma_" + n] = np.sqrt(np.maximum(cov[..., i, i], dtype.type(1e-20)))
        out["sigma_xxp"] = cov[..., 0, 1]
        out["sigma_yyp"] = cov[..., 2, 3]
    else:
        P = beam["particles"]
        dtype = P.dtype
        P64 = P.astype(np.float64)
        for i, n in enumerate(names):
            out["mu_" + n] = P64[..., i].mean(axis=-1)
            out["sigma_" + n] = P64[..., i].std(axis=-1, ddof=ddof)
        for a, b, key in ((0, 1, "sigma_xxp"), (2, 3, "sigma_yyp")):
            da = P64[..., a] - out["mu_" + names[a]][..., None]
            db = P64[..., b] - out["mu_" + names[b]][..., None]
            out[key] = (da * db).mean(axis=-1)
    tiny = np.finfo(dtype).tiny
    with np.errstate(all="ignore"):
        for pl, cross in (("x", "sigma_xxp"), ("y", "sigma_yyp")):
            em = np.sqrt(np.maximum(
                out["sigma_" + pl] ** 2 * out["sigma_" + pl + "p"] ** 2 - out[cross] ** 2, tiny))
            out["emittance_" + pl] = em
            out["beta_" + pl] = out["sigma_" + pl] ** 2 / em
            out["alpha_" + pl] = -out[cross] / em
        gamma = beam["energy"].astype(np.float64) / ELECTRON_MASS_EV  # beam.py:241-243
        out["relativistic_gamma"] = gamma
        rb = np.ones_like(gamma)  # beam.py:245-251
        pos = np.abs(gamma) > 0
        rb[pos] = np.sqrt(1 - 1 / gamma[pos] ** 2)
        out["relativistic_beta"] = rb
        for pl in ("x", "y"):  # beam.py:272-275, 297-300
            out["normalized_emittance_" + pl] = out["emittance_" + pl] * rb * gamma
    out["energy"] = beam["energy"]
    return out


# ---------------------------------------------------------------------------------
# Tracking
# ---------------------------------------------------------------------------------


def _apply_map(tm, beam):
    """lynx/accelerator/element.py:61-94."""
    if beam["type"] == "parameter":
        mu = _matmul(tm, beam["mu"][..., None])[..., 0]
        cov = _matmul(tm, _matmul(beam["cov"], np.swapaxes(tm, -2, -1)))
        return {"type": "parameter", "mu": mu, "cov": cov, "energy": beam["energy"]}
    new_particles = _matmul(beam["particles"], np.swapaxes(tm, -2, -1))
    return {"type": "particle", "particles": new_particles, "energy": beam["energy"]}


def cavity_track(spec, beam, dtype):
    """lynx/accelerator/cavity.py:97-246 (`_track_beam`)."""
    dtype = np.dtype(dtype)
    length = np.asarray(spec["length"], dtype=dtype)
    voltage = _p(spec, "voltage", length, dtype)
    phase = _p(spec, "phase", length, dtype)
    frequency = _p(spec, "frequency", length, dtype)
    energy = beam["energy"]
    shape = np.broadcast_shapes(length.shape, voltage.shape, phase.shape, frequency.shape,
                                energy.shape)
    length, voltage, phase, frequency, energy = (
        np.broadcast_to(v, shape) for v in (length, voltage, phase, frequency, energy)
    )
    me = dtype.type(ELECTRON_MASS_EV)
    one = dtype.type(1.0)
    with np.errstate(all="ignore"):
        beta0 = np.full(shape, 1.0, dtype=dtype)
        igamma2 = np.full(shape, 0.0, dtype=dtype)
        g0 = np.full(shape, 1e10, dtype=dtype)
        mask = energy != 0
        g0[mask] = energy[mask] / me
        igamma2[mask] = 1 / g0[mask] ** 2
        beta0[mask] = np.sqrt(1 - igamma2[mask])

        phi = np.deg2rad(phase)
        tm = cavity_rmatrix(spec, energy, dtype)
        out = _apply_map(tm, {**beam, "energy": energy})
        delta_energy = voltage * np.cos(phi)

        T566 = dtype.type(1.5) * length * igamma2 / beta0**3
        T556 = np.full(shape, 0.0, dtype=dtype)
        T555 = np.full(shape, 0.0, dtype=dtype)

        # NB: if the branch below is not taken the reference leaves `outgoing_energy`
        # undefined (NameError); the restatement keeps the incoming energy there.
        outgoing_energy = energy
        if np.any(energy + delta_energy > 0):  # :128 -- whole batch
            k = 2 * dtype.type(np.pi) * frequency / dtype.type(SPEED_OF_LIGHT)
            outgoing_energy = energy + delta_energy
            g1 = outgoing_energy / me
            beta1 = np.sqrt(1 - 1 / g1**2)

            if beam["type"] == "parameter":
                mu_in, cov_in = beam["mu"], beam["cov"]
                out["mu"][..., 5] = mu_in[..., 5] * energy * beta0 / (
                    outgoing_energy * beta1
                ) + voltage * beta0 / (outgoing_energy * beta1) * (
                    np.cos(-mu_in[..., 4] * beta0 * k + phi) - np.cos(phi)
                )
                out["cov"][..., 5, 5] = cov_in[..., 5, 5]
            else:
                P = beam["particles"]
                u = lambda v: v[..., None]  # noqa: E731  (unsqueeze(-1))
                out["particles"][..., 5] = P[..., 5] * u(energy) * u(beta0) / (
                    u(outgoing_energy) * u(beta1)
                ) + u(voltage) * u(beta0) / (u(outgoing_energy) * u(beta1)) * (
                    np.cos(-one * P[..., 4] * u(beta0) * u(k) + u(phi)) - u(np.cos(phi))
                )

            dgamma = voltage / me
            if np.any(delta_energy > 0):  # :164 -- whole batch
                T566 = (
                    length * (beta0**3 * g0**3 - beta1**3 * g1**3)
                    / (2 * beta0 * beta1**3 * g0 * (g0 - g1) * g1**3)
                )
                T556 = (
                    beta0 * k * length * dgamma * g0
                    * (beta1**3 * g1**3 + beta0 * (g0 - g1**3))
                    * np.sin(phi)
                    / (beta1**3 * g1**3 * (g0 - g1) ** 2)
                )
                T555 = (
                    beta0**2 * k**2 * length * dgamma / dtype.type(2.0)
                    * (
                        dgamma
                        * (2 * g0 * g1**3 * (beta0 * beta1**3 - 1) + g0**2 + 3 * g1**2 - 2)
                        / (beta1**3 * g1**3 * (g0 - g1) ** 3)
                        * np.sin(phi) ** 2
                        - (g1 * g0 * (beta1 * beta0 - 1) + 1)
                        / (beta1 * g1 * (g0 - g1) ** 2)
                        * np.cos(phi)
                    )
                )

            if beam["type"] == "parameter":
                mu_in, cov_in = beam["mu"], beam["cov"]
                out["mu"][..., 4] = out["mu"][..., 4] + (
                    T566 * mu_in[..., 5] ** 2
                    + T556 * mu_in[..., 4] * mu_in[..., 5]
                    + T555 * mu_in[..., 4] ** 2
                )
                v = (
                    T566 * cov_in[..., 5, 5] ** 2
                    + T556 * cov_in[..., 4, 5] * cov_in[..., 5, 5]
                    + T555 * cov_in[..., 4, 4] ** 2
                )
                out["cov"][..., 4, 4] = v
                out["cov"][..., 4, 5] = v
                out["cov"][..., 5, 4] = out["cov"][..., 4, 5]
            else:
                P = beam["particles"]
                out["particles"][..., 4] = out["particles"][..., 4] + (
                    T566[..., None] * P[..., 5] ** 2
                    + T556[..., None] * P[..., 4] * P[..., 5]
                    + T555[..., None] * P[..., 4] ** 2
                )
        out["energy"] = np.asarray(outgoing_energy, dtype=dtype)
    for key in ("mu", "cov", "particles"):
        if key in out:
            out[key] = out[key].astype(dtype)
    return out


def segment_is_skippable(elements) -> bool:
    """lynx/accelerator/segment.py:317-319."""
    return all(is_skippable(e) for e in elements)


def segment_transfer_map(elements, energy, dtype=np.float32):
    """lynx/accelerator/segment.py:329-338: tm = I; tm = M_e @ tm for e in order."""
    dtype = np.dtype(dtype)
    energy = np.asarray(energy, dtype=dtype)
    if not segment_is_skippable(elements):
        return None
    tm = _eye(energy.shape, dtype)
    for spec in elements:
        tm = _matmul(element_transfer_map(spec, energy, dtype), tm)
    return tm


def partition(elements):
    """
    lynx/accelerator/segment.py:344-351: maximal runs of skippable elements, with every
    non-skippable element on its own.  Returns a list of ("run", [specs]) /
    ("single", spec).
    """
    todos = []
    for spec in elements:
        if not is_skippable(spec):
            todos.append(("single", spec))
        elif not todos or todos[-1][0] != "run":
            todos.append(("run", [spec]))
        else:
            todos[-1][1].append(spec)
    return todos


def segment_track(elements, beam, dtype=np.float32, bpm_readings=None):
    """
    lynx/accelerator/segment.py:340-356.  `bpm_readings`, if a list, receives
    (element_index, stack([mu_x, mu_y])) for every active BPM (bpm.py:48-58).
    """
    dtype = np.dtype(dtype)
    if segment_is_skippable(elements):
        tm = segment_transfer_map(elements, beam["energy"], dtype)
        return _apply_map(tm, beam)
    for what, payload in partition(elements):
        if what == "run":
            tm = segment_transfer_map(payload, beam["energy"], dtype)
            beam = _apply_map(tm, beam)
        else:
            spec = payload
            if spec["kind"] == "cavity":
                beam = cavity_track(spec, beam, dtype)
            elif spec["kind"] == "bpm":
                if bpm_readings is not None:
                    m = beam_moments(beam)
                    bpm_readings.append((elements.index(spec), np.stack([m["mu_x"], m["mu_y"]])))
                beam = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in beam.items()}
            else:
                raise ValueError(spec["kind"])
    return beam


def element_track(spec, beam, dtype=np.float32):
    """
    `element.track(beam)` for one element (element.py:61-94, cavity.py:81-95): the
    element's own map is applied directly, without `Segment.transfer_map`'s product with
    eye(7) (which only matters for signed zeros and NaN spreading).
    """
    dtype = np.dtype(dtype)
    if is_skippable(spec):
        return _apply_map(element_transfer_map(spec, beam["energy"], dtype), beam)
    return segment_track([spec], beam, dtype)


def merge_transfer_maps(elements, energy, dtype=np.float32):
    """
    lynx/accelerator/custom_transfer_map.py:48-85 (`from_merging_elements`): product of
    skippable elements' maps at the (unchanged) incoming energy -> CustomTransferMap spec.
    """
    dtype = np.dtype(dtype)
    assert all(is_skippable(e) for e in elements)
    energy = np.asarray(energy, dtype=dtype)
    tm = _eye(energy.shape, dtype)
    total = None
    for spec in elements:
        tm = _matmul(element_transfer_map(spec, energy, dtype), tm)
        if "length" in spec and spec["length"] is not None:
            ln = np.asarray(spec["length"], dtype=dtype)
            total = ln if total is None else total + ln
    return CustomTransferMap(tm, length=total)


# ---------------------------------------------------------------------------------
# Synthetic inputs shared by tests and bench (seeded)
# ---------------------------------------------------------------------------------


def gaussian_particles(batch_shape, n, seed, dtype=np.float32, mu=None, sigma=None):
    """(*batch, n, 7) uncorrelated 6-D Gaussian, 7th column 1."""
    rng = np.random.Generator(np.random.PCG64(seed))
    mu = np.zeros(6) if mu is None else np.asarray(mu, dtype=np.float64)
    sigma = (np.array([175e-9, 2e-7, 175e-9, 2e-7, 1e-6, 1e-6]) if sigma is None
             else np.asarray(sigma, dtype=np.float64))
    P = np.ones((*batch_shape, n, 7), dtype=dtype)
    P[..., :6] = (rng.standard_normal((*batch_shape, n, 6)) * sigma + mu).astype(dtype)
    return P


def ares_like_segment(dtype=np.float32, batch_shape=(1,)):
    """C1/C2 lattice: the README segment (`README.md:34-48`), BPMs inactive."""
    f = lambda v: np.full(batch_shape, v, dtype=dtype)  # noqa: E731
    return [
        BPM(), Drift(f(1.0)), BPM(), Drift(f(1.0)),
        VerticalCorrector(f(0.3), f(3.142e-3)), Drift(f(0.2)),
        HorizontalCorrector(f(0.3), f(1e-4)), Drift(f(7.0)),
        HorizontalCorrector(f(0.3), f(-1e-4)), Drift(f(0.05)), BPM(),
    ]


def fodo_segment(n_cells=32, dtype=np.float64, batch_shape=(1,), k1_scale=None):
    """C3/C4 lattice: n_cells x [Quad(.2,+k), Drift .5, Quad(.2,-k), Drift .5]."""
    f = lambda v: np.full(batch_shape, v, dtype=dtype)  # noqa: E731
    scale = np.ones(batch_shape, dtype=dtype) if k1_scale is None else np.asarray(k1_scale, dtype)
    k = (dtype(4.2) if not isinstance(dtype, np.dtype) else dtype.type(4.2)) * scale
    cell = []
    for _ in range(n_cells):
        cell += [Quadrupole(f(0.2), k1=k), Drift(f(0.5)), Quadrupole(f(0.2), k1=-k), Drift(f(0.5))]
    return cell


# ---------------------------------------------------------------------------------
# Screen read-out (lynx/accelerator/screen.py:86-216)
# ---------------------------------------------------------------------------------


def screen_bin_edges(resolution, pixel_size, binning, dtype=np.float32):
    """screen.py:107-120 (`pixel_bin_edges`)."""
    resolution, pixel_size = np.asarray(resolution, dtype), np.asarray(pixel_size, dtype)
    eff = resolution / np.asarray(binning, dtype)
    half = resolution * pixel_size / 2
    return (np.linspace(-half[0], half[0], int(eff[0]) + 1, dtype=dtype),
            np.linspace(-half[1], half[1], int(eff[1]) + 1, dtype=dtype))


def screen_reading_particles(particles, resolution, pixel_size, binning, dtype=np.float32):
    """screen.py:196-213: histogramdd of (x, y) per sample, then flipud(hist.T)."""
    edges = screen_bin_edges(resolution, pixel_size, binning, dtype)
    P = np.asarray(particles, dtype=dtype)
    flat = P.reshape(-1, P.shape[-2], 7)
    images = []
    for sample in flat:
        hist, _ = np.histogramdd(np.stack((sample[:, 0], sample[:, 2])).T, bins=edges)
        images.append(np.flipud(hist.T))
    return np.stack(images).reshape(*P.shape[:-2], *images[0].shape).astype(dtype)


def screen_reading_parameters(mu, cov, resolution, pixel_size, binning, dtype=np.float32):
    """screen.py:160-195: exp(log_prob) of MultivariateNormal((mu_x, mu_y), cov_xy) on the pixel grid."""
    dtype = np.dtype(dtype)
    resolution, pixel_size = np.asarray(resolution, dtype), np.asarray(pixel_size, dtype)
    half = resolution * pixel_size / 2
    step = pixel_size * np.asarray(binning, dtype)
    xs = np.arange(-half[0], half[0], step[0], dtype=dtype).astype(np.float64)
    ys = np.arange(-half[1], half[1], step[1], dtype=dtype).astype(np.float64)
    mu, cov = np.asarray(mu, np.float64), np.asarray(cov, np.float64)
    out = []
    for m, c in zip(mu.reshape(-1, 7), cov.reshape(-1, 7, 7)):
        s2 = np.array([[c[0, 0], c[0, 2]], [c[2, 0], c[2, 2]]])
        inv, det = np.linalg.inv(s2), np.linalg.det(s2)
        dx, dy = xs[:, None] - m[0], ys[None, :] - m[2]
        maha = inv[0, 0] * dx * dx + 2 * inv[0, 1] * dx * dy + inv[1, 1] * dy * dy
        img = np.exp(-0.5 * maha - np.log(2 * np.pi) - 0.5 * np.log(det))
        out.append(img[::-1])  # flip(dims=[1]) of the stacked (B, nx, ny) image
    return np.stack(out).reshape(*mu.shape[:-1], len(xs), len(ys))


# ---------------------------------------------------------------------------------
# Aperture (lynx/accelerator/aperture.py:69-108)
# ---------------------------------------------------------------------------------


def aperture_mask(particles, x_max, y_max, shape="rectangular"):
    """aperture.py:78-86: which particles survive (x_max, y_max broadcast over the particle axis)."""
    xs, ys = particles[..., 0], particles[..., 2]
    x_max = np.asarray(x_max, dtype=particles.dtype)[..., None]
    y_max = np.asarray(y_max, dtype=particles.dtype)[..., None]
    if shape == "rectangular":
        return np.logical_and(np.logical_and(xs > -x_max, xs < x_max), np.logical_and(ys > -y_max, ys < y_max))
    assert shape == "elliptical", f"Unknown aperture shape {shape}"
    with np.errstate(all="ignore"):
        return xs ** 2 / x_max ** 2 + ys ** 2 / y_max ** 2 <= 1.0
