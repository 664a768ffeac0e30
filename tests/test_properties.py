"""
Property tests (hypothesis) of the host-side pieces that have algebraic laws: batch / particle
sharding covers its range exactly once, merging moment records is independent of how the beam
was cut, the partition of a lattice keeps every element exactly once and in order, and the
oracle's composed map equals the product of its element maps in lattice order.
"""

import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

import lynx_amd as lx
from lynx_amd import _ffi, engine
from lynx_amd.parallel import merge_records, shard_batch
from oracle import lynx_oracle as o


@given(total=st.integers(0, 5000), world=st.integers(1, 16))
def test_shards_tile_the_range(total, world):
    slices = [shard_batch(total, world, r) for r in range(world)]
    assert slices[0][0] == 0 and slices[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(slices, slices[1:]))
    sizes = [b - a for a, b in slices]
    assert max(sizes) - min(sizes) <= 1 and sorted(sizes, reverse=True) == sizes


def _record(Q):
    rec = np.zeros(36)
    rec[:7] = Q.mean(axis=0)
    k = 7
    for i in range(6):
        for j in range(i, 6):
            rec[k] = ((Q[:, i] - rec[i]) * (Q[:, j] - rec[j])).mean()
            k += 1
    rec[35] = len(Q)
    return rec


@settings(max_examples=40, deadline=None)
@given(n=st.integers(2, 400), cuts=st.lists(st.integers(0, 400), min_size=0, max_size=5), seed=st.integers(0, 2**31 - 1))
def test_merging_records_does_not_depend_on_the_cuts(n, cuts, seed):
    rng = np.random.default_rng(seed)
    Q = rng.normal(size=(n, 7)) * [1e-4, 1e-5, 1e-4, 1e-5, 1e-5, 1e-3, 0] + [3e-3, 0, -1e-3, 0, 0, 1e-3, 1]
    edges = sorted({0, n, *[c % (n + 1) for c in cuts]})
    parts = [_record(Q[a:b]) for a, b in zip(edges, edges[1:])]
    merged, whole = merge_records(np.stack(parts)), _record(Q)
    assert merged[35] == n
    np.testing.assert_allclose(merged[:7], whole[:7], rtol=1e-12, atol=1e-18)
    np.testing.assert_allclose(merged[7:28], whole[7:28], rtol=1e-8, atol=1e-22)


KINDS = st.sampled_from(["drift", "quad", "hcor", "marker", "bpm", "bpm_on", "cavity_off", "cavity_on", "screen_on"])


@settings(max_examples=60, deadline=None)
@given(kinds=st.lists(KINDS, min_size=0, max_size=25))
def test_partition_keeps_every_element_once_and_in_order(kinds):
    f = lambda v: np.array([v], dtype=np.float32)  # noqa: E731
    make = {"drift": lambda: lx.Drift(f(0.1)), "quad": lambda: lx.Quadrupole(f(0.1), k1=f(1.0)),
            "hcor": lambda: lx.HorizontalCorrector(f(0.1), angle=f(1e-4)), "marker": lambda: lx.Marker(),
            "bpm": lambda: lx.BPM(), "bpm_on": lambda: lx.BPM(is_active=True),
            "cavity_off": lambda: lx.Cavity(f(1.0)), "cavity_on": lambda: lx.Cavity(f(1.0), voltage=f(1e6)),
            "screen_on": lambda: lx.Screen(is_active=True)}
    elements = [make[k]() for k in kinds]
    items = engine.partition(elements)
    flat = []
    for item in items:
        if isinstance(item, engine.Program):
            assert item.leaves and item.steps  # no empty programs
            covered = []
            for kind, first, last in item.steps:
                assert first < last and (kind == _ffi.STEP_RUN or last == first + 1)
                covered += list(range(first, last))
            assert covered == list(range(len(item.leaves)))  # steps tile the leaves
            for kind, first, last in item.steps:
                for el in item.leaves[first:last]:
                    assert el.is_skippable == (kind == _ffi.STEP_RUN)
            flat += item.leaves
        else:
            assert item._host_barrier
            flat.append(item)
    assert len(flat) == len(elements) and all(a is b for a, b in zip(flat, elements))
    # two neighbouring items are never both programs (a program is a maximal stretch)
    assert not any(isinstance(a, engine.Program) and isinstance(b, engine.Program) for a, b in zip(items, items[1:]))


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(1, 12))
def test_oracle_composition_is_the_ordered_product(seed, n):
    rng = np.random.default_rng(seed)
    f = lambda v: np.array([v])  # noqa: E731
    specs = []
    for _ in range(n):
        specs.append(rng.choice([lambda: o.Drift(f(rng.uniform(0.1, 1))),
                                 lambda: o.Quadrupole(f(rng.uniform(0.1, 0.5)), k1=f(rng.uniform(-5, 5)), tilt=f(rng.uniform(-1, 1))),
                                 lambda: o.HorizontalCorrector(f(0.1), angle=f(rng.normal(0, 1e-3))),
                                 lambda: o.Dipole(f(0.3), angle=f(rng.uniform(-0.2, 0.2)), e1=f(0.05), e2=f(0.02))])())
    energy = f(1e8)
    product = np.eye(7)[None]
    for spec in specs:
        product = np.matmul(o.element_transfer_map(spec, energy, np.float64), product)
    np.testing.assert_allclose(o.segment_transfer_map(specs, energy, np.float64), product, rtol=1e-12, atol=1e-15)
