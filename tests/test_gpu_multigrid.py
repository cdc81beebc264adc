"""GPU tests of the multigrid projection (FS_SOLVER_MULTIGRID; not in the reference): the V-cycle against its
CPU restatement bit for bit, its fixed point against plain sweeps, and the projection loop built on it."""
import numpy as np
import pytest

from oracle.oracle3d import Oracle3D
from tests.util3d import assert_fields_equal, make_pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(16, 16, 16), (24, 48, 24), (33, 9, 17), (70, 12, 10), (64, 40, 36)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_vcycle_matches_oracle_bit_for_bit(fsamd, shape, dtype):
    nx, ny, nz = shape
    o, g = make_pair(fsamd, nx, ny, nz, seed=61, scale=5.0, dtype=dtype)
    for cycle in range(2):
        o.mg_vcycle(2, 2, 30, 2.0)
        g.mg_vcycle()
        assert_fields_equal(o, g, "uvw", what=f"V-cycle {cycle} {shape}")


def test_vcycle_after_the_cell_types_change(fsamd):
    # the level weights are rebuilt when the mask changes
    o, g = make_pair(fsamd, 28, 20, 24, seed=62, scale=3.0, masks=False)
    o.mg_vcycle(2, 2, 30, 2.0); g.mg_vcycle()
    o.solid_sphere(4.0, 14.0, 10.0, 12.0); g.solid_sphere(4.0, 14.0, 10.0, 12.0)
    o.mg_vcycle(2, 2, 30, 2.0); g.mg_vcycle()
    assert_fields_equal(o, g, "uvw", what="V-cycle after solid_sphere")


def test_multigrid_reaches_the_fixed_point_of_the_sweeps(fsamd):
    # closed box, compatible data: both solvers converge to the projection of the same field
    _, a = make_pair(fsamd, 40, 40, 40, seed=63, masks=False, closed=True, stop_mode=2, tol=1e-9, max_iters=20000,
                     check_every=50)
    _, b = make_pair(fsamd, 40, 40, 40, seed=63, masks=False, closed=True, stop_mode=2, tol=1e-9, max_iters=400, solver=1)
    sa, ra = a.project()
    sb, rb = b.project()
    assert ra <= 1e-9 and rb <= 1e-9
    assert sb * 10 < sa                      # an order of magnitude fewer fine sweeps
    num = sum(float(((a.get_field(k).astype(np.float64) - b.get_field(k)) ** 2).sum()) for k in range(3)) ** 0.5
    den = sum(float((a.get_field(k).astype(np.float64) ** 2).sum()) for k in range(3)) ** 0.5
    assert num / den < 1e-5


def test_multigrid_step_runs_the_benchmark_scene(fsamd):
    n = 64
    g = fsamd.FluidSim3D(n, n, n, step_mode=1, stop_mode=2, tol=1e-6, max_iters=200, solver=1)
    g.sphere_influence(n / 10.0)
    for it in range(4):
        g.fill_sphere_influence(n / 10.0, it)
        g.advance(1.0 / 30.0)
    st = g.stats()
    assert st.sweeps_step >= 2
    assert np.isfinite(g.get_field(1)).all()


def test_multigrid_configuration_is_validated(fsamd):
    with pytest.raises(fsamd.FluidSimError):
        fsamd.FluidSim3D(16, 16, 18, solver=1, local_slabs=2)      # 9 + 9 planes: the second slab starts on an odd one
    with pytest.raises(fsamd.FluidSimError):
        fsamd.FluidSim3D(16, 16, 16, solver=1, ordering=fsamd.ORDER_LEXICOGRAPHIC)
    with pytest.raises(fsamd.FluidSimError):
        fsamd.FluidSim3D(16, 16, 16, solver=1, stop_mode=1)


@pytest.mark.parametrize("kernel", [0, 1], ids=["one_pass", "two_pass"])
@pytest.mark.parametrize("shape,slabs", [((28, 18, 24), 3), ((33, 20, 23), 3), ((40, 22, 32), 4), ((64, 40, 36), 2)])
def test_multigrid_in_slabs_equals_single_domain(fsamd, shape, slabs, kernel):
    # every slab contributes its planes of level 1, the coarse levels are solved on the whole grid: not a bit
    # may differ from the single-domain cycle (random obstacles and StaticFluid cells inside)
    nx, ny, nz = shape
    _, a = make_pair(fsamd, nx, ny, nz, seed=6, scale=3.0, sweep_kernel=kernel)
    _, b = make_pair(fsamd, nx, ny, nz, seed=6, scale=3.0, sweep_kernel=kernel, local_slabs=slabs)
    for it in range(3):
        a.mg_vcycle(); b.mg_vcycle()
        for k in range(3):
            assert np.array_equal(a.get_field(k), b.get_field(k)), (it, k)
    ra, rb = a.residual(), b.residual()
    assert ra[1] == rb[1] and ra[0] == pytest.approx(rb[0], rel=1e-12)


def test_multigrid_steps_in_slabs(fsamd):
    n = 48
    cfg = dict(step_mode=1, stop_mode=2, tol=1e-5, max_iters=200, solver=1, check_every=-1)
    a = fsamd.FluidSim3D(n, n, n, **cfg)
    b = fsamd.FluidSim3D(n, n, n, local_slabs=3, **cfg)
    for g in (a, b):
        g.sphere_influence(n / 10.0)
        g.solid_sphere(5.0, 24.0, 30.0, 20.0)
    for it in range(5):
        for g in (a, b):
            g.fill_sphere_influence(n / 10.0, it)
            g.advance(1.0 / 30.0)
        assert a.stats().sweeps_step == b.stats().sweeps_step
        for k in range(4):
            assert np.array_equal(a.get_field(k), b.get_field(k)), (it, k)
