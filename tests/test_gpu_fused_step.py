"""One sweep folded into the advection that follows it (k_sweep_advect) against the oracle's sweep followed by its
semi_lagrange, bit for bit: ragged and multi-tile shapes, several z-chunks, masks with Object and StaticFluid cells,
over-relaxation, both density modes through whole steps; the cells it cannot serve (a backtrace of a cell or more,
non-finite values) leave the state untouched; and whole runs of fs3d_advance, which must give the same fields and
sweep counts whether or not the fused launch is allowed."""
import numpy as np
import pytest

from tests.util3d import assert_fields_equal, make_pair

pytestmark = pytest.mark.gpu

SHAPES = [(20, 13, 11), (8, 3, 3), (3, 3, 3), (64, 16, 9), (65, 17, 12), (70, 40, 21), (130, 35, 8), (200, 20, 6), (33, 50, 40),
          (24, 10, 150), (128, 128, 20)]


@pytest.mark.parametrize("shape", SHAPES)
def test_sweep_and_advection_in_one_launch_bit_exact(fsamd, shape):
    nx, ny, nz = shape
    o, g = make_pair(fsamd, nx, ny, nz, seed=nx * 7 + nz, scale=8.0)
    dt = 1.0 / 30.0          # |v| dt < 0.27 cells
    for it in range(3):
        o.project_rb(stop_mode=0, max_iter=1)
        _, want = o.div_sumsq()
        want_l2 = np.sqrt(o.div_sumsq()[0]) / ((nx - 2) * (ny - 2) * (nz - 2))
        o.semi_lagrange(dt)
        l2, served = g.sweep_advect(dt)
        assert served
        assert l2 == pytest.approx(want_l2, rel=1e-12)
        assert_fields_equal(o, g, what=f"{shape}, fused launch {it}")


def test_fused_launch_with_over_relaxation(fsamd):
    o, g = make_pair(fsamd, 72, 37, 29, seed=5, scale=6.0, omega=1.3)
    for it in range(2):
        o.project_rb(omega=1.3, stop_mode=0, max_iter=1)
        o.semi_lagrange(0.02)
        _, served = g.sweep_advect(0.02)
        assert served
        assert_fields_equal(o, g, what=f"omega 1.3, launch {it}")


@pytest.mark.parametrize("what", ["fast", "inf", "nan"])
def test_fused_launch_refuses_what_it_cannot_serve(fsamd, what):
    # one cell whose backtrace is longer than a cell, or a non-finite value: the launch reports it, the state stays,
    # and the separate launches give the oracle's result
    nx, ny, nz = 40, 20, 18
    o, g = make_pair(fsamd, nx, ny, nz, seed=12, scale=4.0)
    u = o.field(0).copy()
    u[9, 10, 20] = {"fast": 100.0, "inf": np.inf, "nan": np.nan}[what]
    o.field(0)[...] = u
    g.set_field(0, u)
    before = [g.get_field(k) for k in range(4)]
    _, served = g.sweep_advect(1.0 / 30.0)
    assert not served
    for k in range(4):
        assert np.array_equal(g.get_field(k), before[k], equal_nan=True)
    with np.errstate(all="ignore"):
        o.project_rb(stop_mode=0, max_iter=1)
        o.semi_lagrange(1.0 / 30.0)
    g.sweeps(1)
    g.semi_lagrange(1.0 / 30.0)
    for k in range(4):
        assert np.array_equal(o.field(k), g.get_field(k), equal_nan=True), k


def test_fused_launch_needs_its_configuration(fsamd):
    _, g = make_pair(fsamd, 20, 13, 11, dtype=np.float64)
    with pytest.raises(fsamd.FluidSimError):
        g.sweep_advect(0.02)
    _, g = make_pair(fsamd, 20, 13, 16, local_slabs=2)
    with pytest.raises(fsamd.FluidSimError):
        g.sweep_advect(0.02)


def ring_run(fsamd, n, steps, step_mode, **cfg):
    g = fsamd.FluidSim3D(n, n, n, stop_mode=fsamd.STOP_L2_RESIDUAL, max_iters=50, check_every=-1, step_mode=step_mode, **cfg)
    g.sphere_influence(n / 10.0)
    out = []
    for it in range(steps):
        g.fill_sphere_influence(n / 10.0, it)
        g.advance(1.0 / 30.0)
        st = g.stats()
        out.append(([g.get_field(k) for k in range(4)], st.sweeps_step))
    fused = g.stats().fused
    g.close()
    return out, fused


@pytest.mark.parametrize("step_mode", [0, 1])
@pytest.mark.parametrize("n,tol", [(48, 1e-2), (72, 3e-3), (72, 1e-3)])
def test_steps_are_the_same_with_and_without_the_fused_launch(fsamd, monkeypatch, n, tol, step_mode):
    # a tolerance loose enough for projections of one sweep once the start-up is over: the fused launch then takes
    # the first projection of every step, and nothing changes but the number of launches
    monkeypatch.delenv("FSAMD_FUSED_STEP", raising=False)
    want, none = ring_run(fsamd, n, 14, step_mode, tol=tol)
    monkeypatch.setenv("FSAMD_FUSED_STEP", "1")      # (off by default: the launch is correct but not faster, see fs3d.hip)
    got, fused = ring_run(fsamd, n, 14, step_mode, tol=tol)
    assert none == 0 and fused >= 4, (none, fused, [w[1] for w in want])
    for step, ((fa, sa), (fb, sb)) in enumerate(zip(want, got)):
        assert sa == sb, (step, sa, sb)
        for k in range(4):
            assert np.array_equal(fa[k], fb[k]), (step, k)
