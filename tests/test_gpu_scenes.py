"""The benchmark scenes beyond the vortex ring (BASELINE configs 4 and 5) against the oracle at small
sizes, and size-independent properties at the configurations' full sizes."""
import numpy as np
import pytest

from oracle.oracle3d import Oracle3D
from tests.util3d import assert_fields_equal

pytestmark = pytest.mark.gpu


def scenes(fsamd):
    import fluid_simulation_amd.scenes as sc
    return sc


def test_smoke_plume_with_obstacle_matches_oracle(fsamd):
    # non-cubic grid, interior Object cells, buoyancy, source re-applied each step (config 4 in small)
    sc = scenes(fsamd)
    nx, ny, nz = 24, 48, 20
    o = Oracle3D(nx, ny, nz)
    g = fsamd.FluidSim3D(nx, ny, nz, buoyancy=1.0, stop_mode=fsamd.STOP_FIXED, max_iters=12)
    sc.smoke_plume(o); sc.smoke_plume(g)
    assert np.array_equal(o.cells, g.cells) and (g.cells[1:-1, 1:-1, 1:-1] == 1).sum() > 50
    dt = 0.05
    for it in range(6):
        sc.smoke_plume_per_step(o, it); sc.smoke_plume_per_step(g, it)
        o.buoyancy(dt); o.project_rb(stop_mode=0, max_iter=12); o.semi_lagrange(dt)
        g.advance(dt)
        assert_fields_equal(o, g, what=f"plume step {it}")
    assert np.abs(g.v).max() > 0     # the plume rises


def test_ring_collision_matches_oracle(fsamd):
    sc = scenes(fsamd)
    n = 30
    o = Oracle3D(n, n, n)
    g = fsamd.FluidSim3D(n, n, n)
    sc.ring_collision(o); sc.ring_collision(g)
    assert np.array_equal(o.cells, g.cells)
    for it in range(3):
        sc.ring_collision_per_step(o, it); sc.ring_collision_per_step(g, it)
        o.project_rb(stop_mode=1, tol=5.0, max_iter=1000); o.semi_lagrange(1.0 / 30.0)
        g.advance(1.0 / 30.0)
        assert_fields_equal(o, g, what=f"collision step {it}")
    v = g.v
    assert v[:, : n // 2].max() > 0 and v[:, n // 2:].min() < 0   # the two jets oppose each other


def test_smoke_plume_full_size_properties(fsamd):
    # BASELINE config 4: 256 x 512 x 256 with a solid obstacle
    sc = scenes(fsamd)
    nx, ny, nz = 256, 512, 256
    g = fsamd.FluidSim3D(nx, ny, nz, buoyancy=1.0, stop_mode=fsamd.STOP_FIXED, max_iters=10)
    sc.smoke_plume(g)
    cells = g.cells
    solid = cells == 1
    assert solid[1:-1, 1:-1, 1:-1].sum() > 100000
    for it in range(2):
        sc.smoke_plume_per_step(g, it)
        g.advance(0.05)
    u, v, w = g.u, g.v, g.w
    # faces of non-Fluid locations never move (main.cpp:227-232, :260): they are still zero
    assert not u[cells != 0].any() and not v[cells != 0].any() and not w[cells != 0].any()
    assert np.isfinite(v).all() and v.max() > 0
    # slabs give the same bits at full size
    gs = fsamd.FluidSim3D(nx, ny, nz, buoyancy=1.0, stop_mode=fsamd.STOP_FIXED, max_iters=10, local_slabs=2)
    sc.smoke_plume(gs)
    for it in range(2):
        sc.smoke_plume_per_step(gs, it)
        gs.advance(0.05)
    assert np.array_equal(gs.v, v) and np.array_equal(gs.d, g.d)
