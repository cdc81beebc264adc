"""The benchmark configurations at their full sizes, through properties that need no oracle run of that size:
z-slab runs must equal the single domain bit for bit (red-black order is partition independent), every projection
must end on its stopping rule, reductions must agree to rounding.  Fields of the largest grids are compared plane
by plane (fs3d_get_field_plane) and through fs3d_diagnostics instead of 4 GiB host copies."""
import numpy as np
import pytest

import fsamd_loader

fsamd_loader.load()      # registers the package under its importable name
import fluid_simulation_amd.scenes as scenes  # noqa: E402

pytestmark = pytest.mark.gpu

DT = 1.0 / 30.0


def diag_tuple(g):
    d = g.diagnostics()
    return np.array([d.kinetic_energy, d.max_speed, d.density_sum, d.enstrophy])


def test_config2_vortex_ring_256_under_the_residual_rule(fsamd):
    # "Vortex ring 256^3 on one MI355X, GS to residual 1e-6": two reflection steps under the real stopping rule
    # (L2 residual of 3_fluid/FluidSim.cpp:25-43 <= 1e-6 or the 1000-sweep cap of main.cpp:108), single domain
    # against four z-slabs
    n, tol = 256, 1e-6
    cfg = dict(stop_mode=fsamd.STOP_L2_RESIDUAL, tol=tol, max_iters=1000, check_every=0, step_mode=fsamd.STEP_REFLECTION)
    a = fsamd.FluidSim3D(n, n, n, **cfg)
    b = fsamd.FluidSim3D(n, n, n, local_slabs=4, **cfg)
    for g in (a, b):
        scenes.vortex_ring(g)
    for it in range(2):
        for g in (a, b):
            scenes.vortex_ring_per_step(g, it)
            g.advance(DT)
        sa, sb = a.stats(), b.stats()
        # every projection stopped on the rule or on the cap, identically in both decompositions
        assert sa.sweeps_step == sb.sweeps_step and 2 <= sa.sweeps_step <= 2000
        assert sa.metric <= tol or sa.sweeps == 1000
        assert sb.metric == pytest.approx(sa.metric, rel=1e-9)
    for k in range(4):
        assert np.array_equal(a.get_field(k), b.get_field(k)), "uvwd"[k]
    assert np.allclose(diag_tuple(a), diag_tuple(b), rtol=1e-12, atol=0)


def compare_planes(a, b, planes):
    for z in planes:
        for k in range(4):
            assert np.array_equal(a.get_field_plane(k, z), b.get_field_plane(k, z)), (z, "uvwd"[k])


def test_config5_per_gpu_slab_1024x1024x128(fsamd):
    # "1024^3 vortex-ring collision on 8 x MI355X": the grid one GPU holds there (1024 x 1024 x 128, the wide-row
    # variants of every kernel), ring-collision emitters, single domain against four z-slabs on the same GPU
    nx, ny, nz = 1024, 1024, 128
    cfg = dict(stop_mode=fsamd.STOP_FIXED, max_iters=3, step_mode=fsamd.STEP_REFLECTION)
    a = fsamd.FluidSim3D(nx, ny, nz, **cfg)
    b = fsamd.FluidSim3D(nx, ny, nz, local_slabs=4, **cfg)
    for g in (a, b):
        scenes.ring_collision(g)
    before = diag_tuple(a)
    for it in range(2):
        for g in (a, b):
            scenes.ring_collision_per_step(g, it)
            g.advance(DT)
    compare_planes(a, b, [0, 1, 30, 31, 32, 33, 63, 64, 65, 95, 96, 97, 126, 127])
    da, db = diag_tuple(a), diag_tuple(b)
    assert np.allclose(da, db, rtol=1e-12, atol=0)
    # the two jets face each other: energy is there, bounded by what the emitters hold, and the shell stays untouched
    assert 0 < da[0] <= before[0] * 1.01 and da[1] <= 20.0 * 1.01
    assert not a.get_field_plane(1, 0).any() and not a.get_field_plane(1, nz - 1).any()
    l2, mx = a.residual()
    l2b, mxb = b.residual()
    assert mx == mxb and l2 == pytest.approx(l2b, rel=1e-12)


def test_config5_whole_1024_cubed_on_one_gpu_in_eight_slabs(fsamd):
    # the whole 1024^3 grid (46 B per cell: 49 GB, the 64-bit-offset kernels) as the eight slabs of the 8-GPU run,
    # on one GPU, against the single domain
    n = 1024
    cfg = dict(stop_mode=fsamd.STOP_FIXED, max_iters=2, step_mode=fsamd.STEP_REFLECTION)
    a = fsamd.FluidSim3D(n, n, n, **cfg)
    b = fsamd.FluidSim3D(n, n, n, local_slabs=8, **cfg)
    for g in (a, b):
        scenes.ring_collision(g)
        scenes.ring_collision_per_step(g, 0)
        g.advance(DT)
    compare_planes(a, b, [1, 127, 128, 129, 511, 512, 513, 895, 896, 1022])
    assert np.allclose(diag_tuple(a), diag_tuple(b), rtol=1e-12, atol=0)
    assert a.stats().sweeps_step == b.stats().sweeps_step == 4
