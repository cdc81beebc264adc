"""GPU tests of the rows either side of the step: vorticity and energy diagnostics, the block-averaged
density for the renderer, and restart files."""
import os
import struct

import numpy as np
import pytest

from oracle.oracle3d import Oracle3D
from tests.util3d import make_pair

pytestmark = pytest.mark.gpu


def advance_pair(o, g, steps):
    for _ in range(steps):
        o.rb_sweep(); g.sweeps(1)
        o.semi_lagrange(1.0 / 30.0); g.semi_lagrange(1.0 / 30.0)


@pytest.mark.parametrize("slabs", [1, 3])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_vorticity_and_diagnostics_match_oracle(fsamd, slabs, dtype):
    # stencil of computeVorticity (3_fluid/FluidSim.h:220-229) on the three components of the curl; the
    # cell-centred velocity of draw_velocity_field (main.cpp:138-142) for energy and speed
    o, g = make_pair(fsamd, 26, 18, 21, seed=41, scale=8.0, dtype=dtype, local_slabs=slabs)
    advance_pair(o, g, 2)
    vo, ens_o = o.vorticity()
    assert np.array_equal(vo, g.vorticity())
    ke, mx, mass = o.diagnostics()
    d = g.diagnostics()
    assert d.kinetic_energy == pytest.approx(ke, rel=1e-12)
    assert d.max_speed == mx
    assert d.density_sum == pytest.approx(mass, rel=1e-12)
    assert d.enstrophy == pytest.approx(ens_o, rel=1e-12)


@pytest.mark.parametrize("factor", [1, 2, 4, 5])
def test_downsampled_density_matches_oracle(fsamd, factor):
    o, g = make_pair(fsamd, 30, 22, 17, seed=43, scale=5.0)
    advance_pair(o, g, 1)
    assert np.array_equal(o.downsample_density(factor), g.downsample_density(factor))


def test_downsampled_density_on_slabs(fsamd):
    o, g = make_pair(fsamd, 20, 12, 24, seed=44, local_slabs=3)       # slabs of 8 planes
    assert np.array_equal(o.downsample_density(4), g.downsample_density(4))
    with pytest.raises(fsamd.FluidSimError):
        g.downsample_density(3)                                         # slab boundaries are not multiples of 3


@pytest.mark.parametrize("slabs", [1, 2])
def test_restart_file_continues_bit_for_bit(fsamd, tmp_path, slabs):
    path = str(tmp_path / "state.fs3d")
    n = 24
    cfg = dict(step_mode=1, stop_mode=2, tol=1e-4, max_iters=200, check_every=0)
    a = fsamd.FluidSim3D(n, n, n, **cfg)
    a.sphere_influence(n / 10.0)
    a.solid_sphere(3.0, 12.0, 16.0, 12.0)
    for it in range(2):
        a.fill_sphere_influence(n / 10.0, it)
        a.advance(1.0 / 30.0)
    a.save(path)
    for it in range(2, 4):
        a.fill_sphere_influence(n / 10.0, it)
        a.advance(1.0 / 30.0)

    # the file: 64-byte header, cell types, then u, v, w, density
    raw = open(path, "rb").read()
    assert raw[:7] == b"FSAMD3D" and len(raw) == 64 + n ** 3 + 4 * 4 * n ** 3
    version, dtype, nx, ny, nz = struct.unpack_from("<5i", raw, 8)
    assert (version, dtype, nx, ny, nz) == (1, fsamd.F32, n, n, n)
    step, time = struct.unpack_from("<qd", raw, 40)
    assert step == 2 and time == pytest.approx(2.0 / 30.0)

    b = fsamd.FluidSim3D(n, n, n, local_slabs=slabs, **cfg)
    b.load(path)
    assert b.getStep() == 2
    assert np.array_equal(np.frombuffer(raw, np.uint8, n ** 3, 64).reshape(n, n, n), b.cells)
    for it in range(2, 4):
        b.fill_sphere_influence(n / 10.0, it)
        b.advance(1.0 / 30.0)
    for k in range(4):
        assert np.array_equal(a.get_field(k), b.get_field(k))
    assert a.getStep() == b.getStep() == 4

    c = fsamd.FluidSim3D(n + 2, n, n)
    with pytest.raises(fsamd.FluidSimError, match="grid size"):
        c.load(path)
    # a truncated file, and one with bytes left over from an older, larger file, are refused; saving over the
    # latter sets the exact size again
    with open(path, "r+b") as f:
        f.truncate(len(raw) - 5)
    with pytest.raises(fsamd.FluidSimError, match="incomplete"):
        b.load(path)
    with open(path, "wb") as f:
        f.write(raw + b"stale tail")
    with pytest.raises(fsamd.FluidSimError, match="trailing"):
        b.load(path)
    b.save(path)
    assert os.path.getsize(path) == len(raw)
    b.load(path)
