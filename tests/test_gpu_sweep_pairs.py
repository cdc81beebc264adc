"""Two sweeps in one pass over HBM (k_rb_rows2) against the oracle's separate sweeps, bit for bit: shapes with one and
several tiles in x and y, several z-chunks, rows that are and are not multiples of 16 bytes, both precisions, masks
with Object and StaticFluid cells, over-relaxation, every tile shape of the kernel (one, two or four rows per wave)
and the shape the library picks itself; the projection loop, which must stop on the same sweep and
leave the same field whether or not its sweeps run in pairs; and z-slabs, which run the same pass on their owned
planes with four halo planes either side and exchange once per pass."""
import numpy as np
import pytest

from tests.util3d import assert_fields_equal, make_pair

pytestmark = pytest.mark.gpu

# (264 = a 256-column tile and 16 columns, 332 = one and 84: the shapes with a second launch for the last columns)
SHAPES = [(264, 30, 12), (332, 18, 20), (20, 13, 11), (8, 3, 3), (4, 40, 5), (260, 40, 24), (64, 70, 50), (132, 18, 9), (30, 21, 17), (516, 20, 12),
          (96, 33, 140), (258, 19, 7), (72, 150, 10)]


@pytest.mark.parametrize("lanes_per_row", [0, 64, 32, 16, "64+16", "64+32"])
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_sweep_pairs_bit_exact(fsamd, monkeypatch, shape, dtype, lanes_per_row):
    nx, ny, nz = shape
    if lanes_per_row:
        monkeypatch.setenv("FSAMD_ROWS2_LANES_PER_ROW", str(lanes_per_row))
    else:
        monkeypatch.delenv("FSAMD_ROWS2_LANES_PER_ROW", raising=False)
    o, g = make_pair(fsamd, nx, ny, nz, seed=nx + ny + nz, dtype=dtype, stop_mode=fsamd.STOP_FIXED, max_iters=1)
    done = 0
    for n in (2, 3, 4):          # a pair; a pair and a single sweep; two pairs
        g.sweeps(n)
        done += n
        o.project_rb(stop_mode=0, max_iter=n)
        assert_fields_equal(o, g, "uvw", what=f"{shape} {dtype.__name__} lanes/row {lanes_per_row} after {done} sweeps")


def test_sweep_pairs_with_over_relaxation(fsamd):
    o, g = make_pair(fsamd, 72, 37, 29, seed=5, omega=1.3, stop_mode=fsamd.STOP_FIXED, max_iters=6)
    g.project()
    o.project_rb(omega=1.3, stop_mode=0, max_iter=6)
    assert_fields_equal(o, g, "uvw", what="omega 1.3")


@pytest.mark.parametrize("stop_mode,tol,check_every", [(2, 2e-5, 1), (2, 2e-5, 3), (2, 2e-5, 0), (1, 2.0, 2), (1, 2.0, 1)])
def test_projection_stops_on_the_same_sweep_with_and_without_pairs(fsamd, stop_mode, tol, check_every):
    # FS_SWEEP_FUSED pairs its sweeps between evaluations; FS_SWEEP_FUSED_LDS never does
    cfg = dict(seed=11, scale=5.0, stop_mode=stop_mode, tol=tol, max_iters=60, check_every=check_every)
    _, a = make_pair(fsamd, 48, 30, 26, sweep_kernel=fsamd.SWEEP_FUSED, **cfg)
    _, b = make_pair(fsamd, 48, 30, 26, sweep_kernel=fsamd.SWEEP_FUSED_LDS, **cfg)
    for _ in range(2):
        na, ma = a.project()
        nb, mb = b.project()
        assert na == nb and 1 < na <= 60, (na, ma)
        assert ma == pytest.approx(mb, rel=1e-12)
        for k in range(3):
            assert np.array_equal(a.get_field(k), b.get_field(k))
        a.semi_lagrange(0.02); b.semi_lagrange(0.02)


def test_reflection_steps_with_paired_sweeps_equal_the_oracle(fsamd):
    from tests.test_gpu_north_star import reflection_step_oracle
    o, g = make_pair(fsamd, 40, 36, 34, seed=2, scale=3.0, step_mode=fsamd.STEP_REFLECTION, stop_mode=fsamd.STOP_FIXED, max_iters=5)
    for step in range(2):
        g.advance(0.02)
        reflection_step_oracle(o, np.float32(0.02), 5)
        assert_fields_equal(o, g, what=f"reflection step {step} with 5 sweeps per projection")


@pytest.mark.parametrize("lanes_per_row", [64, 16])
def test_sweep_pairs_with_subnormal_huge_and_non_finite_values(fsamd, monkeypatch, lanes_per_row):
    # the quotient of the visit is formed without the division instruction sequence: values of every magnitude, zeros of
    # both signs, infinities and NaNs must still come out as the oracle's IEEE division gives them
    monkeypatch.setenv("FSAMD_ROWS2_LANES_PER_ROW", str(lanes_per_row))
    nx, ny, nz = 72, 21, 14
    o, g = make_pair(fsamd, nx, ny, nz, seed=3, stop_mode=fsamd.STOP_FIXED, max_iters=1)
    rng = np.random.default_rng(8)
    for k in range(3):
        mag = np.exp2(rng.integers(-149, 100, size=(nz, ny, nx)).astype(np.float64))
        f = (mag * rng.choice([-1.0, 1.0], size=mag.shape) * rng.uniform(1.0, 2.0, size=mag.shape)).astype(np.float32)
        f[rng.random(f.shape) < 0.02] = 0.0
        f[rng.random(f.shape) < 0.01] = -0.0
        f[rng.random(f.shape) < 0.002] = np.inf
        f[rng.random(f.shape) < 0.002] = np.nan
        o.field(k)[...] = f
        g.set_field(k, f)
    with np.errstate(all="ignore"):
        for n in (2, 4):
            g.sweeps(n)
            o.project_rb(stop_mode=0, max_iter=n)
            for k in range(3):
                a, b = o.field(k), g.get_field(k)
                assert np.array_equal(a, b, equal_nan=True), (n, k, int((~((a == b) | (np.isnan(a) & np.isnan(b)))).sum()))


def test_pairs_at_512_cubed_equal_single_sweeps(fsamd, monkeypatch):
    # the full size of the multi-GPU configuration: two launches per pass there (256-column tiles and a 64 x 64
    # remainder), several z-chunks; four sweeps in pairs against four one-pass sweeps, bit for bit
    import fluid_simulation_amd.scenes as scenes
    n = 512
    fields = []
    for no_pairs in ("0", "1"):
        monkeypatch.setenv("FSAMD_NO_SWEEP_PAIRS", no_pairs)
        g = fsamd.FluidSim3D(n, n, n, stop_mode=fsamd.STOP_FIXED, max_iters=1, step_mode=fsamd.STEP_REFLECTION)
        init, per_step = scenes.SCENES["vortex_ring"]
        init(g)
        per_step(g, 0)
        g.advance(1.0 / 30.0)          # a field with divergence everywhere the ring has reached
        g.sweeps(4)
        fields.append([g.get_field(k) for k in range(3)])
        g.close()
    for k in range(3):
        assert np.array_equal(fields[0][k], fields[1][k]), k


# ---- z-slabs: four halo planes, one exchange per pass -----------------------------------------------------------
@pytest.mark.parametrize("slabs", [2, 3, 5, 8])
@pytest.mark.parametrize("shape", [(72, 21, 40), (264, 30, 41), (20, 13, 67), (132, 40, 33)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_sweep_pairs_on_local_slabs_bit_exact(fsamd, shape, dtype, slabs):
    nx, ny, nz = shape
    o, g = make_pair(fsamd, nx, ny, nz, seed=nx + nz + slabs, dtype=dtype, stop_mode=fsamd.STOP_FIXED, max_iters=1, local_slabs=slabs)
    done = 0
    for n in (2, 3, 4, 1, 2):    # pairs, a single sweep after a pair, a pair after a single sweep
        g.sweeps(n)
        done += n
        o.project_rb(stop_mode=0, max_iter=n)
        assert_fields_equal(o, g, "uvw", what=f"{shape} {dtype.__name__} {slabs} slabs after {done} sweeps")


def test_slabs_of_fewer_than_four_planes_keep_the_one_pass_sweep(fsamd):
    # 23 planes in 8 slabs: two or three planes each, no room for four halo planes from one neighbour
    o, g = make_pair(fsamd, 40, 18, 23, seed=77, stop_mode=fsamd.STOP_FIXED, max_iters=1, local_slabs=8)
    with pytest.raises(fsamd.FluidSimError):
        g.time_sweep_pairs(1)
    g.sweeps(4)
    o.project_rb(stop_mode=0, max_iter=4)
    assert_fields_equal(o, g, "uvw", what="slabs of 2-3 planes")
    assert g.comm_stats().exchanges >= 4


def test_a_pass_of_two_sweeps_on_slabs_halves_the_exchanges(fsamd, monkeypatch):
    counts = {}
    fields = {}
    for no_pairs in ("1", "0"):
        monkeypatch.setenv("FSAMD_NO_SWEEP_PAIRS", no_pairs)
        _, g = make_pair(fsamd, 64, 24, 48, seed=5, stop_mode=fsamd.STOP_FIXED, max_iters=1, local_slabs=4)
        g.sweeps(1)                      # (flags and the first halos are exchanged here)
        g.comm_stats(reset=True)
        g.sweeps(12)
        counts[no_pairs] = g.comm_stats().exchanges
        fields[no_pairs] = [g.get_field(k) for k in range(3)]
    assert counts["1"] == 12
    assert counts["0"] == 7              # one for the deep planes before the first pass, then one per pass
    for k in range(3):
        assert np.array_equal(fields["0"][k], fields["1"][k])


@pytest.mark.parametrize("stop_mode,tol,check_every", [(2, 2e-5, 1), (2, 2e-5, 3), (2, 2e-5, 0), (1, 2.0, 2), (0, 0.0, 1)])
def test_projection_and_steps_on_slabs_with_and_without_pairs(fsamd, stop_mode, tol, check_every):
    cfg = dict(seed=11, scale=5.0, stop_mode=stop_mode, tol=tol, max_iters=40, check_every=check_every, step_mode=fsamd.STEP_REFLECTION)
    _, a = make_pair(fsamd, 48, 30, 40, local_slabs=3, **cfg)
    _, b = make_pair(fsamd, 48, 30, 40, sweep_kernel=fsamd.SWEEP_FUSED_LDS, **cfg)     # single domain, never pairs
    for step in range(3):
        a.advance(0.02); b.advance(0.02)
        assert a.stats().sweeps_step == b.stats().sweeps_step
        for k in range(4):
            assert np.array_equal(a.get_field(k), b.get_field(k)), (step, k)
