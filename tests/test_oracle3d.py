"""CPU tests of the oracle: the restatement against the known answers recorded from the reference
binary (tests/golden/ref3d_recorded_kats.json), and the invariants that follow from the reference's
code (mask rules, emitter, sampler, reflection identities)."""
import json
import os

import numpy as np
import pytest

from oracle.oracle3d import Oracle3D, FLUID, OBJECT, STATIC
from tests.util3d import close_walls, random_cells, random_field

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref3d_recorded_kats.json")))
DT = GOLD["scene"]["dt"]


def vortex_ring(n):
    o = Oracle3D(n, n, n)
    o.sphere_influence(n / 10.0)     # main.cpp:462
    return o


def test_scene_cell_counts():
    k = GOLD["cases"]["32"]["cells"]
    c = vortex_ring(32).cells
    assert ((c == FLUID).sum(), (c == OBJECT).sum(), (c == STATIC).sum()) == (k["fluid"], k["object"], k["static_fluid"])


@pytest.mark.parametrize("n", [32, 64])
def test_sweeps_per_step_match_the_reference_binary(n):
    # main.cpp:506-509: fill_sphere_influence(N/10, iter), update(dt) with the built-in rule
    o = vortex_ring(n)
    got = []
    for step in range(3):
        o.fill_sphere_influence(n / 10.0, step)
        it, _ = o.update(DT, GOLD["scene"]["change_tol"], GOLD["scene"]["max_iter"])
        got.append(it)
    assert got == GOLD["cases"][str(n)]["sweeps_per_step"]


def test_max_speed_after_five_steps():
    o = vortex_ring(32)
    for step in range(5):
        o.fill_sphere_influence(3.2, step)
        o.update(DT)
    speed = np.sqrt(o.u.astype(np.float64) ** 2 + o.v.astype(np.float64) ** 2 + o.w.astype(np.float64) ** 2).max()
    assert speed == pytest.approx(GOLD["cases"]["32"]["max_speed_after_5_steps"], abs=0.05)


def _sweeps_to_thresholds(n, thresholds):
    o = vortex_ring(n)
    out, k, it = {}, 0, 0
    while k < len(thresholds):
        ch = o.gauss_seidel_projection()
        it += 1
        while k < len(thresholds) and ch < thresholds[k]:
            out[thresholds[k]] = it
            k += 1
        assert it < 20000
    return out


def test_sweeps_to_change_thresholds_32():
    want = GOLD["cases"]["32"]["sweeps_to_change_below"]
    got = _sweeps_to_thresholds(32, [5, 1, 0.1, 0.01, 0.001])
    assert {str(k): v for k, v in got.items()} == {"5": want["5"], "1": want["1"], "0.1": want["0.1"], "0.01": want["0.01"], "0.001": want["0.001"]}


def test_sweeps_to_change_thresholds_64():
    # the two tightest thresholds (6642 / 9316 sweeps) are left to the slow test below
    want = GOLD["cases"]["64"]["sweeps_to_change_below"]
    got = _sweeps_to_thresholds(64, [5, 1])
    assert got[5] == want["5"] and got[1] == want["1"]


@pytest.mark.slow
def test_sweeps_to_change_thresholds_64_full():
    want = GOLD["cases"]["64"]["sweeps_to_change_below"]
    got = _sweeps_to_thresholds(64, [5, 1, 0.1, 0.01, 0.001])
    assert [got[t] for t in (5, 1, 0.1, 0.01, 0.001)] == [want[k] for k in ("5", "1", "0.1", "0.01", "0.001")]


# ---- invariants that follow from the reference's code ----------------------------------------------
def _random_state(shape_xyz, seed, masks=True, scale=1.0, closed=False):
    nx, ny, nz = shape_xyz
    rng = np.random.default_rng(seed)
    o = Oracle3D(nx, ny, nz)
    if masks:
        o.cells[...] = random_cells((nz, ny, nx), rng)
    fields = [random_field((nz, ny, nx), rng, np.float32, scale) for k in range(4)]
    if closed:
        close_walls(fields)
    for k in range(4):
        o.field(k)[...] = fields[k]
    return o


def test_visit_leaves_all_fluid_cell_divergence_free():
    # main.cpp:217-232 with s = 6: six faces each take div/6
    o = _random_state((8, 8, 8), 0, masks=False)
    o.rb_halfsweep(0)
    u, v, w = o.u, o.v, o.w
    div = (u[1:-1, 1:-1, 2:] - u[1:-1, 1:-1, 1:-1]) + (v[1:-1, 2:, 1:-1] - v[1:-1, 1:-1, 1:-1]) + (w[2:, 1:-1, 1:-1] - w[1:-1, 1:-1, 1:-1])
    z, y, x = np.meshgrid(np.arange(1, 7), np.arange(1, 7), np.arange(1, 7), indexing="ij")
    red_inner = ((x + y + z) % 2 == 0) & (x > 1) & (x < 6) & (y > 1) & (y < 6) & (z > 1) & (z < 6)
    assert np.abs(div[red_inner]).max() < 1e-6


def test_objects_and_shell_never_written():
    o = _random_state((14, 10, 12), 1, scale=30.0)
    before = [o.field(k).copy() for k in range(4)]
    cells = o.cells.copy()
    o.gauss_seidel_projection(); o.rb_sweep(); o.semi_lagrange(0.05)
    for k in range(3):
        assert np.array_equal(o.field(k)[cells != FLUID], before[k][cells != FLUID])
    assert np.array_equal(o.d[cells == OBJECT], before[3][cells == OBJECT])


def test_divergence_free_field_is_a_fixed_point():
    o = _random_state((12, 12, 12), 2, masks=False, closed=True)
    o.project_rb(stop_mode=0, max_iter=3000)
    u0 = o.u.copy()
    assert o.gauss_seidel_projection() < 1e-3
    assert np.abs(o.u - u0).max() < 1e-6


def test_dt_zero_advection_is_identity_on_fluid_faces():
    o = _random_state((12, 9, 10), 3)
    before = [o.field(k).copy() for k in range(4)]
    o.semi_lagrange(0.0)
    for k in range(3):
        assert np.array_equal(o.field(k), before[k])


def _rel_l2_all(a, b):
    num = sum(((a.field(k).astype(np.float64) - b.field(k)) ** 2).sum() for k in range(3))
    den = sum((a.field(k).astype(np.float64) ** 2).sum() for k in range(3))
    return np.sqrt(num / den)


def test_red_black_and_lexicographic_converge_to_the_same_field():
    # compatible problem (closed box, no emitter): the fixed point is unique, the visiting order only
    # changes the path to it -- this is the 1e-4 velocity bar of the red-black ordering
    a = _random_state((16, 16, 16), 6, closed=True, masks=False)
    b = _random_state((16, 16, 16), 6, closed=True, masks=False)
    for _ in range(1500):
        a.gauss_seidel_projection()
    b.project_rb(stop_mode=0, max_iter=1500)
    assert _rel_l2_all(a, b) < 1e-4


def test_emitter_makes_the_converged_field_order_dependent_by_its_net_flux():
    # The emitter's fixed faces inject a net flux, the system is incompatible, and every ordering settles
    # in its own limit cycle: the fields at the end of a sweep differ by about (net flux / #cells) / 6 per
    # face, which shrinks like 1/N.  The bound asserted here is that offset, not 1e-4.
    diffs = []
    for n in (16, 32):
        a = vortex_ring(n); b = vortex_ring(n)
        for _ in range(3000):
            a.gauss_seidel_projection()
        b.project_rb(stop_mode=0, max_iter=3000)
        floor = max(a.div_sumsq()[1], b.div_sumsq()[1])     # the constant residual no sweep removes
        worst = max(np.abs(a.field(k) - b.field(k)).max() for k in range(3))
        assert worst <= floor
        diffs.append(_rel_l2_all(a, b))
    assert diffs[1] < 0.5 * diffs[0]


def test_generalised_advect_reduces_to_semi_lagrange():
    o = _random_state((13, 11, 9), 4, scale=25.0)
    q = [o.field(k).copy() for k in range(4)]
    out = o.advect(1.0 / 30.0, q, q[:3], 1)
    o.semi_lagrange(1.0 / 30.0)
    for k in range(4):
        assert np.array_equal(out[k], o.field(k))


def test_reflection_identities():
    o = _random_state((10, 10, 10), 5, masks=False, closed=True)
    tilde = [o.field(k).copy() for k in range(3)]
    o.project_rb(stop_mode=0, max_iter=2000)
    proj = [o.field(k).copy() for k in range(3)]
    refl = [o.reflect(proj[k], tilde[k]) for k in range(3)]
    for k in range(3):
        # u_refl + u_tilde = 2 u_proj
        assert np.allclose(refl[k].astype(np.float64) + tilde[k], 2.0 * proj[k].astype(np.float64), atol=1e-6)
    # P(u_refl) = u_proj: the reflected field has the opposite divergence, projecting it lands on the same field
    for k in range(3):
        o.field(k)[...] = refl[k]
    o.project_rb(stop_mode=0, max_iter=2000)
    for k in range(3):
        assert np.abs(o.field(k) - proj[k]).max() < 2e-4


def test_maccormack_zero_velocity_is_identity():
    # b = f = q when nothing moves, so the correction vanishes
    o = Oracle3D(12, 10, 9)
    rng = np.random.default_rng(3)
    q = [random_field((9, 10, 12), rng, np.float32, 1.0) for _ in range(4)]
    zero = [np.zeros_like(q[0]) for _ in range(3)]
    f = o.advect(0.1, q, zero, 0)
    m = o.maccormack(0.1, q, zero, f)
    for k in range(4):
        assert np.array_equal(m[k], f[k])


def test_maccormack_limiter_and_second_order():
    # 3_fluid/FluidSim.cpp:217-389 in 3D: a smooth density profile carried by a uniform velocity.  The
    # corrected result stays inside the range of the source and is closer to the exact shift than the
    # plain semi-Lagrange result.
    n = 48
    o = Oracle3D(n, n, n)
    x = np.arange(n, dtype=np.float64)
    prof = lambda xx: 1.0 + 0.5 * np.sin(2 * np.pi * (xx - 4.0) / 20.0)
    d = np.broadcast_to(prof(x)[None, None, :], (n, n, n)).astype(np.float32)
    vel = [np.full((n, n, n), 9.0, np.float32), np.zeros((n, n, n), np.float32), np.zeros((n, n, n), np.float32)]
    q = vel + [d]
    dt = 1.0 / 30.0                      # shift 0.3 cells
    f = o.advect(dt, q, vel, 0)
    m = o.maccormack(dt, q, vel, f)
    exact = prof(x - 9.0 * dt)
    core = slice(6, n - 6)
    err_sl = np.abs(f[3][core, core, core] - exact[None, None, core]).max()
    err_mc = np.abs(m[3][core, core, core] - exact[None, None, core]).max()
    assert err_mc < 0.35 * err_sl
    assert m[3].max() <= d.max() and m[3][1:-1, 1:-1, 1:-1].min() >= 0.0


def test_threads_do_not_change_a_bit():
    # the red-black half-sweeps, the advection and the reflection run on OpenMP threads in the CPU baseline leg
    rng = np.random.default_rng(77)
    shape = (14, 12, 18)
    out = []
    for threads in (1, 3):
        o = Oracle3D(shape[2], shape[1], shape[0])
        assert o.lib.o3_threads(threads) == threads
        r = np.random.default_rng(5)
        o.cells[...] = random_cells(shape, r)
        for k in range(4):
            o.field(k)[...] = random_field(shape, r, np.float32, 3.0)
        o.project_rb(stop_mode=0, max_iter=3)
        s0 = [o.field(k).copy() for k in range(4)]
        half = o.advect(0.02, s0, s0[:3], 1)
        refl = [o.reflect(s0[k], half[k]) for k in range(3)]
        out.append([o.field(k).copy() for k in range(4)] + list(half) + refl)
        o.lib.o3_threads(1)
    for a, b in zip(*out):
        assert np.array_equal(a, b)
