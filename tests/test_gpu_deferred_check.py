"""FS_CHECK_DEFERRED: the advection that follows a projection evaluates the projection's residual.

The deferred path must (1) end every projection on a verified residual <= tol, (2) leave exactly the fields
of the every-sweep check whenever both ran the same number of sweeps, (3) recover from a misprediction (drop
the advected generation, go on sweeping, advect again) without a trace in the result."""
import numpy as np
import pytest

from tests.util3d import make_pair

pytestmark = pytest.mark.gpu

CFG = dict(stop_mode=2, max_iters=200, step_mode=1)


def ring(fsamd, n, **cfg):
    g = fsamd.FluidSim3D(n, n, n, **cfg)
    g.sphere_influence(n / 10.0)
    return g


def run_pair(fsamd, a, b, steps, dt, n=None, disturb_at=()):
    """Steps both; returns per-step sweep counts; asserts bit-equal fields while the counts agree."""
    same = True
    counts = []
    for it in range(steps):
        for g in (a, b):
            if n:
                g.fill_sphere_influence(n / 10.0, it)
            if it in disturb_at:   # a kick the previous step's count cannot have foreseen
                w = g.get_field(2)
                w[w.shape[0] // 3: w.shape[0] // 2, 5:12, 5:12] += 3.0
                g.set_field(2, w)
            g.advance(dt)
        sa, sb = a.stats(), b.stats()
        counts.append((sa.sweeps_step, sb.sweeps_step))
        assert sb.metric <= b.config.tol, (it, sb.metric)
        same = same and sa.sweeps_step == sb.sweeps_step
        if same:
            for k in range(4):
                assert np.array_equal(a.get_field(k), b.get_field(k)), (it, k, counts)
            assert sb.metric == pytest.approx(sa.metric, rel=1e-10)
    return counts


@pytest.mark.parametrize("mode", [0, 1])
def test_deferred_check_equals_every_sweep_check(fsamd, mode):
    # a tolerance the ring meets after one or two sweeps per projection once it has started up
    n = 40
    cfg = dict(CFG, tol=1e-2, step_mode=mode)
    a = ring(fsamd, n, check_every=1, **cfg)
    b = ring(fsamd, n, check_every=fsamd.CHECK_DEFERRED, **cfg)
    counts = run_pair(fsamd, a, b, 14, 1.0 / 30.0, n=n)
    assert counts[-1][1] <= (2 if mode == 0 else 4), counts     # the regime the deferral is for
    assert b.stats().evaluations_step == 0 and a.stats().evaluations_step >= 1   # no residual pass of its own left
    assert sum(c[1] for c in counts) <= sum(c[0] for c in counts) + 6, counts


def test_misprediction_is_rolled_back(fsamd):
    n = 40
    cfg = dict(CFG, tol=1e-2)
    a = ring(fsamd, n, check_every=1, **cfg)
    b = ring(fsamd, n, check_every=fsamd.CHECK_DEFERRED, **cfg)
    counts = run_pair(fsamd, a, b, 12, 1.0 / 30.0, n=n, disturb_at=(8,))
    assert counts[8][0] > counts[7][0] + 1, counts          # the kick did need more sweeps
    assert counts[8][1] == counts[8][0], counts             # and the deferred run found the same count
    assert b.stats().redone >= 1                            # by repeating an advection


def test_deferred_check_random_state_with_obstacles(fsamd):
    # random masks (Object / StaticFluid cells inside), both generations of every field in play
    _, a = make_pair(fsamd, 140, 27, 14, seed=3, scale=0.05, check_every=1, tol=5e-3, **CFG)
    _, b = make_pair(fsamd, 140, 27, 14, seed=3, scale=0.05, check_every=-1, tol=5e-3, **CFG)
    run_pair(fsamd, a, b, 8, 0.02)


@pytest.mark.parametrize("slabs", [2, 3])
def test_deferred_check_in_slabs(fsamd, slabs):
    n = 40
    cfg = dict(CFG, tol=1e-2, check_every=fsamd.CHECK_DEFERRED)
    a = ring(fsamd, n, **cfg)
    b = ring(fsamd, n, local_slabs=slabs, **cfg)
    for it in range(10):
        for g in (a, b):
            g.fill_sphere_influence(n / 10.0, it)
            g.advance(1.0 / 30.0)
        assert a.stats().sweeps_step == b.stats().sweeps_step
        for k in range(4):
            assert np.array_equal(a.get_field(k), b.get_field(k)), (it, k)


def test_deferred_check_falls_back_where_it_cannot_apply(fsamd):
    # direct advection kernel, MacCormack: the adaptive schedule runs instead, same results as check_every = 0
    n = 32
    for extra in (dict(advect_kernel=1), dict(advection=1)):
        cfg = dict(CFG, tol=1e-2, **extra)
        a = ring(fsamd, n, check_every=0, **cfg)
        b = ring(fsamd, n, check_every=-1, **cfg)
        for it in range(6):
            a.advance(1.0 / 30.0); b.advance(1.0 / 30.0)
        assert a.stats().sweeps_step == b.stats().sweeps_step
        for k in range(4):
            assert np.array_equal(a.get_field(k), b.get_field(k))


def test_full_size_512_slabs_and_deferral(fsamd):
    # the multi-GPU configuration's grid: eight z-slabs on this GPU against the single domain, three benchmark
    # steps (sweeps capped at 40) with the deferred evaluation; fields bit for bit
    n = 512
    cfg = dict(stop_mode=2, tol=1e-6, max_iters=40, step_mode=1, check_every=fsamd.CHECK_DEFERRED)
    a = ring(fsamd, n, **cfg)
    b = ring(fsamd, n, local_slabs=8, **cfg)
    for it in range(3):
        for g in (a, b):
            g.fill_sphere_influence(n / 10.0, it)
            g.advance(1.0 / 30.0)
        sa, sb = a.stats(), b.stats()
        assert sa.sweeps_step == sb.sweeps_step and sb.metric == pytest.approx(sa.metric, rel=1e-9)
    for k in range(4):
        assert np.array_equal(a.get_field(k), b.get_field(k)), k
    assert a.stats().sweeps_step >= 2


@pytest.mark.parametrize("check_every", [-1, 0, 1])
def test_polled_host_record_equals_stream_synchronisation(fsamd, check_every, monkeypatch):
    # a single-GPU handle reads its residuals from a pinned record the reduction kernel writes, polling a ticket;
    # FSAMD_NO_POLL=1 waits for the stream instead: same numbers, same sweep counts, same fields
    n = 40
    cfg = dict(CFG, tol=3e-3, check_every=check_every)
    monkeypatch.delenv("FSAMD_NO_POLL", raising=False)
    a = ring(fsamd, n, **cfg)
    for it in range(5):
        a.fill_sphere_influence(n / 10.0, it)
        a.advance(1 / 30.)
    sa = a.stats()
    res_a = a.residual()
    monkeypatch.setenv("FSAMD_NO_POLL", "1")
    b = ring(fsamd, n, **cfg)
    for it in range(5):
        b.fill_sphere_influence(n / 10.0, it)
        b.advance(1 / 30.)
    sb = b.stats()
    assert (sa.sweeps_step, sa.evaluations_step, sa.metric) == (sb.sweeps_step, sb.evaluations_step, sb.metric)
    assert res_a == b.residual()
    for k in range(4):
        assert np.array_equal(a.get_field(k), b.get_field(k))
