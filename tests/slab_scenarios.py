"""Scenarios run identically by a single-domain simulation and by the ranks of a multi-process z-slab run; each
returns a dict of owned-plane fields (3D arrays, concatenated over ranks by the caller) and scalars (which must
agree on every rank)."""
import os
import tempfile

import numpy as np

from tests.util3d import random_cells, random_field


def _create(fs, nx, ny, nz, rank, nranks, comm_id, seed=9, scale=20.0, masks=True, **cfg):
    g = fs.FluidSim3D(nx, ny, nz, rank=rank, nranks=nranks, comm_id=comm_id, device=0, **cfg)
    rng = np.random.default_rng(seed)
    shape = (nz, ny, nx)
    z0, z1 = g.z0, g.z1
    if masks:
        g.set_cell_types(random_cells(shape, rng)[z0:z1])
    fields = [random_field(shape, rng, np.float32, scale if k < 3 else 1.0) for k in range(4)]
    fields[3] = np.abs(fields[3])
    for k in range(4):
        g.set_field(k, fields[k][z0:z1])
    return g


def _snapshot(g, rec, tag):
    for k, name in enumerate("uvwd"):
        rec[f"{tag}_{name}"] = g.get_field(k)
    st = g.stats()
    rec[f"{tag}_sweeps"] = st.sweeps_step
    rec[f"{tag}_metric"] = st.metric


def steps_l1(fs, sweep_kernel=0, **kw):
    """Both step modes with the reference's stopping rule (L1 change of one sweep)."""
    g = _create(fs, 28, 18, 23, sweep_kernel=sweep_kernel, **kw)
    rec = {}
    for mode in (0, 1):
        g.set_config(step_mode=mode, tol=1.0)
        for step in range(2):
            g.advance(0.02)
            _snapshot(g, rec, f"m{mode}s{step}")
    l2, mx = g.residual()
    rec["res_l2"], rec["res_max"] = l2, mx
    g.close()
    return rec


def steps_two_pass(fs, **kw):
    return steps_l1(fs, sweep_kernel=1, **kw)


def steps_residual(fs, **kw):
    """Reflection steps stopped by the all-reduced L2 residual with the adaptive check schedule, then the
    diagnostics reductions and a restart file written by every rank."""
    g = _create(fs, 40, 22, 37, seed=4, scale=6.0, stop_mode=fs.STOP_L2_RESIDUAL, tol=2e-3, max_iters=60,
                check_every=0, step_mode=1, **kw)
    rec = {}
    for step in range(3):
        g.advance(0.02)
        _snapshot(g, rec, f"s{step}")
    d = g.diagnostics()
    rec["diag"] = np.array([getattr(d, f[0]) for f in d._fields_], dtype=np.float64)
    rec["vort"] = g.vorticity()
    # the restart file: one file, every rank writes its own planes (FSAMD_TEST_RESTART_PATH: the same path on every rank)
    path = os.environ.get("FSAMD_TEST_RESTART_PATH") or os.path.join(tempfile.mkdtemp(), "state.fs3d")
    g.save(path)
    g.residual()          # an all-reduce: every rank has written its planes when it returns
    rec["restart_file"] = np.fromfile(path, dtype=np.uint8)
    g.close()
    return rec


def steps_maccormack(fs, **kw):
    g = _create(fs, 28, 18, 23, advection=1, step_mode=1, tol=1.0, **kw)
    rec = {}
    for step in range(2):
        g.advance(0.02)
        _snapshot(g, rec, f"s{step}")
    g.close()
    return rec


def vortex_ring(fs, check_every=0, steps=3, **kw):
    """The benchmark scene (emitter in global coordinates, density re-stamp) at a small size."""
    n = 48
    g = fs.FluidSim3D(n, n, n, device=0, stop_mode=fs.STOP_L2_RESIDUAL, tol=1e-4, max_iters=40, check_every=check_every,
                      step_mode=1, **kw)
    g.sphere_influence(n / 10.0)
    rec = {}
    for it in range(steps):
        g.fill_sphere_influence(n / 10.0, it)
        g.advance(1.0 / 30.0)
        _snapshot(g, rec, f"s{it}")
    g.close()
    return rec


def vortex_ring_deferred(fs, **kw):
    """The same with the residual evaluated by the advection that follows each projection."""
    return vortex_ring(fs, check_every=-1, steps=8, **kw)


def vortex_ring_multigrid(fs, **kw):
    """The multigrid solver: every rank contributes its planes of level 1 and solves the coarse levels itself."""
    return vortex_ring(fs, check_every=-1, steps=5, solver=1, **kw)


def oob_one_rank(fs, rank=0, nranks=1, comm_id=None):
    """A backtrace that leaves the two halo planes on the upper slab only (|w| dt = 4 cells just above the
    interface): the advection does not wait for its flag, and the next call that does (a residual, then
    fs3d_synchronize_all) must fail on EVERY rank, not only on the one whose kernel raised it.  A single domain has
    no halo to leave: it returns the record every rank must produce."""
    if nranks == 1:
        return {"residual_raised": np.float64(1), "synchronize_raised": np.float64(1), "after_reset_ok": np.float64(1)}
    nx, ny, nz = 24, 16, 12 * nranks
    g = fs.FluidSim3D(nx, ny, nz, rank=rank, nranks=nranks, comm_id=comm_id, device=0, stop_mode=fs.STOP_FIXED, max_iters=1)
    w = np.zeros((nz, ny, nx), np.float32)
    w[nz - 11, 4:12, 4:20] = 200.0          # the second owned plane of the last slab; 200 * 0.02 = 4 cells down
    g.set_field(2, w[g.z0:g.z1])
    rec = {}
    g.semi_lagrange(0.02)                   # queued; nobody has looked at the flag yet
    for name, call in (("residual", g.residual), ("synchronize", g.synchronize_all)):
        try:
            call()
            rec[f"{name}_raised"] = np.float64(0)
        except fs.FluidSimError as e:
            assert "halo" in str(e)
            rec[f"{name}_raised"] = np.float64(1)
    g.reset()                               # forgets the flag on every rank
    g.semi_lagrange(0.02)
    g.synchronize()
    rec["after_reset_ok"] = np.float64(1)
    g.close()
    return rec


SPECIAL = {"oob_one_rank": oob_one_rank}

SCENARIOS = {f.__name__: f for f in (steps_l1, steps_two_pass, steps_residual, steps_maccormack, vortex_ring,
                                     vortex_ring_deferred, vortex_ring_multigrid)}


def random_configurations(fs, seed, world, count, rank=0, nranks=1, comm_ids=None):
    """`count` random configurations drawn from `seed` for a run on `world` ranks (the single domain draws the same
    ones): sizes, precision, kernels, step mode, advection scheme, stopping rule and schedule, solver. `comm_ids(i)`
    gives the communicator id of the i-th configuration."""
    rng = np.random.default_rng(seed)
    rec = {}
    for i in range(count):
        nx, ny = int(rng.integers(6, 90)), int(rng.integers(5, 30))
        nz = int(rng.integers(3 * world, 12 * world))
        cfg = dict(dtype=int(rng.integers(0, 2)), step_mode=int(rng.integers(0, 2)), sweep_kernel=int(rng.integers(0, 3)),
                   advect_kernel=int(rng.integers(0, 2)), advection=int(rng.integers(0, 2)),
                   stop_mode=int(rng.choice([0, 1, 2])), max_iters=int(rng.integers(1, 12)),
                   check_every=int(rng.choice([-1, 0, 1, 3])), tol=float(rng.choice([1.0, 1e-2, 1e-4])))
        cfg["solver"] = int(rng.integers(0, 2)) if cfg["stop_mode"] != 1 else 0
        if cfg["solver"] == 1:
            nz = 2 * world * int(rng.integers(2, 6))        # multigrid wants slab interfaces on even planes
        scale, inner = float(rng.choice([1.0, 10.0, 20.0])), int(rng.integers(1 << 30))
        dt = np.float64 if cfg["dtype"] == 1 else np.float32
        g = fs.FluidSim3D(nx, ny, nz, rank=rank, nranks=nranks, comm_id=comm_ids(i) if nranks > 1 else None, device=0, **cfg)
        r2 = np.random.default_rng(inner)
        shape = (nz, ny, nx)
        g.set_cell_types(random_cells(shape, r2)[g.z0:g.z1])
        for k in range(4):
            g.set_field(k, random_field(shape, r2, dt, scale if k < 3 else 1.0)[g.z0:g.z1])
        for step in range(3):
            g.advance(0.02)
            _snapshot(g, rec, f"c{i}s{step}")
        g.close()
    return rec


def run(fs, name, rank=0, nranks=1, comm_id=None, comm_ids=None):
    if name.startswith("random:"):       # random:<seed>:<world>:<count>
        seed, world, count = map(int, name.split(":")[1:])
        return random_configurations(fs, seed, world, count, rank=rank, nranks=nranks, comm_ids=comm_ids)
    return (SPECIAL.get(name) or SCENARIOS[name])(fs, rank=rank, nranks=nranks, comm_id=comm_id)
