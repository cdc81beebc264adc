"""Random configurations of the slab path against the single domain, bit for bit (a search for corner cases:
slabs of two or three planes, odd sizes, every kernel choice, both precisions, the deferred evaluation, multigrid).
One configuration in three also writes a restart file after the first step and continues it on a third simulation
with yet another slab count; diagnostics (sums to 1e-12, the maximum exactly) and block-averaged density are compared
at the end."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fsamd_loader
fs = fsamd_loader.load()
from tests.util3d import random_cells, random_field

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
WIDE = len(sys.argv) > 3 and sys.argv[3] == "wide"
TALL = len(sys.argv) > 3 and sys.argv[3] == "tall"
t0, runs, skipped, restarts = time.time(), 0, 0, 0
tmp = tempfile.mkdtemp()
while time.time() - t0 < budget:
    nx, ny, nz = int(rng.integers(6, 150)), int(rng.integers(5, 40)), int(rng.integers(6, 48))
    if WIDE:      # long rows (several x tiles, both row-segment variants of the sweep), few of them
        nx, ny, nz = int(rng.integers(150, 700)), int(rng.integers(5, 12)), int(rng.integers(6, 20))
    slabs = int(rng.integers(2, 17 if TALL else 9))
    if TALL:      # many planes of few cells (several z chunks per launch, many slabs)
        nx, ny, nz = int(rng.integers(6, 40)), int(rng.integers(5, 16)), int(rng.integers(40, 400))
    cfg = dict(dtype=int(rng.integers(0, 2)), step_mode=int(rng.integers(0, 2)), sweep_kernel=int(rng.integers(0, 3)),
               advect_kernel=int(rng.integers(0, 2)), advection=int(rng.integers(0, 2)),
               stop_mode=int(rng.choice([0, 1, 2])), max_iters=int(rng.integers(1, 12)),
               check_every=int(rng.choice([-1, 0, 1, 3])), tol=float(rng.choice([1.0, 1e-2, 1e-4])))
    cfg["solver"] = int(rng.integers(0, 2)) if cfg["stop_mode"] != 1 else 0
    scale = float(rng.choice([1.0, 10.0, 40.0]))
    try:
        a = fs.FluidSim3D(nx, ny, nz, **cfg)
        b = fs.FluidSim3D(nx, ny, nz, local_slabs=slabs, **cfg)
    except fs.FluidSimError:
        skipped += 1      # e.g. too many slabs for nz, multigrid with slabs on odd planes
        continue
    dt = np.float64 if cfg["dtype"] == 1 else np.float32
    inner = int(rng.integers(1 << 30))
    r2 = np.random.default_rng(inner)
    shape = (nz, ny, nx)
    cells = random_cells(shape, r2)
    fields = [random_field(shape, r2, dt, scale if k < 3 else 1.0) for k in range(4)]
    for g in (a, b):
        g.set_cell_types(cells)
        for k in range(4):
            g.set_field(k, fields[k])
    ok = True
    c = None
    try:
        for it in range(3):
            if it and rng.integers(0, 2):      # the configuration changes between steps, on every simulation alike
                new = dict(step_mode=int(rng.integers(0, 2)), sweep_kernel=int(rng.integers(0, 3)),
                           advect_kernel=int(rng.integers(0, 2)), advection=int(rng.integers(0, 2)),
                           stop_mode=int(rng.choice([0, 1, 2])), max_iters=int(rng.integers(1, 12)),
                           check_every=int(rng.choice([-1, 0, 1, 3])), tol=float(rng.choice([1.0, 1e-2, 1e-4])),
                           solver=int(rng.integers(0, 2)))
                new = {k: v for k, v in new.items() if rng.integers(0, 2)}
                sims = [g for g in (a, b, c) if g is not None]
                try:
                    for g in sims:
                        g.set_config(**new)
                    cfg.update(new)
                except fs.FluidSimError:
                    for g in sims:
                        g.set_config(**{k: cfg[k] for k in new})      # back to what it was
            a.advance(0.02); b.advance(0.02)
            b.synchronize()    # a backtrace beyond the slab halo is reported here (and is no mismatch)
            if c is not None:
                c.advance(0.02); c.synchronize()
            if a.stats().sweeps_step != b.stats().sweeps_step:
                ok = False
            for k in range(4):
                want = a.get_field(k)
                if not np.array_equal(want, b.get_field(k)):
                    ok = False
                if c is not None and not np.array_equal(want, c.get_field(k)):
                    print("RESTART MISMATCH field", k, flush=True)
                    ok = False
            if c is not None and c.stats().sweeps_step != a.stats().sweeps_step:
                print("RESTART MISMATCH sweeps", c.stats().sweeps_step, a.stats().sweeps_step, flush=True)
                ok = False
            if it == 0 and ok and rng.integers(0, 3) == 0:
                try:
                    c = fs.FluidSim3D(nx, ny, nz, local_slabs=int(rng.integers(1, 6)), **cfg)
                except fs.FluidSimError:
                    c = None
                if c is not None:
                    path = os.path.join(tmp, "state.fs3d")
                    (a if rng.integers(0, 2) else b).save(path)
                    c.load(path)
                    restarts += 1
            if not ok:
                print("MISMATCH", (nx, ny, nz), slabs, cfg, scale, "seed", inner, "step", it, a.stats().sweeps_step, b.stats().sweeps_step, flush=True)
                break
        if ok:
            da, db = a.diagnostics(), b.diagnostics()
            for name in ("kinetic_energy", "density_sum", "enstrophy"):
                x, y = getattr(da, name), getattr(db, name)
                # the density of this fuzz is signed: the bound of a reordered sum is relative to the sum of magnitudes
                size = float(np.abs(a.get_field(3).astype(np.float64)).sum()) if name == "density_sum" else max(abs(x), abs(y))
                if not (abs(x - y) <= 1e-12 * size or (np.isnan(x) and np.isnan(y))):
                    print("DIAGNOSTICS MISMATCH", name, x, y, flush=True)
                    ok = False
            if not (da.max_speed == db.max_speed or (np.isnan(da.max_speed) and np.isnan(db.max_speed))):
                print("DIAGNOSTICS MISMATCH max_speed", da.max_speed, db.max_speed, flush=True)
                ok = False
            try:
                small = b.downsample_density(2)
            except fs.FluidSimError:
                small = None       # slab boundaries on odd planes
            if small is not None and not np.array_equal(small, a.downsample_density(2)):
                print("DOWNSAMPLE MISMATCH", flush=True)
                ok = False
            if not ok:
                print("MISMATCH after the steps", (nx, ny, nz), slabs, cfg, scale, "seed", inner, flush=True)
    except fs.FluidSimError as e:
        if "halo" not in str(e):
            print("ERROR", (nx, ny, nz), slabs, cfg, scale, e, flush=True)
            ok = False
    a.close(); b.close()
    if c is not None:
        c.close()
    runs += 1
    if runs % 25 == 0:
        print("  ...", runs, "configurations", flush=True)
    if not ok:
        sys.exit(1)
print("fuzz ok:", runs, "configurations,", restarts, "continued from a restart file,", skipped, "rejected at creation", flush=True)
