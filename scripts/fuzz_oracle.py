"""Random configurations of the GPU step against the CPU oracle, bit for bit: sizes (with and without tiles
clear of the shell), kernel choices, both precisions, both step modes, the three stopping rules, omega, buoyancy.
(The residual rule divides by dt inside a step; the oracle's projection takes tol * dt.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fsamd_loader
fs = fsamd_loader.load()
from oracle.oracle3d import Oracle3D
from tests.util3d import random_cells, random_field

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
WIDE = len(sys.argv) > 3 and sys.argv[3] == "wide"
TALL = len(sys.argv) > 3 and sys.argv[3] == "tall"
FAST = len(sys.argv) > 3 and sys.argv[3] == "fast"
t0, runs = time.time(), 0
while time.time() - t0 < budget:
    nx, ny, nz = int(rng.integers(3, 170)), int(rng.integers(3, 36)), int(rng.integers(3, 30))
    if WIDE:      # long rows (several x tiles, both row-segment variants of the sweep), few of them
        nx, ny, nz = int(rng.integers(150, 700)), int(rng.integers(3, 12)), int(rng.integers(3, 10))
    dtype = int(rng.integers(0, 2))
    if TALL:      # many planes of few cells (several z chunks per launch, many slabs)
        nx, ny, nz = int(rng.integers(6, 40)), int(rng.integers(5, 16)), int(rng.integers(40, 400))
    cfg = dict(dtype=dtype, step_mode=int(rng.integers(0, 2)), sweep_kernel=int(rng.integers(0, 3)),
               advect_kernel=int(rng.integers(0, 2)), stop_mode=int(rng.choice([0, 1, 2])), max_iters=int(rng.integers(1, 9)),
               check_every=int(rng.choice([1, 2, 3])), tol=float(rng.choice([5.0, 1e-1, 1e-3])),
               omega=float(rng.choice([1.0, 1.0, 1.3])), advection=int(rng.integers(0, 2)))
    # the tile shape of the two-sweep kernel: the library's own choice or a forced one (read per call)
    shape = str(rng.choice(["", "64", "32", "16", "64+16", "64+32"]))
    if shape: os.environ["FSAMD_ROWS2_LANES_PER_ROW"] = shape
    else: os.environ.pop("FSAMD_ROWS2_LANES_PER_ROW", None)
    scale = float(rng.choice([1.0, 10.0, 60.0]))
    if FAST:      # backtraces of many cells: beyond the staged planes of the tiled advection, out of the domain
        scale = float(rng.choice([150.0, 400.0, 2000.0]))
    dt = float(rng.choice([1.0 / 30.0, 0.02]))
    npdt = np.float64 if dtype == 1 else np.float32
    inner = int(rng.integers(1 << 30))
    r2 = np.random.default_rng(inner)
    shape = (nz, ny, nx)
    cells = random_cells(shape, r2)
    fields = [random_field(shape, r2, npdt, scale if k < 3 else 1.0) for k in range(4)]
    fields[3] = np.abs(fields[3])
    g = fs.FluidSim3D(nx, ny, nz, **cfg)
    o = Oracle3D(nx, ny, nz, dtype=npdt)
    o.cells[...] = cells
    g.set_cell_types(cells)
    for k in range(4):
        o.field(k)[...] = fields[k]
        g.set_field(k, fields[k])
    def oproject():
        otol = cfg["tol"] * (dt if cfg["stop_mode"] == 2 else 1.0)
        return o.project_rb(omega=cfg["omega"], stop_mode=cfg["stop_mode"], tol=otol, max_iter=cfg["max_iters"],
                            check_every=cfg["check_every"])
    def advect_mc(t, q, vel):   # advection followed by the MacCormack correction and limiter over the same q, vel, t
        return o.maccormack(t, q, vel, o.advect(t, q, vel, 0))
    ok = True
    for it in range(3):
        if it and rng.integers(0, 2):      # the configuration changes between steps
            new = dict(step_mode=int(rng.integers(0, 2)), sweep_kernel=int(rng.integers(0, 3)),
                       advect_kernel=int(rng.integers(0, 2)), stop_mode=int(rng.choice([0, 1, 2])),
                       max_iters=int(rng.integers(1, 9)), check_every=int(rng.choice([1, 2, 3])),
                       tol=float(rng.choice([5.0, 1e-1, 1e-3])), omega=float(rng.choice([1.0, 1.0, 1.3])),
                       advection=int(rng.integers(0, 2)))
            new = {k: v for k, v in new.items() if rng.integers(0, 2)}
            try:
                g.set_config(**new)
                cfg.update(new)
            except fs.FluidSimError:
                g.set_config(**{k: cfg[k] for k in new})      # back to what it was
        g.advance(dt)
        if cfg["advection"] == 1:
            oproject()
            q0 = [o.field(k).copy() for k in range(4)]
            if cfg["step_mode"] == 0:
                m = advect_mc(dt, q0, q0[:3])
            else:
                m = advect_mc(0.5 * dt, q0, q0[:3])
                for k in range(4): o.field(k)[...] = m[k]
                tilde = [m[k].copy() for k in range(3)]
                oproject()
                uh = [o.field(k).copy() for k in range(3)]
                refl = [o.reflect(uh[k], tilde[k]) for k in range(3)]
                m = advect_mc(0.5 * dt, refl + [o.field(3).copy()], uh)
            for k in range(4): o.field(k)[...] = m[k]
        elif cfg["step_mode"] == 0:
            oproject(); o.semi_lagrange(dt)
        else:
            oproject()
            u0 = [o.field(k).copy() for k in range(4)]
            half = o.advect(0.5 * dt, u0, u0[:3], 0)
            for k in range(4): o.field(k)[...] = half[k]
            oproject()
            uh = [o.field(k).copy() for k in range(3)]
            refl = [o.reflect(uh[k], half[k]) for k in range(3)]
            out = o.advect(0.5 * dt, refl + [o.field(3).copy()], uh, 0)
            for k in range(4): o.field(k)[...] = out[k]
        for k in range(4):
            x, y = g.get_field(k), o.field(k)
            if not np.array_equal(x, y) and not (np.array_equal(np.isnan(x), np.isnan(y)) and
                                                 np.array_equal(np.nan_to_num(x), np.nan_to_num(y))):
                print("MISMATCH", (nx, ny, nz), cfg, "scale", scale, "dt", dt, "seed", inner, "step", it, "field", k,
                      int((x != y).sum()), flush=True)
                ok = False
        if not ok:
            break
    if ok:   # one multigrid cycle on whatever the steps left
        o.mg_vcycle(2, 2, 30, 2.0); g.mg_vcycle()
        for k in range(3):
            x, y = g.get_field(k), o.field(k)
            if not (np.array_equal(np.isnan(x), np.isnan(y)) and np.array_equal(np.nan_to_num(x), np.nan_to_num(y))):
                print("MISMATCH after a multigrid cycle", (nx, ny, nz), cfg, "scale", scale, "seed", inner, "field", k,
                      int((x != y).sum()), flush=True)
                ok = False
    g.close()
    runs += 1
    if runs % 25 == 0:
        print("  ...", runs, "configurations", flush=True)
    if not ok:
        sys.exit(1)
print("fuzz ok:", runs, "configurations", flush=True)
