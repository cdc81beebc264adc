import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fsamd_loader
fs = fsamd_loader.load()
from tests.util3d import random_cells, random_field
def run(shape3, slabs, scale, seed=5, **cfg):
    nx, ny, nz = shape3
    a = fs.FluidSim3D(nx, ny, nz, **cfg); b = fs.FluidSim3D(nx, ny, nz, local_slabs=slabs, **cfg)
    r2 = np.random.default_rng(seed); shape = (nz, ny, nx)
    cells = random_cells(shape, r2)
    fields = [random_field(shape, r2, np.float32, scale if k < 3 else 1.0) for k in range(4)]
    for g in (a, b):
        g.set_cell_types(cells)
        for k in range(4): g.set_field(k, fields[k])
    a.advance(0.02); b.advance(0.02)
    try:
        b.synchronize()
    except fs.FluidSimError as e:
        return "halo error"
    bad = []
    for k in range(4):
        x, y = a.get_field(k), b.get_field(k)
        if not np.array_equal(x, y):
            idx = np.argwhere(x != y)
            bad.append((k, len(idx), sorted(set(idx[:, 0].tolist()))[:12]))
    return bad or "equal"
cfg = dict(dtype=0, step_mode=0, sweep_kernel=0, advect_kernel=1, advection=1, stop_mode=1, max_iters=11, check_every=-1, tol=1e-4, solver=0)
seed = 800870221
print("as found:", run((91, 34, 23), 6, 40.0, seed=seed, **cfg), flush=True)
for name, change in (("tiled advect", dict(advect_kernel=0)), ("semi-Lagrange", dict(advection=0)), ("tiled + semi-Lagrange", dict(advect_kernel=0, advection=0)),
                     ("fixed 11 sweeps", dict(stop_mode=0)), ("check_every 1", dict(check_every=1)), ("two-pass sweeps", dict(sweep_kernel=1))):
    print(name, run((91, 34, 23), 6, 40.0, seed=seed, **dict(cfg, **change)), flush=True)
for slabs in (2, 3, 4, 5):
    print("slabs", slabs, run((91, 34, 23), slabs, 40.0, seed=seed, **cfg), flush=True)
