import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fsamd_loader
fs = fsamd_loader.load()
from tests.util3d import random_cells, random_field
from oracle.oracle3d import Oracle3D
nx, ny, nz = 91, 34, 23
cfg = dict(dtype=0, step_mode=0, sweep_kernel=0, advect_kernel=0, advection=0, stop_mode=0, max_iters=11)
r2 = np.random.default_rng(800870221); shape = (nz, ny, nx)
cells = random_cells(shape, r2)
fields = [random_field(shape, r2, np.float32, 40.0 if k < 3 else 1.0) for k in range(4)]
a = fs.FluidSim3D(nx, ny, nz, **cfg); b = fs.FluidSim3D(nx, ny, nz, local_slabs=6, **cfg)
o = Oracle3D(nx, ny, nz); o.cells[...] = cells
for k in range(4): o.field(k)[...] = fields[k]
for g in (a, b):
    g.set_cell_types(cells)
    for k in range(4): g.set_field(k, fields[k])
# projection alone
a.project(); b.project(); o.project_rb(stop_mode=0, max_iter=11)
print("after projection equal:", [bool(np.array_equal(a.get_field(k), b.get_field(k))) for k in range(3)],
      "oracle:", [bool(np.array_equal(a.get_field(k), o.field(k))) for k in range(3)])
pre = [a.get_field(k) for k in range(4)]
a.semi_lagrange(0.02); b.semi_lagrange(0.02); o.semi_lagrange(0.02)
try:
    b.synchronize(); print("no halo error")
except fs.FluidSimError as e:
    print("halo error", e)
for k in range(4):
    x, y, z = a.get_field(k), b.get_field(k), o.field(k)
    idx = np.argwhere(x != y)
    print("field", k, "single==oracle", bool(np.array_equal(x, z)), "slab==oracle", bool(np.array_equal(y, z)), "mismatches", len(idx))
    for i in idx[:5]:
        i = tuple(i)
        print("   at (z,y,x)", i, "single", x[i], "slabs", y[i], "oracle", z[i], "pre u,v,w", pre[0][i], pre[1][i], pre[2][i], "cell", cells[i])
