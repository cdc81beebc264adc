"""Convergence of the multigrid V-cycle of the oracle against plain red-black sweeps (CPU)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.oracle3d import Oracle3D

def closed_box(n, seed=0):
    o = Oracle3D(n, n, n)
    rng = np.random.default_rng(seed)
    for k in range(3):
        f = rng.standard_normal((n, n, n)).astype(np.float32)
        o.field(k)[...] = f
    # zero the faces on the walls so that the problem is compatible
    u, v, w = o.field(0), o.field(1), o.field(2)
    u[:, :, :2] = 0; u[:, :, -1] = 0; v[:, :2, :] = 0; v[:, -1, :] = 0; w[:2] = 0; w[-1] = 0
    u[0] = u[-1] = 0; u[:, 0] = u[:, -1] = 0; v[0] = v[-1] = 0; v[:, :, 0] = v[:, :, -1] = 0
    w[:, 0] = w[:, -1] = 0; w[:, :, 0] = w[:, :, -1] = 0
    return o

def collision(n):
    o = Oracle3D(n, n, n)
    r = n / 10.0
    o.sphere_emitter(r, n / 2.0, r, n / 2.0, 20.0)
    o.sphere_emitter(r, n / 2.0, n - 1 - r, n / 2.0, -20.0)
    return o

def ring(n):
    o = Oracle3D(n, n, n)
    o.sphere_influence(n / 10.0)
    return o

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, make in (("closed box, random", closed_box), ("ring collision", collision), ("vortex ring", ring)):
    for alpha in (1.0, 1.5, 2.0):
        o = make(n)
        r0 = o.residual_l2()
        hist = []
        for c in range(8):
            o.mg_vcycle(2, 2, 30, alpha)
            hist.append(o.residual_l2() / r0)
        print(f"{name:20s} n={n} alpha={alpha}: V(2,2) residual ratio per cycle", " ".join(f"{h:.2e}" for h in hist), flush=True)
    o = make(n)
    r0 = o.residual_l2()
    hist = []
    for c in range(8):
        for _ in range(5):
            o.rb_sweep()
        hist.append(o.residual_l2() / r0)
    print(f"{name:20s} n={n} plain sweeps, 5 per column          ", " ".join(f"{h:.2e}" for h in hist), flush=True)
