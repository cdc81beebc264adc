"""Sweeps per step of the small vortex ring under check_every = 1 and the deferred check, for a few tolerances."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fsamd_loader
fs = fsamd_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for tol in [float(t) for t in sys.argv[2:]] or [1e-3, 3e-3, 1e-2]:
    for ce in (1, -1):
        g = fs.FluidSim3D(n, n, n, stop_mode=2, max_iters=200, step_mode=1, tol=tol, check_every=ce)
        g.sphere_influence(n / 10.0)
        c, e = [], []
        for it in range(16):
            g.fill_sphere_influence(n / 10.0, it)
            g.advance(1.0 / 30.0)
            c.append(g.stats().sweeps_step)
            e.append(g.stats().evaluations_step)
        print(tol, ce, c, "passes", e, "redone", g.stats().redone, g.stats().metric, flush=True)
