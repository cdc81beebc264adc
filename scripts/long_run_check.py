"""A long vortex-ring run (density re-stamps at steps 100, 200 kick the sweep counts) under the deferred evaluation
against the every-sweep evaluation: steps whose sweep counts differ, repeated advections, final relative difference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fsamd_loader
fs = fsamd_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 250
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-6
for solver in (0, 1):
    a = fs.FluidSim3D(n, n, n, stop_mode=2, tol=tol, max_iters=1000, check_every=1, step_mode=1, solver=solver)
    b = fs.FluidSim3D(n, n, n, stop_mode=2, tol=tol, max_iters=1000, check_every=-1, step_mode=1, solver=solver)
    for g in (a, b):
        g.sphere_influence(n / 10.0)
    differ, sa, sb, worst = 0, 0, 0, 0.0
    for it in range(steps):
        for g in (a, b):
            g.fill_sphere_influence(n / 10.0, it)
            g.advance(1.0 / 30.0)
        ca, cb = a.stats().sweeps_step, b.stats().sweeps_step
        sa += ca; sb += cb
        differ += ca != cb
        worst = max(worst, b.stats().metric)
        if it % 50 == 49:
            print("  step", it + 1, "sweeps", ca, cb, flush=True)
    num = sum(float(((a.get_field(k).astype(np.float64) - b.get_field(k)) ** 2).sum()) for k in range(3)) ** 0.5
    den = sum(float((a.get_field(k).astype(np.float64) ** 2).sum()) for k in range(3)) ** 0.5
    print(f"solver {solver}: {steps} steps, sweeps every-sweep {sa} deferred {sb}, steps with another count {differ}, "
          f"repeated advections {b.stats().redone}, largest final residual {worst:.3e} (tol {tol:g}), rel L2 difference {num / den:.2e}", flush=True)
    a.close(); b.close()
