"""Velocity after 100 vortex-ring steps: red-black against the reference's lexicographic order (both on the
GPU; the lexicographic mode is bit-identical to the serial sweep).  Relative L2 difference of (u,v,w)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fsamd_loader
fs = fsamd_loader.load()

def run(n, ordering, stop_mode, tol, steps, step_mode):
    g = fs.FluidSim3D(n, n, n, ordering=ordering, stop_mode=stop_mode, tol=tol, max_iters=1000, check_every=1,
                      step_mode=step_mode)
    g.sphere_influence(n / 10.0)
    sweeps = 0
    for it in range(steps):
        g.fill_sphere_influence(n / 10.0, it)
        g.advance(1.0 / 30.0)
        sweeps += g.stats().sweeps_step
        if it % 10 == 9:
            print(f"  .. {n}^3 ordering {ordering} step {it + 1} sweeps so far {sweeps}", flush=True)
    out = [g.get_field(k).astype(np.float64) for k in range(3)]
    g.close()
    return out, sweeps

rules = (("reference rule (L1 change <= 5)", fs.STOP_L1_CHANGE, 5.0), ("L1 change <= 0.01", fs.STOP_L1_CHANGE, 0.01))
if "--rule-only" in sys.argv:
    rules = rules[:1]
for n in [int(a) for a in sys.argv[1:] if a.isdigit()] or [32, 64]:
    for label, stop_mode, tol in rules:
        a, sa = run(n, fs.ORDER_LEXICOGRAPHIC, stop_mode, tol, 100, 0)
        b, sb = run(n, fs.ORDER_RED_BLACK, stop_mode, tol, 100, 0)
        num = sum(((x - y) ** 2).sum() for x, y in zip(a, b)) ** 0.5
        den = sum((x ** 2).sum() for x in a) ** 0.5
        print(f"{n}^3 {label}: rel L2 {num / den:.3e}  (sweeps lexicographic {sa}, red-black {sb})", flush=True)
