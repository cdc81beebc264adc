"""What the reflection and the MacCormack correction preserve: kinetic energy and enstrophy of the 128^3
vortex ring after 60 and 120 steps (dt 1/30) under the three advection variants, same projection rule."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fsamd_loader
fs = fsamd_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
variants = (("reference step (project, semi-Lagrange)", 0, 0), ("advection-reflection", 1, 0),
            ("reference step + MacCormack", 0, 1), ("advection-reflection + MacCormack", 1, 1))
for label, step_mode, advection in variants:
    g = fs.FluidSim3D(n, n, n, stop_mode=fs.STOP_L2_RESIDUAL, tol=1e-6, max_iters=1000, check_every=0,
                      step_mode=step_mode, advection=advection)
    g.sphere_influence(n / 10.0)
    row = []
    for it in range(120):
        g.fill_sphere_influence(n / 10.0, it)
        g.advance(1.0 / 30.0)
        if it in (59, 119):
            d = g.diagnostics()
            row.append(f"step {it + 1}: E {d.kinetic_energy:.4e} enstrophy {d.enstrophy:.4e} max|u| {d.max_speed:.2f}")
    print(f"{label:40s} " + " | ".join(row), flush=True)
    g.close()
