"""What the reflection and the MacCormack correction preserve when nothing is forced: a vortex ring is blown
for 40 steps by the reference's emitter, the emitter cells are then turned into ordinary fluid, and the ring travels
freely for 160 more steps (dt 1/30, multigrid projection to 1e-6).  Kinetic energy and enstrophy relative to the
moment the emitter stops."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fsamd_loader
fs = fsamd_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
# (label, step mode, advection scheme, sub-steps per 1/30 s)
variants = (("reference step (project, semi-Lagrange)", 0, 0, 1), ("reference step, two half steps (equal cost)", 0, 0, 2),
            ("advection-reflection", 1, 0, 1),
            ("reference step + MacCormack", 0, 1, 1), ("reference + MacCormack, two half steps", 0, 1, 2),
            ("advection-reflection + MacCormack", 1, 1, 1))
start = None
for label, step_mode, advection, sub in variants:
    g = fs.FluidSim3D(n, n, n, stop_mode=fs.STOP_L2_RESIDUAL, tol=1e-6, max_iters=1000, check_every=0, solver=1,
                      step_mode=1, advection=0)
    g.sphere_influence(n / 10.0)
    for it in range(40):          # the same start for every variant: blown with the plain reflection step
        g.advance(1.0 / 30.0)
    cells = g.cells
    cells[cells == 2] = 0
    g.set_cell_types(cells)
    g.set_config(step_mode=step_mode, advection=advection)
    d0 = g.diagnostics()
    row = []
    for it in range(160):
        for _ in range(sub):
            g.advance(1.0 / 30.0 / sub)
        if it in (39, 79, 159):
            d = g.diagnostics()
            row.append(f"+{it + 1}: E {d.kinetic_energy / d0.kinetic_energy:.3f} Z {d.enstrophy / d0.enstrophy:.3f}")
    print(f"{label:40s} " + " | ".join(row), flush=True)
    g.close()
