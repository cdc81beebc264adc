import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fsamd_loader
fs = fsamd_loader.load()
for n in (48, 72):
    for tol in (1e-3, 3e-3, 1e-2):
        for mode in (0, 1):
            g = fs.FluidSim3D(n, n, n, stop_mode=fs.STOP_L2_RESIDUAL, max_iters=50, check_every=-1, step_mode=mode, tol=tol)
            g.sphere_influence(n / 10.0)
            sw = []
            for it in range(14):
                g.fill_sphere_influence(n / 10.0, it); g.advance(1 / 30.); sw.append(g.stats().sweeps_step)
            print(n, tol, mode, sw, "fused", g.stats().fused, "redone", g.stats().redone, flush=True)
            g.close()
