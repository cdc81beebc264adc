"""Print the sweep count of every step for several check schedules (GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fsamd_loader
fs = fsamd_loader.load()
import fluid_simulation_amd.scenes as scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ces = [int(a) for a in sys.argv[3].split(',')] if len(sys.argv) > 3 else [1, 0, 8]
for ce in ces:
    sim = fs.FluidSim3D(n, n, n, stop_mode=fs.STOP_L2_RESIDUAL, tol=1e-6, max_iters=1000,
                        check_every=ce, step_mode=1, solver=1 if 'mg' in sys.argv else 0)
    init, per_step = scenes.SCENES["vortex_ring"]
    init(sim)
    out, ev, ms = [], [], []
    for it in range(steps):
        per_step(sim, it)
        sim.advance(1.0 / 30.0)
        out.append(sim.stats().sweeps_step)
        ev.append(sim.stats().evaluations_step)
        ms.append(round(sim.stats().ms_step, 2))
    sim.synchronize()
    print("check_every", ce, out, "residual passes", ev, "redone", sim.stats().redone, "device ms", ms, flush=True)
    sim.close()
