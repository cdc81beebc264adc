"""Sweep counts of the two projections of every reflection step (GPU), and the evaluations each step needed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fsamd_loader
fs = fsamd_loader.load()
import fluid_simulation_amd.scenes as scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for ce in (1, -1):
    sim = fs.FluidSim3D(n, n, n, stop_mode=fs.STOP_L2_RESIDUAL, tol=1e-6, max_iters=1000, check_every=ce, step_mode=1)
    init, per_step = scenes.SCENES["vortex_ring"]
    init(sim)
    first, second, ev = [], [], []
    for it in range(steps):
        per_step(sim, it)
        sim.advance(1.0 / 30.0)
        st = sim.stats()
        first.append(st.sweeps_step - st.sweeps); second.append(st.sweeps); ev.append(st.evaluations_step)
    print("check_every", ce, "\n first ", first, "\n second", second, "\n evals ", ev, "redone", sim.stats().redone, flush=True)
    sim.close()
