"""One sweep + semi_lagrange as two launches against the one launch that does both (fs3d_sweep_advect), wall clock
around synchronised batches.  usage: time_fused.py [n] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fsamd_loader
fs = fsamd_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
DT = 1.0 / 60.0
g = fs.FluidSim3D(n, n, n, stop_mode=fs.STOP_FIXED, max_iters=2, step_mode=1)
g.sphere_influence(n / 10.0)
for it in range(4):
    g.fill_sphere_influence(n / 10.0, it)
    g.advance(1.0 / 30.0)
g.synchronize()
for name in ("two launches", "one launch", "two launches", "one launch"):
    t = time.perf_counter()
    ok = True
    for _ in range(reps):
        if name == "two launches":
            g.sweeps(1); g.semi_lagrange(DT)
        else:
            _, served = g.sweep_advect(DT)
            ok = ok and served
    g.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    print(f"{n}^3 sweep + self-advection, {name}: {ms:.3f} ms per pair{'' if ok else ' (NOT SERVED)'}", flush=True)
g.close()
