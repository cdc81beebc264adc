"""Wall time of the per-step scene call (fill_sphere_influence) next to a step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fsamd_loader
fs = fsamd_loader.load()
for n in (256, 512):
    g = fs.FluidSim3D(n, n, n, stop_mode=0, max_iters=1)
    g.sphere_influence(n / 10.0)
    g.synchronize()
    t = time.perf_counter()
    for it in range(1, 11):
        g.fill_sphere_influence(n / 10.0, it)
    g.synchronize()
    print(n, "fill_sphere_influence ms", (time.perf_counter() - t) / 10 * 1e3, flush=True)
    g.close()
