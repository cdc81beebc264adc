"""Times the lexicographic sweep (block diagonals) and the serial change sum."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fsamd_loader
fs = fsamd_loader.load()
for n in (32, 64, 128, 256):
    g = fs.FluidSim3D(n, n, n, ordering=fs.ORDER_LEXICOGRAPHIC)
    g.sphere_influence(n / 10.0)
    g.sweeps(2); g.synchronize()
    t = time.perf_counter(); g.sweeps(20); g.synchronize(); ts = (time.perf_counter() - t) / 20
    g.gauss_seidel_projection()
    t = time.perf_counter()
    for _ in range(5): g.gauss_seidel_projection()
    tc = (time.perf_counter() - t) / 5
    print(f"{n}^3: lexicographic sweep {ts*1e3:.3f} ms, sweep + snapshot + serial change {tc*1e3:.3f} ms", flush=True)
    g.close()
