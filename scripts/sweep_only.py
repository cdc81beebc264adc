"""Runs a handful of red-black sweeps of the n^3 vortex ring (for rocprofv3 counter passes)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import fsamd_loader
fs = fsamd_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = fs.FluidSim3D(n, n, n, stop_mode=0, max_iters=1)
g.sphere_influence(n / 10.0)
print("ms per sweep", g.time_sweeps(6) / 6)
print("ms per pass of two sweeps", g.time_sweep_pairs(6) / 6)
# kernels with exactly known traffic, to calibrate the counters on this access pattern:
# k_change_l1 reads 24 B per interior cell, k_div_sumsq reads 13 B per interior cell, both write ~nothing
for _ in range(3):
    g.gauss_seidel_projection()
    g.residual()
g.close()
