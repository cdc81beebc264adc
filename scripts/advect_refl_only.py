"""Runs a few reflection steps of the n^3 vortex ring with one sweep per projection (for rocprofv3 counter passes:
both advection kernels of the step, the reflected one included)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import fsamd_loader
fs = fsamd_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = fs.FluidSim3D(n, n, n, stop_mode=0, max_iters=1, step_mode=1)
g.sphere_influence(n / 10.0)
for it in range(4):
    g.advance(1 / 30.)
g.synchronize()
g.close()
