"""Runs a few reflection steps of the n^3 vortex ring with a fixed sweep count (for rocprofv3 counter passes)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import fsamd_loader
fs = fsamd_loader.load()
if os.environ.get("FSAMD_LIB"):
    fs.load_library(os.environ["FSAMD_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = fs.FluidSim3D(n, n, n, stop_mode=0, max_iters=2, step_mode=1)
g.sphere_influence(n / 10.0)
for it in range(6):
    g.advance(1 / 30.)
g.synchronize()
g.close()
