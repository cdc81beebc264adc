"""A few launches of the fused sweep + advection kernel (for rocprofv3 counter passes).  usage: fused_only.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fsamd_loader
fs = fsamd_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
g = fs.FluidSim3D(n, n, n, stop_mode=fs.STOP_FIXED, max_iters=2, step_mode=1)
g.sphere_influence(n / 10.0)
for it in range(3):
    g.fill_sphere_influence(n / 10.0, it)
    g.advance(1.0 / 30.0)
for _ in range(4):
    _, served = g.sweep_advect(1.0 / 60.0)
    assert served
g.synchronize()
g.close()
