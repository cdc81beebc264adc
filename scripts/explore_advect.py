"""Advection kernel timing (wall time of back-to-back calls) and a parity spot check."""
import sys, time
sys.path.insert(0,'.')
import fsamd_loader
m=fsamd_loader.load()
for n in [int(a) for a in sys.argv[1:]] or [256,512]:
    g=m.FluidSim3D(n,n,n, stop_mode=0, max_iters=20)
    g.sphere_influence(n/10.0); g.project()
    for _ in range(3): g.semi_lagrange(1/30.)
    g.synchronize(); t=time.time()
    for _ in range(10): g.semi_lagrange(1/30.)
    g.synchronize(); dt=(time.time()-t)/10
    print(f"n={n} advect {dt*1e3:.3f} ms  {n**3/dt/1e9:.2f} Gcells/s  33 B/cell -> {n**3*33/dt/1e9:.0f} GB/s", flush=True)
    g.close()
