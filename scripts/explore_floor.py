import sys, time
sys.path.insert(0,'.')
import numpy as np
import fsamd_loader
m=fsamd_loader.load()
for n, omega in ((256,1.0),(256,1.9),(512,1.9)):
    g=m.FluidSim3D(n,n,n, stop_mode=0, max_iters=1, omega=omega)
    g.sphere_influence(n/10.0)
    tot=0
    for k in range(8):
        g.sweeps(500); tot+=500
        ch=g.gauss_seidel_projection(); tot+=1
        l2,mx=g.residual()
        print(f"n={n} omega={omega} sweeps={tot} change={ch:.4e} l2={l2:.4e} l2/dt={l2*30:.4e} maxdiv={mx:.4e}", flush=True)
    g.close()
