import sys, time
sys.path.insert(0,'.')
import numpy as np
import fsamd_loader
m=fsamd_loader.load()
for n in (256,512):
    g=m.FluidSim3D(n,n,n, stop_mode=2, tol=1e-6, max_iters=1000, check_every=4, step_mode=1)
    g.sphere_influence(n/10.0)
    ms=g.time_sweeps(10); ms=g.time_sweeps(50)
    cells=n**3
    print(f"n={n} sweep {ms/50:.4f} ms  -> {cells*50/ms/1e3*1e-6:.1f} Gcell-sweeps/s, model 50B/cell: {cells*50*50/ms/1e3/1e9:.1f} GB/s")
    g.reset(); g.sphere_influence(n/10.0)
    for step in range(6):
        g.fill_sphere_influence(n/10.0, step)
        t=time.time(); g.advance(1/30.); g.synchronize(); dt=time.time()-t
        st=g.stats()
        print(f"  step {step}: sweeps {st.sweeps_step} metric {st.metric:.3e} wall {dt*1e3:.2f} ms dev {st.ms_step:.2f} ms (proj {st.ms_project:.2f}, adv {st.ms_advect:.2f}) -> {cells/dt/1e6:.1f} Mcells/s; resid {g.residual()}")
    g.close()
