"""Sweep-kernel timing at several sizes (both implementations) and a few full steps."""
import sys, time
sys.path.insert(0,'.')
import fsamd_loader
m=fsamd_loader.load()
sizes = [int(a) for a in sys.argv[1:]] or [256, 512]
for n in sizes:
    cells=n**3
    for kern,name in ((0,'fused'),(2,'fusedlds'),(1,'two_pass')):
        g=m.FluidSim3D(n,n,n, stop_mode=0, max_iters=1, sweep_kernel=kern)
        g.sphere_influence(n/10.0)
        g.time_sweeps(5); ms=g.time_sweeps(40)/40
        print(f"n={n} {name:8s} sweep {ms*1e3:8.1f} us  {cells/ms/1e6:8.2f} Gcell-sweeps/s  algorithmic(50 B/cell/sweep) {cells*50/ms/1e6:8.1f} GB/s", flush=True)
        g.close()
    g=m.FluidSim3D(n,n,n, stop_mode=2, tol=1e-6, max_iters=1000, check_every=8, step_mode=1)
    g.sphere_influence(n/10.0)
    for step in range(0):
        g.fill_sphere_influence(n/10.0, step)
        t=time.time(); g.advance(1/30.); g.synchronize(); dt=time.time()-t
        st=g.stats()
        print(f"  step {step}: sweeps {st.sweeps_step} metric {st.metric:.3e} wall {dt*1e3:.2f} ms (proj {st.ms_project:.2f}, adv {st.ms_advect:.2f}) -> {cells/dt/1e6:.1f} Mcells/s", flush=True)
    g.close()
