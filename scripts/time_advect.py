"""Times the two advection kernels of the reflection step (and the reference step's semi_lagrange) with HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fsamd_loader
fs = fsamd_loader.load()
if os.environ.get("FSAMD_LIB"):      # an experimental build of the library
    fs.load_library(os.environ["FSAMD_LIB"])
for n in [int(a) for a in sys.argv[1:]] or [256, 512]:
    g = fs.FluidSim3D(n, n, n, stop_mode=fs.STOP_FIXED, max_iters=30, step_mode=1)
    g.sphere_influence(n / 10.0)
    for it in range(6):
        g.advance(1 / 30.)      # a developed ring
    cells = n ** 3
    out = []
    for label, refl, bpc in (("self", False, 33.0), ("reflected", True, 45.0)):
        g.time_advections(2, 1 / 60., refl)
        ms = g.time_advections(8, 1 / 60., refl) / 8
        out.append(f"{label} {ms:.3f} ms ({bpc * cells / ms / 1e6:.0f} GB/s, {bpc * cells / ms / 1e6 / 8000:.2f} of peak)")
    st = None
    g.set_config(stop_mode=fs.STOP_L2_RESIDUAL, tol=1e-6, max_iters=1000, check_every=-1)
    for it in range(6, 12):
        g.fill_sphere_influence(n / 10.0, it); g.advance(1 / 30.)
    st = g.stats()
    print(f"{n}^3: " + "; ".join(out) + f"; full step {st.ms_step:.3f} ms at {st.sweeps_step} sweeps", flush=True)
    g.close()
