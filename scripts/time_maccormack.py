"""Steps of the vortex ring with MacCormack advection (for rocprofv3 --kernel-trace --stats: the time of k_maccormack)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fsamd_loader
fs = fsamd_loader.load()
if os.environ.get("FSAMD_LIB"):      # an experimental build of the library
    fs.load_library(os.environ["FSAMD_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
g = fs.FluidSim3D(n, n, n, stop_mode=fs.STOP_FIXED, max_iters=2, step_mode=1, advection=1)
g.sphere_influence(n / 10.0)
for it in range(6):
    g.fill_sphere_influence(n / 10.0, it)
    g.advance(1 / 30.)
st = g.stats()
print(f"{n}^3 MacCormack reflection step: {st.ms_step:.3f} ms on device", flush=True)
g.close()
