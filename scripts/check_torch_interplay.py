"""Loads torch first (its bundled HIP runtime and RCCL), then the library: RCCL loopback + one small step."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
torch.cuda.set_device(0)
x = torch.ones(4, device="cuda") * 2
import fsamd_loader
fs = fsamd_loader.load()
rc = fs.load_library().fsamd_comm_selftest(0)
g = fs.FluidSim3D(32, 32, 32, device=0)
g.sphere_influence(3.2); g.advance(1 / 30.)
print("selftest rc", rc, "sweeps", g.stats().sweeps_step, "torch ok", float(x.sum()))
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "rccl" in l or "amdhip64" in l})
print("\n".join(libs))
assert rc == 0
