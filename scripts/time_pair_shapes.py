"""Times the tile shapes of the two-sweep kernel (64, 32, 16 lanes per row) and the library's own choice
on a list of grids: microseconds per sweep.  usage: time_pair_shapes.py [nx,ny,nz ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fsamd_loader
fs = fsamd_loader.load()
if os.environ.get("FSAMD_LIB"):      # an experimental build of the library
    fs.load_library(os.environ["FSAMD_LIB"])
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [
    (128, 128, 128), (192, 192, 192), (256, 256, 256), (384, 384, 384), (512, 512, 512), (256, 512, 256),
    (1024, 1024, 128), (64, 256, 256), (128, 256, 256), (640, 320, 320)]
for nx, ny, nz in shapes:
    g = fs.FluidSim3D(nx, ny, nz, stop_mode=fs.STOP_FIXED, max_iters=1)
    g.sphere_influence(min(nx, ny, nz) / 10.0)
    out = []
    for lanes in ("64", "32", "16", "64+16", "64+32", ""):
        if lanes: os.environ["FSAMD_ROWS2_LANES_PER_ROW"] = lanes
        else: del os.environ["FSAMD_ROWS2_LANES_PER_ROW"]
        g.time_sweep_pairs(2)
        out.append(f"{lanes or 'auto'}: {g.time_sweep_pairs(8) / 16 * 1e3:7.1f}")
    g.time_sweeps(2)
    print(f"{nx}x{ny}x{nz}  " + "  ".join(out) + f"  one-pass kernel: {g.time_sweeps(8) / 8 * 1e3:7.1f}", flush=True)
    g.close()
