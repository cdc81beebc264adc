"""One-pass sweep against the pass of two sweeps: milliseconds per sweep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fsamd_loader
fs = fsamd_loader.load()
if os.environ.get("FSAMD_LIB"):      # an experimental build of the library
    fs.load_library(os.environ["FSAMD_LIB"])
for n in [int(a) for a in sys.argv[1:]] or [128, 256, 512]:
    g = fs.FluidSim3D(n, n, n, stop_mode=fs.STOP_FIXED, max_iters=1)
    g.sphere_influence(n / 10.0)
    g.time_sweeps(4); one = g.time_sweeps(20) / 20
    cells = n ** 3
    for lanes in ("64", "32", "16"):   # tile shapes of the two-sweep kernel
        os.environ["FSAMD_ROWS2_LANES_PER_ROW"] = lanes
        g.time_sweep_pairs(2)
        print(f"{n}^3: {lanes} lanes per row: {g.time_sweep_pairs(10) / 20 * 1e3:.1f} us per sweep", flush=True)
    del os.environ["FSAMD_ROWS2_LANES_PER_ROW"]
    g.time_sweep_pairs(2); two = g.time_sweep_pairs(10) / 20
    print(f"{n}^3: one sweep per pass {one * 1e3:.1f} us per sweep ({25 * cells / one / 1e6:.0f} GB/s); two per pass "
          f"{two * 1e3:.1f} us per sweep ({12.5 * cells / two / 1e6:.0f} GB/s of HBM model traffic, {50 * cells / two / 1e6:.0f} GB/s algorithmic)", flush=True)
    g.close()
