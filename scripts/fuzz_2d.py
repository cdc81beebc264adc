"""Random configurations of the 2D fp64 solver on the GPU against the 2D CPU oracle, bit for bit: sizes from the
smallest the solver accepts, random states and forces, time steps, Poisson accuracy and iteration caps, wind,
both advection schemes, both Poisson orderings; whole advance() calls and single stages.
With a third argument "ref" the oracle is compared with the reference's own FluidSim compiled into oracle/_ref
instead (CPU only, where that build exists; the reference's lexicographic ordering only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle2d as o2
AGAINST_REF = len(sys.argv) > 3 and sys.argv[3] == "ref"
if not AGAINST_REF:
    import fsamd_loader
    fs = fsamd_loader.load()

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
STAGES = (o2.ADD_BUOYANCY, o2.ADD_FORCE, o2.SET_NEUMANN, o2.SET_ZERO, o2.COMPUTE_DIVERGENCE, o2.COPY_BORDER,
          o2.SOLVE_POISSON, o2.CORRECT_VELOCITY, o2.COMPUTE_VORTICITY, o2.ADVECT_VALUES, o2.SOLVE_PRESSURE)
t0, runs, said = time.time(), 0, 0
ref_wind_calls = 0      # the reference counts addWind calls per process (FluidSim.h:123), the oracle per simulation
while time.time() - t0 < budget:
    rx, ry = int(rng.integers(4, 200)), int(rng.integers(4, 200))
    red_black = bool(rng.integers(0, 2)) and not AGAINST_REF
    par = dict(dt=float(rng.choice([0.0, 0.005, 0.02, 0.1])), acc=float(rng.choice([0.0, 1e-3, 1e-7])),
               iters=int(rng.choice([0, 1, 7, 60])), wind=bool(rng.integers(0, 2)), maccormack=bool(rng.integers(0, 2)))
    scale = float(rng.choice([0.0, 1.0, 15.0, 300.0]))
    ora = o2.Oracle2D(rx, ry)
    ora.params(**par)
    if AGAINST_REF:
        sim = o2.Reference2D(rx, ry)
        sim.params(**par)
        ora.set_wind_calls(ref_wind_calls)
    else:
        try:
            sim = fs.FluidSim2D(rx, ry)
        except fs.FluidSimError:
            continue
        sim.params(ordering=fs.ORDER_RED_BLACK if red_black else fs.ORDER_LEXICOGRAPHIC, **par)
    if scale:
        for k in (o2.DENSITY, o2.PRESSURE, o2.DIVERGENCE, o2.U, o2.V, o2.FX, o2.FY):
            a = (rng.random(ora.shape(k)) * 2 - 1) * (scale if k in (o2.U, o2.V) else 1.0)
            ora.set(k, a); sim.set(k, a)
    what = f"{rx}x{ry} red_black={red_black} scale={scale} {par}"
    def same(tag, fields=range(8)):
        for k in fields:
            got, want = sim.get(k), ora.get(k)
            if not np.array_equal(got, want):
                print(f"MISMATCH {what}: {tag}: {o2.FIELD_NAMES[k]} max abs diff {np.nanmax(np.abs(got - want)):.3e}", flush=True)
                sys.exit(1)
    if rng.integers(0, 3) == 0 and not red_black:
        for op in rng.permutation(STAGES)[:4]:
            ora.stage(int(op)); sim.stage(int(op))
            same(f"stage {int(op)}")
    for step in range(int(rng.integers(1, 4))):
        ora.advance(red_black); sim.advance()
        ref_wind_calls += int(AGAINST_REF and par["wind"])
        same(f"step {step}", range(6))
        if not AGAINST_REF and sim.info()["last_iters"] != ora.last_iters:
            print(f"MISMATCH {what}: step {step}: iterations {sim.info()['last_iters']} != {ora.last_iters}", flush=True)
            sys.exit(1)
    if not AGAINST_REF:
        sim.close()
    runs += 1
    if time.time() - t0 > said + 20:
        said = time.time() - t0
        print(f"  ... {runs} configurations", flush=True)
print(f"fuzz ok: {runs} configurations")
