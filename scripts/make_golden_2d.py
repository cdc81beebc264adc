"""Generates tests/golden/ref2d_*.npz from the reference's own 2D solver.

Runs only where /root/reference is mounted: it needs oracle/_ref/libref2d.so, which `make -C oracle ref`
compiles from the reference's sources.  The outputs are data (inputs and expected outputs), not code:
  ref2d_advance_32x48.npz    full state after 1, 2, 5, 10 advance() calls, MacCormack on (the default)
  ref2d_advance_sl_32x48.npz the same with MacCormack off and wind on
  ref2d_stages_24x36.npz     a seeded random state and the state after each single stage applied to it
  ref2d_native_128x192.npz   the reference's own resolution: density/u/v/pressure after 10 steps
"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import oracle2d as o2  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def random_state(sim, seed):
    rng = np.random.default_rng(seed)
    st = {}
    for k in (o2.DENSITY, o2.PRESSURE, o2.DIVERGENCE, o2.U, o2.V, o2.FX, o2.FY):
        a = rng.random(sim.shape(k)) * 2 - 1
        if k == o2.DENSITY:
            a = np.abs(a)
        if k in (o2.U, o2.V):
            a *= 8.0          # backtraces of several cells, so that the clamps are exercised
        if k in (o2.PRESSURE, o2.DIVERGENCE):
            a *= 0.05
        st[k] = a
    return st


def main():
    assert o2.reference_available(), "build oracle/_ref first: make -C oracle ref"
    os.makedirs(GOLD, exist_ok=True)
    for name, mac, wind in (("ref2d_advance_32x48", True, False), ("ref2d_advance_sl_32x48", False, True)):
        sim = o2.Reference2D(32, 48)
        sim.params(maccormack=mac, wind=wind)
        out = {"dt": sim.dt, "maccormack": mac, "wind": wind}
        for step in range(1, 11):
            sim.advance()
            if step in (1, 2, 5, 10):
                for k, v in sim.state().items():
                    out[f"s{step}_{k}"] = v
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    # note: addWind keeps its call counter in a function-static variable (3_fluid/FluidSim.h:123), so the
    # wind run above saw r = 1..10; it is the first and only wind run of this process

    sim = o2.Reference2D(24, 36)
    st = random_state(sim, 2025)
    out = {"dt": sim.dt}
    for k, a in st.items():
        out[f"in_{o2.FIELD_NAMES[k]}"] = a
    for op in (o2.ADD_BUOYANCY, o2.ADD_FORCE, o2.SET_NEUMANN, o2.SET_ZERO, o2.COMPUTE_DIVERGENCE, o2.COPY_BORDER,
               o2.SOLVE_POISSON, o2.CORRECT_VELOCITY, o2.COMPUTE_VORTICITY, o2.ADVECT_VALUES, o2.SOLVE_PRESSURE):
        for mac in ((True, False) if op == o2.ADVECT_VALUES else (True,)):
            sim = o2.Reference2D(24, 36)
            sim.params(maccormack=mac)
            for k, a in st.items():
                sim.set(k, a)
            sim.stage(op)
            tag = f"op{op}" + ("" if mac else "_sl")
            for k in range(8):
                out[f"{tag}_{o2.FIELD_NAMES[k]}"] = sim.get(k)
    np.savez_compressed(os.path.join(GOLD, "ref2d_stages_24x36.npz"), **out)

    sim = o2.Reference2D(128, 192)
    for _ in range(10):
        sim.advance()
    s = sim.state()
    np.savez_compressed(os.path.join(GOLD, "ref2d_native_128x192.npz"), dt=sim.dt,
                        **{k: s[k] for k in ("density", "u", "v", "pressure")})
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()
