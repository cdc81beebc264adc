"""The north star's ordering bar: velocity after `steps` vortex-ring steps (the reference's update(): project, then
self-advect), red-black against the reference's lexicographic order, both on the GPU (the lexicographic mode is
bit-identical to the serial sweep of advection-reflection/main.cpp:235).  Relative L2 difference of (u, v, w).

    python scripts/north_star_orders.py N [steps] [--rule l1:5|l1:0.01|l2:1e-5|fixed:200] [--free K]

--free K: the compatible variant.  The ring is blown for K steps by the red-black solver, the emitter cells are
then turned into ordinary fluid (no net flux any more: the projection has a unique fixed point), the state is
handed to both orders and `steps` free steps follow.
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fsamd_loader
fs = fsamd_loader.load()
DT = 1.0 / 30.0


def parse_rule(text):
    kind, val = text.split(":")
    if kind == "l1":
        return dict(stop_mode=fs.STOP_L1_CHANGE, tol=float(val), max_iters=1000)
    if kind == "l2":
        return dict(stop_mode=fs.STOP_L2_RESIDUAL, tol=float(val), max_iters=1000)
    return dict(stop_mode=fs.STOP_FIXED, tol=0.0, max_iters=int(val))


def rel_l2(a, b):
    num = sum(((x.astype(np.float64) - y) ** 2).sum() for x, y in zip(a, b)) ** 0.5
    den = sum((x.astype(np.float64) ** 2).sum() for x in a) ** 0.5
    return num / den


def run(n, ordering, rule, steps, check_every, start=None, log=None):
    g = fs.FluidSim3D(n, n, n, ordering=ordering, check_every=check_every, **rule)
    if start is None:
        g.sphere_influence(n / 10.0)
    else:
        g.set_cell_types(start["cells"])
        for k in range(4):
            g.set_field(k, start["fields"][k])
    sweeps, t0 = 0, time.perf_counter()
    for it in range(steps):
        if start is None:
            g.fill_sphere_influence(n / 10.0, it)
        g.advance(DT)
        sweeps += g.stats().sweeps_step
        if log and it % 20 == 19:
            print(f"  .. {n}^3 ordering {ordering} step {it + 1}: {sweeps} sweeps, metric {g.stats().metric:.3e}, "
                  f"{time.perf_counter() - t0:.1f} s", flush=True)
    out = [g.get_field(k) for k in range(3)]
    l2, mx = g.residual()
    g.close()
    return out, sweeps, mx


def blown_state(n, rule, k, check_every):
    g = fs.FluidSim3D(n, n, n, check_every=check_every, **rule)
    g.sphere_influence(n / 10.0)
    for it in range(k):
        g.fill_sphere_influence(n / 10.0, it)
        g.advance(DT)
    cells = g.cells
    cells[cells == fs.CELL_STATIC_FLUID] = fs.CELL_FLUID
    st = {"cells": cells, "fields": [g.get_field(i) for i in range(4)]}
    g.close()
    return st


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]); steps = int(args[1]) if len(args) > 1 else 100
    rule_text = "l1:5"
    free = 0
    check_every = 1
    av = sys.argv[1:]
    for i, a in enumerate(av):
        if a == "--rule": rule_text = av[i + 1]
        if a == "--free": free = int(av[i + 1])
        if a == "--check-every": check_every = int(av[i + 1])
    rule = parse_rule(rule_text)
    start = blown_state(n, rule, free, check_every) if free else None
    a, sa, da = run(n, fs.ORDER_LEXICOGRAPHIC, rule, steps, check_every, start, log=True)
    b, sb, db = run(n, fs.ORDER_RED_BLACK, rule, steps, check_every, start, log=True)
    print(f"RESULT {n}^3 {'free ring after ' + str(free) + ' blown steps' if free else 'emitter on'}, rule {rule_text}, "
          f"check every {check_every}, {steps} steps: rel L2 {rel_l2(a, b):.3e}  (sweeps lexicographic {sa}, red-black {sb}; "
          f"max|div| of the final state {da:.3e} / {db:.3e})", flush=True)


if __name__ == "__main__":
    main()
