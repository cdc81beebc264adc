"""One projection, two visiting orders, no advection: does the sweep have ONE fixed point with the emitter?

From one identical state (the vortex ring after `warm` red-black steps of update(), emitter on) a single projection
is run to the sweeps' fixed point in the reference's lexicographic order (advection-reflection/main.cpp:235) and in
red-black order.  Printed: the relative L2 difference of the two projected velocity fields as the sweep count
grows, the change one more sweep still makes, and where the largest divergence a visit could still remove sits
(cell, its type and its neighbours' types).  A difference that stays at ~1e-4 however long the sweeps run means two
distinct fixed points (the bar "1e-4 after 100 steps" cannot be met by ANY other order at that size); a difference
at rounding level means the 100-step difference is amplification by the advection.

    python scripts/fixed_point_orders.py N [warm] [--free]      (--free: emitter cells turned into fluid first)
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fsamd_loader
fs = fsamd_loader.load()
DT = 1.0 / 30.0


def rel_l2(a, b):
    num = sum(((x.astype(np.float64) - y) ** 2).sum() for x, y in zip(a, b)) ** 0.5
    den = sum((x.astype(np.float64) ** 2).sum() for x in a) ** 0.5
    return num / den


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]); warm = int(args[1]) if len(args) > 1 else 20
    free = "--free" in sys.argv
    g = fs.FluidSim3D(n, n, n, stop_mode=fs.STOP_FIXED, max_iters=300)
    g.sphere_influence(n / 10.0)
    for it in range(warm):
        g.fill_sphere_influence(n / 10.0, it)
        g.update(DT)
    cells = g.cells
    if free:
        cells[cells == fs.CELL_STATIC_FLUID] = fs.CELL_FLUID
    start = [g.get_field(k) for k in range(4)]
    g.close()
    sims = {}
    for name, order in (("lexicographic", fs.ORDER_LEXICOGRAPHIC), ("red-black", fs.ORDER_RED_BLACK)):
        s = fs.FluidSim3D(n, n, n, ordering=order, stop_mode=fs.STOP_FIXED, max_iters=1)
        s.set_cell_types(cells)
        for k in range(4):
            s.set_field(k, start[k])
        sims[name] = s
    print(f"{n}^3 vortex ring after {warm} steps, {'emitter cells turned into fluid' if free else 'emitter on'}; "
          f"one projection from the same state in both orders", flush=True)
    done = 0
    for target in (100, 300, 1000, 3000, 10000, 30000):
        for s in sims.values():
            s.sweeps(target - done)
        done = target
        fields = {k: [s.get_field(i) for i in range(3)] for k, s in sims.items()}
        d = rel_l2(fields["lexicographic"], fields["red-black"])
        line = f"sweeps {target:6d}: rel L2 (lexicographic vs red-black) {d:.3e}"
        for name, s in sims.items():
            l2, mx = s.residual()
            line += f" | {name}: max|div| {mx:.3e}, P6 residual {l2:.3e}"
        print(line, flush=True)
    for name, s in sims.items():
        before = [s.get_field(i) for i in range(3)]
        change = s.gauss_seidel_projection()
        after = [s.get_field(i) for i in range(3)]
        print(f"{name}: L1 change of one more sweep {change:.3e}, rel L2 it makes {rel_l2(before, after):.3e}", flush=True)
    # where the divergence that is left sits
    s = sims["red-black"]
    div = np.abs(s.divergence.astype(np.float64))
    c = s.cells
    fluid_nb = np.zeros(c.shape, dtype=np.int32)
    for ax in range(3):
        for sh in (1, -1):
            fluid_nb += (np.roll(c, sh, axis=ax) == fs.CELL_FLUID)
    relaxable = (c != fs.CELL_OBJECT) & (fluid_nb > 0)
    relaxable[0] = relaxable[-1] = False; relaxable[:, 0] = relaxable[:, -1] = False; relaxable[:, :, 0] = relaxable[:, :, -1] = False
    names = {0: "Fluid", 1: "Object", 2: "StaticFluid"}
    for label, mask in (("any interior cell", np.ones_like(relaxable)), ("cells a visit relaxes", relaxable),
                        ("Fluid cells a visit relaxes", relaxable & (c == fs.CELL_FLUID)),
                        ("StaticFluid cells a visit relaxes", relaxable & (c == fs.CELL_STATIC_FLUID))):
        m = mask.copy()
        m[0] = m[-1] = False; m[:, 0] = m[:, -1] = False; m[:, :, 0] = m[:, :, -1] = False
        if not m.any():
            print(f"largest |div| over {label}: none"); continue
        dm = np.where(m, div, -1.0)
        z, y, x = np.unravel_index(np.argmax(dm), dm.shape)
        nb = [names[int(c[z, y, x - 1])], names[int(c[z, y, x + 1])], names[int(c[z, y - 1, x])], names[int(c[z, y + 1, x])],
              names[int(c[z - 1, y, x])], names[int(c[z + 1, y, x])]]
        print(f"largest |div| over {label}: {dm[z, y, x]:.4e} at (x,y,z)=({x},{y},{z}), a {names[int(c[z, y, x])]} cell; "
              f"neighbours x-,x+,y-,y+,z-,z+ = {nb}; rms over the set {np.sqrt((div[m] ** 2).mean()):.3e}", flush=True)
    for s in sims.values():
        s.close()


if __name__ == "__main__":
    main()
