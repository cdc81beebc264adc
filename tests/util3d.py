"""Shared helpers of the 3D parity tests: identical seeded states in the oracle and on the GPU."""
import numpy as np

from oracle.oracle3d import Oracle3D, FLUID, OBJECT, STATIC


def random_cells(shape, rng, p_object=0.05, p_static=0.03):
    """Interior cells: mostly Fluid with scattered Object / StaticFluid cells; shell = Object."""
    nz, ny, nx = shape
    c = np.zeros(shape, dtype=np.uint8)
    r = rng.random(shape)
    c[r < p_object] = OBJECT
    c[(r >= p_object) & (r < p_object + p_static)] = STATIC
    c[0], c[-1] = OBJECT, OBJECT
    c[:, 0], c[:, -1] = OBJECT, OBJECT
    c[:, :, 0], c[:, :, -1] = OBJECT, OBJECT
    return c


def random_field(shape, rng, dtype, scale=1.0):
    """Uniform [-scale, scale] in the interior, zero on the shell (the reference never writes the shell)."""
    f = np.zeros(shape, dtype=dtype)
    f[1:-1, 1:-1, 1:-1] = (rng.random((shape[0] - 2, shape[1] - 2, shape[2] - 2)) * 2 - 1) * scale
    return f


def close_walls(fields):
    """Zero the wall-normal faces next to the shell (they can never move, main.cpp:227-232): with them
    zero and no emitter the total divergence vanishes and the projection can converge to rounding."""
    fields[0][:, :, 1] = 0
    fields[1][:, 1, :] = 0
    fields[2][1, :, :] = 0


def make_pair(fsamd, nx, ny, nz, seed=0, dtype=np.float32, scale=1.0, masks=True, closed=False, **cfg):
    """An oracle and a GPU simulation holding the same random state."""
    rng = np.random.default_rng(seed)
    shape = (nz, ny, nx)
    o = Oracle3D(nx, ny, nz, dtype=dtype)
    g = fsamd.FluidSim3D(nx, ny, nz, dtype=fsamd.F64 if dtype is np.float64 else fsamd.F32, **cfg)
    if masks:
        cells = random_cells(shape, rng)
        o.cells[...] = cells
        g.set_cell_types(cells)
    fields = [random_field(shape, rng, dtype, scale if k < 3 else 1.0) for k in range(4)]
    fields[3] = np.abs(fields[3])
    if closed:
        close_walls(fields)
    for k in range(4):
        o.field(k)[...] = fields[k]
        g.set_field(k, fields[k])
    return o, g


def assert_fields_equal(o, g, names="uvwd", what=""):
    for k, name in enumerate("uvwd"):
        if name not in names:
            continue
        a, b = o.field(k), g.get_field(k)
        if not np.array_equal(a, b):
            diff = np.abs(a.astype(np.float64) - b.astype(np.float64))
            idx = np.unravel_index(np.argmax(diff), diff.shape)
            raise AssertionError(
                f"{what}: field {name} differs: max abs {diff.max():.3e} at (z,y,x)={idx}, "
                f"{(diff > 0).sum()} of {diff.size} elements, oracle={a[idx]!r} gpu={b[idx]!r}")


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.sqrt((a * a).sum())
    return np.sqrt(((a - b) ** 2).sum()) / (den if den > 0 else 1.0)
