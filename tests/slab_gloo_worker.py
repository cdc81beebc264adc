"""Worker of tests/test_slab_gloo.py: one rank = one z-slab, the oracle does the arithmetic, the
exchange schedule is the library's (one interface plane of w per half-sweep, merged by parity;
per-rank residual sums added across ranks)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import fsamd_loader  # noqa: E402

fs = fsamd_loader.load()
from fluid_simulation_amd import dist as fdist  # noqa: E402
from fluid_simulation_amd.slab import SlabLayout, take_parity  # noqa: E402
from oracle.oracle3d import Oracle3D, OBJECT  # noqa: E402
from tests.util3d import random_cells, random_field  # noqa: E402


def main():
    nx, ny, nz, sweeps = (int(v) for v in sys.argv[1:5])
    rank, world = fdist.init_control_plane("gloo")
    lay = SlabLayout(nz, world, rank)
    rng = np.random.default_rng(1234)          # every rank draws the same global state
    cells = random_cells((nz, ny, nx), rng)
    fields = [random_field((nz, ny, nx), rng, np.float32) for _ in range(4)]

    o = Oracle3D(nx, ny, lay.nzl)
    o.cells[...] = OBJECT
    for lz in range(lay.nzl):
        gz = lz + lay.gz0
        if 0 <= gz < nz:
            o.cells[lz] = cells[gz]
            for k in range(4):
                o.field(k)[lz] = fields[k][gz]

    parity = (np.add.outer(np.arange(ny), np.arange(nx)) & 1)    # (x + y) & 1

    def exchange(colour):
        for upper, peer, lz in ((True, rank - 1, lay.lz_lo_interface), (False, rank + 1, lay.lz_hi_interface)):
            if peer < 0 or peer >= world:
                continue
            theirs = fdist.exchange_plane(np.array(o.w[lz]), peer)
            take = ((parity + lz + lay.gz0) & 1) == take_parity(colour, upper)
            o.w[lz][take] = theirs[take]

    for _ in range(sweeps):
        for colour in (0, 1):
            o.rb_halfsweep_range(colour, lay.zb, lay.ze, lay.gz0)
            exchange(colour)

    # residual: per-rank sum over the owned interior planes, added across ranks
    u, v, w = o.u, o.v, o.w
    sl = slice(lay.zb, lay.ze)
    div = (u[sl, 1:-1, 2:] - u[sl, 1:-1, 1:-1]) + (v[sl, 2:, 1:-1] - v[sl, 1:-1, 1:-1]) + \
          (w[lay.zb + 1:lay.ze + 1, 1:-1, 1:-1] - w[sl, 1:-1, 1:-1])
    total = fdist.sum_over_ranks(float((div.astype(np.float64) ** 2).sum()))
    slowest = fdist.max_over_ranks(rank + 1.0)

    own = slice(lay.halo, lay.halo + lay.z1 - lay.z0)
    gathered = [fdist.gather_planes(np.array(o.field(k)[own])) for k in range(3)]
    token = fdist.broadcast_bytes(b"id-from-rank-0" if rank == 0 else None)
    assert token == b"id-from-rank-0"
    if rank == 0:
        ref = Oracle3D(nx, ny, nz)
        ref.cells[...] = cells
        for k in range(4):
            ref.field(k)[...] = fields[k]
        ref.project_rb(stop_mode=0, max_iter=sweeps)
        for k in range(3):
            assert gathered[k].shape == (nz, ny, nx)
            assert np.array_equal(gathered[k], ref.field(k)), f"slab run differs from the single domain in field {k}"
        ru, rv, rw = ref.u, ref.v, ref.w
        rdiv = (ru[1:-1, 1:-1, 2:] - ru[1:-1, 1:-1, 1:-1]) + (rv[1:-1, 2:, 1:-1] - rv[1:-1, 1:-1, 1:-1]) + \
               (rw[2:, 1:-1, 1:-1] - rw[1:-1, 1:-1, 1:-1])
        want = float((rdiv.astype(np.float64) ** 2).sum())
        assert abs(total - want) <= 1e-12 * want
        assert slowest == float(world)
        print("SLAB_GLOO_OK")
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
