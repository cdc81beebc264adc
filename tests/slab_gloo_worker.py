"""Worker of tests/test_slab_gloo.py: one rank = one z-slab, the oracle does the arithmetic, the
exchange schedule is the library's.  Mode `two_pass`: one interface plane of w per half-sweep, merged by
parity.  Mode `one_pass` (the production schedule of the one-pass sweep): per full sweep a slab applies the
first colour to its planes and to the two ghost planes below and one above, the second colour to its planes
and one ghost plane below (its halo planes go stale, its owned planes come out exact), and then hands its
two outermost owned planes of u, v and w to each neighbour's halo.  Mode `pair` (the schedule of the pass of two
sweeps): four halo planes either side; per pass the four half-sweeps of two sweeps run on ranges that shrink by one
plane per half-sweep (the halo planes go stale ring by ring, the owned planes come out exact) and ONE exchange of
four planes of u, v, w per neighbour follows; an odd last sweep runs the one-pass schedule on the inner two halo
planes.  In all modes the per-rank residual sums are added across ranks."""
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
    mode = sys.argv[5] if len(sys.argv) > 5 else "two_pass"
    rank, world = fdist.init_control_plane("gloo")
    lay = SlabLayout(nz, world, rank, halo=4 if mode == "pair" else 2)
    rng = np.random.default_rng(1234)          # every rank draws the same global state
    cells = random_cells((nz, ny, nx), rng)
    fields = [random_field((nz, ny, nx), rng, np.float32) for _ in range(4)]

    # The library keeps a flag byte per cell that says which neighbours are Fluid, taken from the owner of every
    # plane; the oracle looks the neighbours' cell types up instead, so the slab gets one more plane of cell types
    # either side than the library holds (the visit of the outermost halo plane reads the type below it).
    pad = 1
    nzl, gz0 = lay.nzl + 2 * pad, lay.gz0 - pad
    zb, ze = lay.zb + pad, lay.ze + pad
    o = Oracle3D(nx, ny, nzl)
    o.cells[...] = OBJECT
    for lz in range(nzl):
        gz = lz + gz0
        if 0 <= gz < nz:
            o.cells[lz] = cells[gz]
            if pad <= lz < nzl - pad:
                for k in range(4):
                    o.field(k)[lz] = fields[k][gz]

    parity = (np.add.outer(np.arange(ny), np.arange(nx)) & 1)    # (x + y) & 1

    def exchange(colour):
        for upper, peer, lz in ((True, rank - 1, lay.lz_lo_interface + pad), (False, rank + 1, lay.lz_hi_interface + pad)):
            if peer < 0 or peer >= world:
                continue
            theirs = fdist.exchange_plane(np.array(o.w[lz]), peer)
            take = ((parity + lz + gz0) & 1) == take_parity(colour, upper)
            o.w[lz][take] = theirs[take]

    exchanges = [0]

    def exchange_planes(depth):
        # my `depth` outermost owned planes of u, v, w go into the neighbour's halo, theirs into mine
        # (fs3d.hip, one_pass_driver / launch_rows2: one grouped send/receive per neighbour and pass)
        H, own = lay.halo, lay.z1 - lay.z0
        exchanges[0] += 1
        for peer, mine_at, halo_at in ((rank - 1, pad + H, pad + H - depth), (rank + 1, pad + H + own - depth, pad + H + own)):
            if peer < 0 or peer >= world:
                continue
            mine = np.stack([np.array(o.field(k)[mine_at:mine_at + depth]) for k in range(3)])
            theirs = fdist.exchange_plane(mine, peer)
            for k in range(3):
                o.field(k)[halo_at:halo_at + depth] = theirs[k]

    def exchange_two_planes():
        exchange_planes(2)

    def ranges(extra_lo, extra_hi):
        return zb - (extra_lo if lay.has_lo else 0), ze + (extra_hi if lay.has_hi else 0)

    todo = sweeps
    while mode == "pair" and todo > 0:
        if todo >= 2:
            # fs_sweep_rows2.hpp: the four half-sweeps of a pass visit [c0-4, c1+3), [c0-3, c1+2), [c0-2, c1+1), [c0-1, c1)
            for k, colour in enumerate((0, 1, 0, 1)):
                a, b = ranges(4 - k, 3 - k)
                o.rb_halfsweep_range(colour, a, b, gz0)
            exchange_planes(4)
            todo -= 2
        else:
            o.rb_halfsweep_range(0, *ranges(2, 1), gz0)
            o.rb_halfsweep_range(1, *ranges(1, 0), gz0)
            exchange_planes(2)
            todo -= 1
    if mode == "pair":
        assert exchanges[0] == (sweeps + 1) // 2

    for _ in range(sweeps if mode != "pair" else 0):
        if mode == "one_pass":
            # the plane ranges of the kernel (fs3d.hip, fused_ranges): first colour on [zb-2, ze+1), second on [zb-1, ze)
            o.rb_halfsweep_range(0, zb - (2 if lay.has_lo else 0), ze + (1 if lay.has_hi else 0), gz0)
            o.rb_halfsweep_range(1, zb - (1 if lay.has_lo else 0), ze, gz0)
            exchange_two_planes()
            continue
        for colour in (0, 1):
            o.rb_halfsweep_range(colour, zb, ze, gz0)
            exchange(colour)

    # residual: per-rank sum over the owned interior planes, added across ranks
    u, v, w = o.u, o.v, o.w
    sl = slice(zb, ze)
    div = (u[sl, 1:-1, 2:] - u[sl, 1:-1, 1:-1]) + (v[sl, 2:, 1:-1] - v[sl, 1:-1, 1:-1]) + \
          (w[zb + 1:ze + 1, 1:-1, 1:-1] - w[sl, 1:-1, 1:-1])
    total = fdist.sum_over_ranks(float((div.astype(np.float64) ** 2).sum()))
    slowest = fdist.max_over_ranks(rank + 1.0)

    own = slice(pad + lay.halo, pad + lay.halo + lay.z1 - lay.z0)
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
