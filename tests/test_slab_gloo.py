"""Multi-rank path on CPU: world_size-2 and -3 gloo process groups run the two slab exchange schedules
(two-pass: interface w plane per half-sweep merged by parity; one-pass, the production schedule: ghost planes
recomputed, two planes of u, v, w per neighbour and sweep; summed residual, gathered planes, id broadcast)
pair: the pass of two sweeps with four halo planes and one exchange per pass)
with the oracle as the per-rank arithmetic; the result must be bitwise the single-domain sweep."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("mode", ["two_pass", "one_pass", "pair"])
@pytest.mark.parametrize("world,shape", [(2, (14, 10, 16)), (3, (9, 12, 17))])
def test_slab_exchange_schedule_under_gloo(world, shape, mode):
    nx, ny, nz = shape
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "slab_gloo_worker.py"), str(nx), str(ny), str(nz), "5" if mode == "pair" else "3", mode]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "SLAB_GLOO_OK" in res.stdout


def test_slab_layout_matches_the_library_rules():
    from fsamd_loader import load
    load()
    from fluid_simulation_amd.slab import SlabLayout, split_planes, take_parity
    assert [split_planes(10, 3, p) for p in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert split_planes(512, 8, 7) == (448, 512)
    one = SlabLayout(32, 1, 0)
    assert (one.halo, one.gz0, one.nzl, one.zb, one.ze) == (0, 0, 32, 1, 31)
    lo, hi = SlabLayout(32, 2, 0), SlabLayout(32, 2, 1)
    assert (lo.gz0, lo.nzl, lo.zb, lo.ze, lo.lz_hi_interface) == (-2, 20, 3, 18, 18)
    assert (hi.gz0, hi.nzl, hi.zb, hi.ze, hi.lz_lo_interface) == (14, 20, 2, 17, 2)
    assert lo.local_plane(16) == lo.lz_hi_interface and hi.local_plane(16) == hi.lz_lo_interface
    assert take_parity(0, True) == 1 and take_parity(0, False) == 0
    with pytest.raises(ValueError):
        SlabLayout(7, 4, 3)
