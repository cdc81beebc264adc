"""Several ranks of the z-slab path as separate processes on ONE GPU.

RCCL does not accept two ranks of a communicator on one device, so the processes bind the shared-memory
stand-in of tests/fake_rccl through FSAMD_RCCL_LIBRARY; everything above the RCCL entry points (which planes go to
which neighbour on which stream, the all-reduced stopping scalars, and, in the last test, the torch.distributed
control plane and the benchmark's launch line) is the product code, unchanged.  Red-black order is partition independent: the gathered
fields must equal the single-domain run bit for bit."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import slab_scenarios

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(argv, world, timeout=150, extra_env=None):
    """Starts `world` copies of argv with the rank environment of torch.distributed.run; returns their outputs."""
    assert os.path.exists(FAKE), "build tests/fake_rccl first (make -C tests/fake_rccl)"
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FSAMD_RCCL_LIBRARY=FAKE, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, "-u"] + argv, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True, cwd=ROOT))
    outs = []
    failed = False
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            failed = True
        outs.append(out)
        failed = failed or p.returncode != 0
    assert not failed, "\n-----\n".join(o[-3000:] for o in outs)
    return outs


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("scenario", sorted(slab_scenarios.SCENARIOS))
def test_ranks_equal_single_domain(fsamd, tmp_path, scenario, world):
    out = str(tmp_path / "ranks")
    run_ranks([os.path.join(ROOT, "tests", "ranks", "slab_rank_worker.py"), scenario, out], world)
    parts = [np.load(f"{out}.rank{r}.npz") for r in range(world)]
    want = slab_scenarios.run(fsamd, scenario)
    assert sorted(parts[0].files) == sorted(want)
    for key, val in want.items():
        val = np.asarray(val)
        if val.ndim == 3:      # owned planes of every rank, one after the other
            assert np.array_equal(np.concatenate([p[key] for p in parts], axis=0), val), key
            continue
        for p in parts[1:]:    # every rank holds the same scalars
            assert np.array_equal(p[key], parts[0][key]), key
        if key.endswith("_sweeps") or key == "res_max":
            assert np.array_equal(parts[0][key], val), key
        else:   # sums reduced in a different order
            assert np.allclose(parts[0][key], val, rtol=1e-9, atol=0), key


def compare_with_single_domain(fsamd, parts, scenario):
    want = slab_scenarios.run(fsamd, scenario)
    assert sorted(parts[0].files) == sorted(want)
    for key, val in want.items():
        val = np.asarray(val)
        if val.ndim == 3:
            assert np.array_equal(np.concatenate([p[key] for p in parts], axis=0), val), key
            continue
        for p in parts[1:]:
            assert np.array_equal(p[key], parts[0][key]), key
        if key.endswith("_sweeps") or key == "res_max":
            assert np.array_equal(parts[0][key], val), key
        else:
            assert np.allclose(parts[0][key], val, rtol=1e-9, atol=0), key


def test_reductions_on_a_communicator_of_their_own(fsamd, tmp_path):
    # FSAMD_SPLIT_COMMUNICATOR=1: the scalar all-reduces run on a copy of the communicator, beside the halo traffic
    # (the default keeps everything on one communicator)
    out = str(tmp_path / "ranks")
    run_ranks([os.path.join(ROOT, "tests", "ranks", "slab_rank_worker.py"), "vortex_ring_deferred", out], 2,
              extra_env={"FSAMD_SPLIT_COMMUNICATOR": "1"})
    compare_with_single_domain(fsamd, [np.load(f"{out}.rank{r}.npz") for r in range(2)], "vortex_ring_deferred")


@pytest.mark.parametrize("world", [2, 3])
def test_out_of_halo_backtrace_is_reported_by_every_rank(fsamd, tmp_path, world):
    # only the last rank's kernel raises the flag; every rank must fail at the same call (else the others would
    # wait for it in their next exchange), and fs3d_reset must clear it everywhere
    out = str(tmp_path / "ranks")
    run_ranks([os.path.join(ROOT, "tests", "ranks", "slab_rank_worker.py"), "oob_one_rank", out], world)
    for r in range(world):
        rec = np.load(f"{out}.rank{r}.npz")
        assert rec["residual_raised"] == 1 and rec["synchronize_raised"] == 1 and rec["after_reset_ok"] == 1, (r, dict(rec))


def test_bench_launch_line_with_two_ranks(fsamd):
    # the driver's own command for N = 2, at a small grid, both ranks on this GPU
    port = free_port()
    env = dict(os.environ, FSAMD_RCCL_LIBRARY=FAKE, FSAMD_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--grid", "96,96,96"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["grid"] == [96, 96, 96] and line["value"] > 0
    assert line["config"]["parallelism"] == "z-slabs x2"
    # the per-rank communication record: what makes a multi-GPU run diagnosable
    pr = line["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and sum(r["planes"] for r in pr) == 96
    plane_bytes = 96 * 96 * 4
    for r in pr:
        assert r["exchanges_per_step"] > 0 and r["reductions_per_step"] > 0 and r["halo_ms_per_step"] > 0
        assert r["exposed_wait_ms_per_step"] >= 0 and r["wall_ms_per_step"] > 0
    assert pr[0]["bytes_sent_lo_per_step"] == 0 and pr[1]["bytes_sent_hi_per_step"] == 0
    assert pr[0]["bytes_sent_hi_per_step"] == pr[1]["bytes_sent_lo_per_step"] > 0
    assert pr[0]["bytes_sent_hi_per_step"] % (2 * plane_bytes) == 0      # whole pairs of halo planes
    assert line["config"]["after_projection"]["max_abs_div"] > 0 and line["roofline"]["frac"] < 1.0
    # same sweep counts as the single-domain run of the same steps
    g = fsamd.FluidSim3D(96, 96, 96, stop_mode=fsamd.STOP_L2_RESIDUAL, tol=1e-6, max_iters=1000, check_every=0, step_mode=1)
    g.sphere_influence(9.6)
    sweeps = 0
    for it in range(3):
        g.fill_sphere_influence(9.6, it)
        g.advance(1.0 / 30.0)
        sweeps += g.stats().sweeps_step if it >= 1 else 0
    assert line["config"]["sweeps_per_step"] == pytest.approx(sweeps / 2)
