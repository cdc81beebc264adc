#!/usr/bin/env python3
"""Benchmark of the 3D advection-reflection fluid step on MI355X.

    python bench.py --gpus 1 --steps K --warmup W            one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full fs3d_advance(): projection, half advection, projection, reflection, half advection
of the synthetic vortex-ring scene, with the projection run to the L2 residual of
3_fluid/FluidSim.cpp:17-43 <= 1e-6 or the reference's cap of 1000 sweeps
(advection-reflection/main.cpp:108); the residual is evaluated from one sweep before the count the
previous step needed and then after every sweep (the reference evaluates it after every sweep); where
that count was small and steady, the last evaluation is left to the advection kernel that follows, which
adds up the residual of the field it reads: every projection still ends on a verified residual <= tol,
and a miss is redone with more sweeps.

Workloads (BASELINE.json configs):
  N = 1   "Vortex ring 256^3 on one MI355X, GS to residual 1e-6, fp32"        (configs[1])
  N > 1   "Vortex ring 512^3 z-slab decomposed across 2/4/8 MI355X"           (configs[2]; strong scaling)

The printed JSON line carries
  roofline          the dominant kernel of the step.  On one GPU that is k_rb_rows2, which applies TWO full red-black
                    sweeps in one pass over HBM wherever no evaluation of the stopping rule separates them; else
                    (odd nx, slabs of fewer than four planes, or a step of very few sweeps) the one-pass sweep k_rb_rows.  `achieved` = the bytes
                    one launch has to move (25 B per cell: every face and flag read once, every face written once) /
                    its average duration, measured with HIP events on the library's stream.  The reference's
                    formulation touches every face once per colour (50 B per cell and sweep): that figure and the
                    blocking factor stand beside it as `algorithmic_*`.  The other of the two sweep kernels is
                    reported as roofline_one_sweep_pass / roofline_two_sweep_pass;
                    velocity_copy_us / launch_over_copy: the time of a kernel that only streams the three velocity fields
                    from one generation to the other (the fastest plain copy of three fields found: one 16-byte
                    element of each per thread, no loop), measured in the same run - the floor of any pass;
  roofline_advect   the same for the two advection kernels of the step (33 and 45 B per cell);
  cpu_baseline      the CPU oracle in the reference's own execution model (one thread, lexicographic sweeps,
                    serial-order advection) on a bounded sample, and `cpu_baseline_openmp`, its red-black loops on
                    the host cores this process may use.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, datasheet peak
SWEEP_BYTES_PER_CELL = 50.0    # algorithmic bytes of one full red-black sweep in velocity form: each of the
                               # two colours reads and writes every face once (2 x (12 + 12)) + 2 x 1 flag byte
ONE_PASS_BYTES_PER_CELL = 25.0   # what the one-pass kernel has to move: both colours share one read and one
                                 # write of every face (temporal blocking factor 2)
ADVECT_BYTES_PER_CELL = 33.0     # self-advection: u, v, w, d and the flag byte in, u, v, w, d out
REFLECTED_BYTES_PER_CELL = 45.0  # reflected advection: unprojected + projected velocity, d, flag in; u, v, w, d out
DT = 1.0 / 30.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=0, help="cubic grid size override")
    ap.add_argument("--solver", choices=["sweeps", "multigrid"], default="sweeps",
                    help="sweeps: red-black Gauss-Seidel only (the north star's solver); multigrid: V-cycles around them")
    ap.add_argument("--local-slabs", type=int, default=1,
                    help="cut each rank's planes into this many z-slabs on its own GPU (rehearses the slab path)")
    ap.add_argument("--no-multigrid-leg", action="store_true", help="skip the extra run with the multigrid solver")
    ap.add_argument("--no-512-leg", action="store_true", help="skip the extra single-GPU run of the 512^3 configuration")
    ap.add_argument("--no-matched-leg", action="store_true",
                    help="skip the extra 512^3 run whose tolerance gives the same RMS divergence per cell as 256^3 at 1e-6")
    ap.add_argument("--no-steady-leg", action="store_true", help="skip the extra 256^3 run with warm-up 10, steps 20")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tol", type=float, default=1e-6)
    ap.add_argument("--max-iters", type=int, default=1000)
    ap.add_argument("--check-every", type=int, default=-1,
                    help="n: residual after every n-th sweep; 0: adaptive schedule; -1: adaptive, and the last evaluation of a "
                         "short, steady projection is made by the advection that follows it")
    ap.add_argument("--omega", type=float, default=1.0)
    ap.add_argument("--step-mode", type=int, default=1, help="0 reference step, 1 advection-reflection")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32", help="f32 = the reference's precision")
    ap.add_argument("--advection", choices=["semi_lagrange", "maccormack"], default="semi_lagrange",
                    help="maccormack: the correction and limiter of 3_fluid/FluidSim.cpp:217-389 after every advection")
    ap.add_argument("--scene", choices=["vortex_ring", "ring_collision", "smoke_plume"], default="vortex_ring")
    ap.add_argument("--grid", type=str, default="", help="nx,ny,nz override (e.g. 256,512,256 for the plume)")
    return ap.parse_args()


def host_threads():
    # the box exports its CPU share as OMP_NUM_THREADS; never more than the CPUs there are
    return max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1), os.cpu_count() or 1))


def cpu_baseline(n, sweeps_per_step, sample_sweeps=None, threads=None):
    """The CPU oracle's order-independent loops (red-black half-sweeps, advection, reflection) on OpenMP threads, on a
    bounded sample of the same workload: some sweeps and the two advections + reflection of one step, scaled to the
    sweep count the GPU run needed per step."""
    from oracle.oracle3d import Oracle3D, build_native
    native = build_native()
    o = Oracle3D(n, n, n, native=native)
    if threads is None:
        threads = host_threads()
    threads = o.lib.o3_threads(threads)
    o.sphere_influence(n / 10.0)
    cells = n ** 3
    if sample_sweeps is None:
        sample_sweeps = max(2, min(60, int(5.0 * threads / (cells * 7e-9))))   # a few seconds of sweeps at ~7 ns per cell and thread
    o.project_rb(stop_mode=0, max_iter=1)   # first touch, thread start-up
    t = time.perf_counter()
    o.project_rb(stop_mode=0, max_iter=sample_sweeps)
    t_sweep = (time.perf_counter() - t) / sample_sweeps
    s0 = [o.field(k).copy() for k in range(4)]
    t = time.perf_counter()
    half = o.advect(0.5 * DT, s0, s0[:3], 0)
    refl = [o.reflect(s0[k], half[k]) for k in range(3)]
    o.advect(0.5 * DT, refl + [half[3]], s0[:3], 0)
    t_adv = time.perf_counter() - t
    t_step = t_sweep * sweeps_per_step + t_adv
    return {
        "value": cells / t_step / 1e6, "unit": "Mcells/s", "cores": threads, "kind": "port",
        "sample": f"{sample_sweeps} red-black sweeps ({t_sweep * 1e3:.1f} ms each) + 2 half advections + reflection "
                  f"({t_adv:.2f} s) of the same {n}^3 vortex ring on {threads} threads, scaled to {sweeps_per_step:.0f} sweeps per step",
        "ms_per_sweep": t_sweep * 1e3,
        "build": "-O3 -march=native -ffp-contract=off" if native else "-O2 -ffp-contract=off",
    }


def cpu_baseline_reference_order(n, sweeps_per_step, sample_sweeps=None):
    """What the reference executes (advection-reflection/main.cpp:189-254, :256-291): one thread, gauss_seidel_projection()
    = snapshot + lexicographic sweep + L1 change, and the serial-order semi_lagrange(); the step is priced as the GPU
    run's sweeps per step of the former plus two of the latter (the two half advections)."""
    from oracle.oracle3d import Oracle3D, build_native
    native = build_native()
    o = Oracle3D(n, n, n, native=native)
    o.lib.o3_threads(1)
    o.sphere_influence(n / 10.0)
    cells = n ** 3
    if sample_sweeps is None:
        sample_sweeps = max(2, min(40, int(3.0 / (cells * 17e-9))))   # about three seconds at ~17 ns per cell
    o.gauss_seidel_projection()   # first touch
    t = time.perf_counter()
    for _ in range(sample_sweeps):
        o.gauss_seidel_projection()
    t_sweep = (time.perf_counter() - t) / sample_sweeps
    t = time.perf_counter()
    o.semi_lagrange(0.5 * DT)
    t_adv = time.perf_counter() - t
    t_step = t_sweep * sweeps_per_step + 2 * t_adv
    return {
        "value": cells / t_step / 1e6, "unit": "Mcells/s", "cores": 1, "kind": "port",
        "sample": f"{sample_sweeps} lexicographic sweeps with snapshot and L1 change ({t_sweep * 1e3:.1f} ms each) + one "
                  f"serial-order semi_lagrange ({t_adv:.2f} s, counted twice) of the same {n}^3 vortex ring on 1 thread, "
                  f"scaled to {sweeps_per_step:.0f} sweeps per step",
        "ms_per_sweep": t_sweep * 1e3, "mcell_sweeps_per_s": cells / t_sweep / 1e6,
        "build": "-O3 -march=native -ffp-contract=off" if native else "-O2 -ffp-contract=off",
    }


def after_projection(sim, sweeps, nx, ny, nz):
    """What a projection of this depth leaves behind, measured on the final state: `sweeps` more sweeps, then the
    divergence of the field in the three norms (the reference's sqrt(sum)/#cells, the size-independent RMS, the max)."""
    k = max(1, int(round(sweeps)))
    sim.sweeps(k)
    l2, mx = sim.residual()
    ninterior = float(nx - 2) * (ny - 2) * (nz - 2)
    return {"diagnostic_sweeps": k, "p6_residual": l2 / DT, "rms_div": l2 * ninterior ** 0.5, "max_abs_div": mx}


def run_steps(sim, scene_init, scene_step, warmup, steps, barrier):
    scene_init(sim)
    it = 0
    for _ in range(warmup):
        scene_step(sim, it); it += 1
        sim.advance(DT)
    barrier()
    before = sim.stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        scene_step(sim, it); it += 1
        sim.advance(DT)          # (no query between steps: fs3d_get_stats drains the stream)
    barrier()
    elapsed = time.perf_counter() - t0
    st = sim.stats()
    return elapsed, st.sweeps_total - before.sweeps_total, st.evaluations_total - before.evaluations_total


def extra_leg(fs, scenes, a, nx, ny, nz, device, solver, note, tol=None, rank=0, world=1, comm_id=None, reduce_max=None,
              warmup=None, steps=None):
    """The same steps on a fresh state with another solver, grid, tolerance or window, reported beside the headline value."""
    tol = a.tol if tol is None else tol
    warmup = a.warmup if warmup is None else warmup
    steps = a.steps if steps is None else steps
    sim = fs.FluidSim3D(nx, ny, nz, rank=rank, nranks=world, comm_id=comm_id, device=device,
                        dtype=fs.F64 if a.dtype == "f64" else fs.F32,
                        stop_mode=fs.STOP_L2_RESIDUAL, tol=tol, max_iters=a.max_iters, check_every=a.check_every,
                        omega=a.omega, step_mode=a.step_mode, solver=solver, buoyancy=1.0 if a.scene == "smoke_plume" else 0.0)
    scene_init, scene_step = scenes.SCENES[a.scene]

    def barrier():
        sim.synchronize()
        if reduce_max:
            reduce_max(0.0)
    elapsed, sweeps, evals = run_steps(sim, scene_init, scene_step, warmup, steps, barrier)
    if reduce_max:
        elapsed = reduce_max(elapsed)
    st = sim.stats()
    leg = {"grid": [nx, ny, nz], "value": nx * ny * nz / (elapsed / steps) / 1e6, "unit": "Mcells/s",
           "ms_per_step": elapsed / steps * 1e3, "sweeps_per_step": sweeps / steps, "residual_passes_per_step": evals / steps,
           "warmup": warmup, "steps": steps, "tol": tol, "last_residual": st.metric, "note": note}
    if solver == 0:
        leg["after_projection"] = after_projection(sim, sweeps / steps / 2.0, nx, ny, nz)
    sim.close()
    return leg


def step_bytes_per_cell(sweeps_per_step, esz_scale):
    """Bytes the kernels of one reflection step have to move per cell: self-advection 33, reflected advection 45,
    one-pass sweeps 25 each (fp32; the 4-byte parts double for fp64)."""
    def sc(b, flags=1.0):
        return (b - flags) * esz_scale + flags
    return sc(ADVECT_BYTES_PER_CELL) + sc(REFLECTED_BYTES_PER_CELL) + sc(ONE_PASS_BYTES_PER_CELL) * sweeps_per_step


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
        a.gpus = world

    # RCCL across processes needs dmabuf IPC on this driver stack (already exported on the boxes; kept here so that a
    # bare shell gives the same run)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import fsamd_loader
    fs = fsamd_loader.load()

    device = local_rank
    if os.environ.get("FSAMD_BENCH_SHARE_GPU") == "1":
        # rehearsal of the launch line on a machine with fewer GPUs than ranks (needs a transport that accepts
        # several ranks per device, see tests/fake_rccl); never a measurement
        device = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)   # control plane only

    def new_comm_id():
        if world == 1:
            return None
        box = [fs.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def reduce_max(v):
        if world == 1:
            return v
        tt = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    import fluid_simulation_amd.scenes as scenes
    n = a.n or (256 if world == 1 else 512)
    nx, ny, nz = (int(v) for v in a.grid.split(",")) if a.grid else (n, n, n)
    sim = fs.FluidSim3D(nx, ny, nz, rank=rank, nranks=world, comm_id=new_comm_id(), device=device,
                        dtype=fs.F64 if a.dtype == "f64" else fs.F32,
                        stop_mode=fs.STOP_L2_RESIDUAL, tol=a.tol, max_iters=a.max_iters,
                        check_every=a.check_every, omega=a.omega, step_mode=a.step_mode, local_slabs=a.local_slabs, solver=1 if a.solver == "multigrid" else 0,
                        advection=1 if a.advection == "maccormack" else 0,
                        buoyancy=1.0 if a.scene == "smoke_plume" else 0.0)
    scene_init, scene_step = scenes.SCENES[a.scene]

    def barrier():
        torch.cuda.synchronize()
        sim.synchronize()
        if world > 1:
            dist.barrier()

    if world > 1:
        sim.comm_timing(True)   # an event pair around every exchange and wait: a few microseconds per sweep, stated in the line

    scene_init(sim)
    it = 0
    for _ in range(a.warmup):
        scene_step(sim, it); it += 1
        sim.advance(DT)
    barrier()
    if world > 1:
        sim.comm_stats(reset=True)
    before = sim.stats()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        scene_step(sim, it); it += 1
        sim.advance(DT)          # (no query between steps: fs3d_get_stats drains the stream)
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = elapsed * 1e3       # this rank's own wall time of the timed steps (the max over ranks is `elapsed` below)
    elapsed = reduce_max(elapsed)
    st = sim.stats()
    sweeps = st.sweeps_total - before.sweeps_total
    evals = st.evaluations_total - before.evaluations_total
    per_rank = None
    if world > 1:
        cs = sim.comm_stats()
        mine = {"rank": rank, "planes": sim.shape[0], "wall_ms_per_step": dev_ms / a.steps,
                "halo_ms_per_step": cs.exchange_ms / a.steps, "exposed_wait_ms_per_step": cs.wait_ms / a.steps,
                "reduction_host_ms_per_step": cs.reduction_host_ms / a.steps,
                "exchanges_per_step": cs.exchanges / a.steps, "waits_per_step": cs.waits / a.steps,
                "reductions_per_step": cs.reductions / a.steps,
                "bytes_sent_lo_per_step": cs.bytes_sent_lo / a.steps, "bytes_sent_hi_per_step": cs.bytes_sent_hi / a.steps}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    # the kernels of the step, timed live with HIP events on the library's stream
    esz_scale = 2.0 if a.dtype == "f64" else 1.0   # 8-byte faces
    cells_local = sim.shape[0] * ny * nx
    nsw = 20
    sim.time_sweeps(4)
    sweep_s = sim.time_sweeps(nsw) * 1e-3 / nsw           # one launch per full sweep
    one_pass_bytes = (ONE_PASS_BYTES_PER_CELL - 1.0) * esz_scale + 1.0
    algorithmic_bytes = (SWEEP_BYTES_PER_CELL - 2.0) * esz_scale + 2.0
    # two consecutive sweeps that no evaluation separates run as ONE pass over HBM (k_rb_rows2)
    pair_s = None
    if nx % 2 == 0 and a.solver == "sweeps" and os.environ.get("FSAMD_NO_SWEEP_PAIRS") != "1":
        npair = 10
        try:    # (z-slabs run the pass too where every slab owns at least four planes: its time then includes the exchange)
            sim.time_sweep_pairs(2)
            pair_s = sim.time_sweep_pairs(npair) * 1e-3 / npair   # one launch = two full sweeps
        except fs.FluidSimError:
            pair_s = None
    # the yardstick of a sweep pass: the three velocity fields streamed from one generation of the arrays to the other
    sim.time_field_copies(2)
    ncopy = 10
    copy_s = sim.time_field_copies(ncopy) * 1e-3 / ncopy
    adv = None
    if a.advection == "semi_lagrange":
        nadv = 6
        sim.time_advections(2, 0.5 * DT, False)
        adv_s = sim.time_advections(nadv, 0.5 * DT, False) * 1e-3 / nadv
        sim.time_advections(2, 0.5 * DT, True)
        refl_s = sim.time_advections(nadv, 0.5 * DT, True) * 1e-3 / nadv
        adv_b = (ADVECT_BYTES_PER_CELL - 1.0) * esz_scale + 1.0
        refl_b = (REFLECTED_BYTES_PER_CELL - 1.0) * esz_scale + 1.0
        adv = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "self": {"kernel": "k_advect_tiled<SEP=false> (u, v, w, d along the field itself)", "bytes_per_cell": adv_b,
                        "launch_us": adv_s * 1e6, "achieved": adv_b * cells_local / adv_s / 1e9,
                        "frac": adv_b * cells_local / adv_s / 1e9 / HBM_PEAK_GBS},
               "reflected": {"kernel": "k_advect_tiled<SEP=true, REFL=true> (reflection formed while staging)",
                             "bytes_per_cell": refl_b, "launch_us": refl_s * 1e6,
                             "achieved": refl_b * cells_local / refl_s / 1e9,
                             "frac": refl_b * cells_local / refl_s / 1e9 / HBM_PEAK_GBS},
               "note": "copy_floor_us: the measured time of a plain copy of the velocity (velocity_copy_us) scaled to the bytes "
                       "this kernel has to move; launch_over_floor says how far above it the kernel runs"}
        for key, nbytes in (("self", adv_b), ("reflected", refl_b)):
            adv[key]["copy_floor_us"] = copy_s * 1e6 * nbytes / (24.0 * esz_scale)
            adv[key]["launch_over_floor"] = adv[key]["launch_us"] / adv[key]["copy_floor_us"]
    barrier()

    line = None
    if rank == 0:
        cells = nx * ny * nz
        ms_step = elapsed / a.steps * 1e3
        sweeps_per_step = sweeps / a.steps
        achieved = one_pass_bytes * cells_local / sweep_s / 1e9
        # a projection of K sweeps between two evaluations is K // 2 passes of two sweeps and K % 2 of one
        paired = pair_s is not None and sweeps_per_step >= 4
        bpc = step_bytes_per_cell(sweeps_per_step * (0.5 if paired else 1.0), esz_scale)
        line = {
            "metric": "full-step Mcells/s (advect + reflect + projection to residual 1e-6)",
            "value": cells / (elapsed / a.steps) / 1e6,
            "unit": "Mcells/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": a.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{a.scene.replace('_', ' ')} {nx}x{ny}x{nz}, dt 1/30, "
                            f"{'advection-reflection' if a.step_mode == 1 else 'reference'} step"
                            f"{' with MacCormack advection' if a.advection == 'maccormack' else ''}, red-black GS omega {a.omega:g}"
                            f"{' inside multigrid V(2,2) cycles' if a.solver == 'multigrid' else ''}, "
                            f"stop L2 residual <= {a.tol:g} or {a.max_iters} sweeps, "
                            + ("residual evaluated from one sweep before the previous step's count, then every sweep"
                               if a.check_every == 0 else
                               "residual evaluated from one sweep before the previous step's count, then every sweep; the last "
                               "evaluation of a projection whose count was the same (<= 4) in the last two steps is made "
                               "by the advection kernel that reads its result (redone with more sweeps if the residual "
                               "exceeds the tolerance)"
                               if a.check_every == -1 else f"checked every {a.check_every}"),
                "grid": [nx, ny, nz], "parallelism": f"z-slabs x{world}" + (f" (x{a.local_slabs} slabs per GPU)" if a.local_slabs > 1 else ""),
                "sweeps_per_step": sweeps_per_step, "residual_passes_per_step": evals / a.steps,
                "redone_advections": st.redone, "last_residual": st.metric,
                "residual_norm": "sqrt(sum (div/dt)^2) / #interior cells, the norm of 3_fluid/FluidSim.cpp:25-43; it shrinks "
                                 "with the grid (x N^-1.5 at equal RMS divergence per cell), see after_projection.rms_div",
                "device_ms_until_last_projection": st.ms_project, "device_ms_last_advection": st.ms_advect,
            },
            "roofline_one_sweep_pass" if paired else "roofline": {
                "bound": "hbm", "kernel": f"k_rb_rows<{'double' if a.dtype == 'f64' else 'float'}> (one-pass red-black sweep)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "bytes_per_cell": one_pass_bytes, "bytes_per_launch": one_pass_bytes * cells_local, "launch_us": sweep_s * 1e6,
                "basis": "bytes one launch has to move (every face and flag read once, every face written once)",
                # the reference's formulation (one pass per colour) beside it: the one-pass kernel does the work of 50 B per cell
                "algorithmic_bytes_per_cell": algorithmic_bytes, "blocking_factor": 2,
                "algorithmic_rate": algorithmic_bytes * cells_local / sweep_s / 1e9,
                # the whole step against the same peak, on the bytes its kernels have to move
                "full_step_bytes_per_cell": bpc,
                "full_step_frac": bpc * cells / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS / world,
                "full_step_algorithmic_bytes_per_cell": 25 + 36 + 45 + 50 * sweeps_per_step,
            },
        }
        if pair_s is not None:
            pa = one_pass_bytes * cells_local / pair_s / 1e9
            pair = {
                "bound": "hbm", "kernel": f"k_rb_rows2<{'double' if a.dtype == 'f64' else 'float'}> (two red-black sweeps in one pass)",
                "achieved": pa, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": pa / HBM_PEAK_GBS, "traffic": None,
                "bytes_per_cell": one_pass_bytes, "bytes_per_launch": one_pass_bytes * cells_local, "launch_us": pair_s * 1e6,
                "sweeps_per_launch": 2, "us_per_sweep": pair_s * 1e6 / 2,
                "basis": "bytes one launch has to move (every face and flag read once, every face written once); the launch does the "
                         "work of two sweeps; see launch_over_copy for its distance from a plain copy of the velocity",
                "algorithmic_bytes_per_cell": 2 * algorithmic_bytes, "blocking_factor": 4,
                "algorithmic_rate": 2 * algorithmic_bytes * cells_local / pair_s / 1e9,
                "full_step_bytes_per_cell": bpc,
                "full_step_frac": bpc * cells / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS / world,
                "full_step_algorithmic_bytes_per_cell": 25 + 36 + 45 + 50 * sweeps_per_step,
            }
            # the dominant kernel of the step is the one most sweeps run in
            line["roofline" if paired else "roofline_two_sweep_pass"] = pair
        # what the memory system gives a kernel that only copies the velocity (24 of the 25 bytes per cell of a pass), measured
        # in this run: a pass cannot be faster, whatever the datasheet peak
        copy_rate = 24.0 * esz_scale * cells_local / copy_s / 1e9
        for key in ("roofline", "roofline_one_sweep_pass", "roofline_two_sweep_pass"):
            if key in line:
                line[key]["velocity_copy_us"] = copy_s * 1e6
                line[key]["velocity_copy_rate"] = copy_rate
                line[key]["launch_over_copy"] = line[key]["launch_us"] / (copy_s * 1e6)
        if adv:
            line["roofline_advect"] = adv
        pmc = os.path.join(ROOT, "profiles", "sweep_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc)).get(str(n))
                if rec and world == 1 and a.dtype == "f32" and not a.grid:
                    # counters cannot be read inside this run: the committed record of the same kernel and grid
                    one = line["roofline_one_sweep_pass" if paired else "roofline"]
                    one["traffic"] = rec["hbm_bytes_per_launch"]
                    one["traffic_rate"] = rec["hbm_bytes_per_launch"] / sweep_s / 1e9
                    one["traffic_source"] = "profiles/sweep_traffic.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE of " \
                                                         f"{rec.get('kernel', 'the kernel')}, collected {rec.get('collected', 'earlier')}; not measured in this run)"
                    rec2 = json.load(open(pmc)).get(f"{n}_pairs")
                    if rec2 and pair_s is not None:
                        two = line["roofline" if paired else "roofline_two_sweep_pass"]
                        two["traffic"] = rec2["hbm_bytes_per_launch"]
                        two["traffic_rate"] = rec2["hbm_bytes_per_launch"] / pair_s / 1e9
                        two["traffic_source"] = f"profiles/sweep_traffic.json ({rec2.get('kernel')}, collected {rec2.get('collected')}; not measured in this run)"
            except Exception:
                pass
        if a.solver == "sweeps":
            line["config"]["after_projection"] = None   # filled below, by every rank together
    # a projection's leftovers on the final state (collective with several ranks)
    if a.solver == "sweeps":
        ap_ = after_projection(sim, sweeps / a.steps / 2.0, nx, ny, nz)
        if rank == 0:
            line["config"]["after_projection"] = ap_
    sim.close()
    if world > 1 and rank == 0:
        line["per_rank"] = per_rank
        line["comm_note"] = "halo_ms: device time of the grouped send/receive exchanges on the exchange stream; exposed_wait_ms: " \
                            "device time the compute stream waited for them; measured with an event pair around every exchange " \
                            "and wait, inside the timed region"

    default_scene = a.scene == "vortex_ring" and not a.n and not a.grid and a.dtype == "f32" and a.solver == "sweeps" \
        and a.advection == "semi_lagrange" and a.local_slabs == 1
    big = (512, 512, 512)
    # the size-independent reading of the tolerance: the P6 norm divides by the cell count, so at 512^3 a tolerance of
    # 1e-6 allows sqrt(8) times the RMS divergence per cell it allows at 256^3; this tolerance allows the same
    matched_tol = a.tol * (254.0 ** 3 / 510.0 ** 3) ** 0.5
    if world > 1 and default_scene and (nx, ny, nz) == big and not a.no_matched_leg:
        leg = extra_leg(fs, scenes, a, *big, device, 0,
                        "vortex ring 512^3 on all ranks, tolerance scaled so that the RMS divergence per cell equals that of "
                        "256^3 at 1e-6: the projection is then tens of sweeps per step and the sweep exchange is what scales",
                        tol=matched_tol, rank=rank, world=world, comm_id=new_comm_id(), reduce_max=reduce_max)
        if rank == 0:
            line["rms_matched_512"] = leg
    if rank == 0:
        if world == 1 and a.solver == "sweeps" and a.local_slabs == 1 and not a.no_multigrid_leg:
            line["solver_multigrid"] = extra_leg(
                fs, scenes, a, nx, ny, nz, device, 1,
                "FS_SOLVER_MULTIGRID: V(2,2) cycles around the same red-black sweeps, 2x2x2 coarsening, "
                "piecewise-constant transfer, same stopping rule, a cycle counted as 4 sweeps")
        if world == 1 and default_scene and not a.no_steady_leg:
            # the timed window of the headline (warm-up 2, 5 steps) lies in the start-up transient of the ring, where a
            # projection needs up to 57 sweeps and its count is not yet predictable; this is the measurement protocol of
            # a developed flow: warm-up 10 steps, 20 timed steps
            line["steady_state_256"] = extra_leg(fs, scenes, a, nx, ny, nz, device, 0,
                                                 "the headline configuration past its start-up transient: warm-up 10 steps, 20 timed steps",
                                                 warmup=10, steps=20)
        if not a.no_512_leg and default_scene:
            # the configuration the multi-GPU lines are quoted on, on this rank's GPU alone (the other ranks wait):
            # with N = 1 it sits beside the 256^3 headline, with N > 1 it is the base of the strong-scaling ratio
            leg = extra_leg(fs, scenes, a, *big, device, 0, "vortex ring 512^3 on one GPU, same steps and stopping rule")
            leg["hbm_roofline_frac_full_step"] = (step_bytes_per_cell(leg["sweeps_per_step"], 1.0) * big[0] * big[1] * big[2]
                                                  / (leg["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS)
            leg["hbm_roofline_frac_full_step_algorithmic"] = ((25 + 36 + 45 + 50 * leg["sweeps_per_step"]) * big[0] * big[1] * big[2]
                                                              / (leg["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS)
            line["one_gpu_512"] = leg
            if world > 1 and (nx, ny, nz) == big:
                line["speedup_vs_one_gpu"] = line["value"] / leg["value"]
            if not a.no_matched_leg:
                m = extra_leg(fs, scenes, a, *big, device, 0,
                              "vortex ring 512^3 on one GPU, tolerance scaled so that the RMS divergence per cell equals that "
                              "of 256^3 at 1e-6", tol=matched_tol)
                # (its sweeps run two per pass over HBM)
                m["hbm_roofline_frac_full_step"] = (step_bytes_per_cell(m["sweeps_per_step"] * (0.5 if m["sweeps_per_step"] >= 4 else 1.0), 1.0)
                                                    * big[0] * big[1] * big[2] / (m["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS)
                line["one_gpu_512_rms_matched"] = m
                if world > 1 and "rms_matched_512" in line:
                    line["rms_matched_512"]["speedup_vs_one_gpu"] = line["rms_matched_512"]["value"] / m["value"]
        if world == 1 and not a.no_cpu_baseline and a.scene == "vortex_ring" and not a.grid and a.dtype == "f32":
            line["cpu_baseline"] = cpu_baseline_reference_order(n, line["config"]["sweeps_per_step"])
            line["cpu_baseline_openmp"] = cpu_baseline(n, line["config"]["sweeps_per_step"])
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
