#!/usr/bin/env python3
"""Benchmark of the 3D advection-reflection fluid step on MI355X.

    python bench.py --gpus 1 --steps K --warmup W            one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full fs3d_advance(): (buoyancy off) projection, half advection, projection,
reflection, half advection of the synthetic vortex-ring scene, with the projection run to the L2
residual of 3_fluid/FluidSim.cpp:17-43 <= 1e-6 or the reference's cap of 1000 sweeps
(advection-reflection/main.cpp:108); the residual is evaluated from one sweep before the count the
previous step needed and then after every sweep (the reference evaluates it after every sweep); where
that count was small and steady, the last evaluation is left to the advection kernel that follows, which
adds up the residual of the field it reads: every projection still ends on a verified residual <= tol,
and a miss is redone with more sweeps.

Workloads (BASELINE.json configs):
  N = 1   "Vortex ring 256^3 on one MI355X, GS to residual 1e-6, fp32"        (configs[1])
  N > 1   "Vortex ring 512^3 z-slab decomposed across 2/4/8 MI355X"           (configs[2]; strong scaling)

The printed JSON line carries `roofline` for the dominant kernel, the one-pass red-black sweep
(`k_rb_rows<float,NSEG>`): one launch = one full sweep = 50 algorithmic bytes per cell (the reference's
velocity-form sweep touched once per colour); the kernel applies both colours in one pass over HBM, so
its own traffic model is 25 B per cell (`blocking_factor` 2, reported beside `achieved`) and
`cpu_baseline` (the CPU oracle, one thread, timed on a bounded sample of the same workload).
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
FUSED_HBM_BYTES_PER_CELL = 25.0  # what the one-pass kernel actually has to move: both colours share one read
                                 # and one write of every face (temporal blocking factor 2)
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


def cpu_baseline(n, sweeps_per_step, sample_sweeps=None, threads=None):
    """The CPU oracle (a port of the reference's algorithm, red-black order, OpenMP over the host cores this
    process may use) on a bounded sample of the same workload: some sweeps and the two advections + reflection
    of one step, scaled to the sweep count the GPU run needed per step."""
    import numpy as np
    from oracle.oracle3d import Oracle3D
    o = Oracle3D(n, n, n)
    if threads is None:   # the box exports its CPU share as OMP_NUM_THREADS; never more than the CPUs there are
        threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1), os.cpu_count() or 1))
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
    }


def extra_leg(fs, scenes, a, nx, ny, nz, device, solver, note):
    """The same steps on a fresh single-GPU state with another solver or grid, reported beside the headline value."""
    sim = fs.FluidSim3D(nx, ny, nz, device=device, dtype=fs.F64 if a.dtype == "f64" else fs.F32,
                        stop_mode=fs.STOP_L2_RESIDUAL, tol=a.tol, max_iters=a.max_iters, check_every=a.check_every,
                        omega=a.omega, step_mode=a.step_mode, solver=solver, buoyancy=1.0 if a.scene == "smoke_plume" else 0.0)
    scene_init, scene_step = scenes.SCENES[a.scene]
    scene_init(sim)
    it = 0
    for _ in range(a.warmup):
        scene_step(sim, it); it += 1
        sim.advance(DT)
    sim.synchronize()
    t0 = time.perf_counter()
    sweeps = 0
    for _ in range(a.steps):
        scene_step(sim, it); it += 1
        sim.advance(DT)
        sweeps += sim.stats().sweeps_step
    sim.synchronize()
    elapsed = time.perf_counter() - t0
    st = sim.stats()
    sim.close()
    return {"grid": [nx, ny, nz], "value": nx * ny * nz / (elapsed / a.steps) / 1e6, "unit": "Mcells/s",
            "ms_per_step": elapsed / a.steps * 1e3, "sweeps_per_step": sweeps / a.steps, "last_residual": st.metric, "note": note}


def multigrid_leg(fs, scenes, a, nx, ny, nz, device):
    return extra_leg(fs, scenes, a, nx, ny, nz, device, 1,
                     "FS_SOLVER_MULTIGRID: V(2,2) cycles around the same red-black sweeps, 2x2x2 coarsening, "
                     "piecewise-constant transfer, same stopping rule, a cycle counted as 4 sweeps")


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
        a.gpus = world

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
    comm_id = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)   # control plane only
        box = [fs.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm_id = box[0]

    import fluid_simulation_amd.scenes as scenes
    n = a.n or (256 if world == 1 else 512)
    nx, ny, nz = (int(v) for v in a.grid.split(",")) if a.grid else (n, n, n)
    sim = fs.FluidSim3D(nx, ny, nz, rank=rank, nranks=world, comm_id=comm_id, device=device,
                        dtype=fs.F64 if a.dtype == "f64" else fs.F32,
                        stop_mode=fs.STOP_L2_RESIDUAL, tol=a.tol, max_iters=a.max_iters,
                        check_every=a.check_every, omega=a.omega, step_mode=a.step_mode, local_slabs=a.local_slabs, solver=1 if a.solver == "multigrid" else 0,
                        advection=1 if a.advection == "maccormack" else 0,
                        buoyancy=1.0 if a.scene == "smoke_plume" else 0.0)
    scene_init, scene_step = scenes.SCENES[a.scene]
    scene_init(sim)

    def barrier():
        torch.cuda.synchronize()
        sim.synchronize()
        if world > 1:
            dist.barrier()

    it = 0
    for _ in range(a.warmup):
        scene_step(sim, it); it += 1
        sim.advance(DT)
    barrier()
    t0 = time.perf_counter()
    sweeps = evals = 0
    for _ in range(a.steps):
        scene_step(sim, it); it += 1
        sim.advance(DT)
        st_ = sim.stats()
        sweeps += st_.sweeps_step
        evals += st_.evaluations_step
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    st = sim.stats()

    # dominant kernel, timed live with HIP events on the library's stream
    nsw = 20
    sim.time_sweeps(4)
    ms = sim.time_sweeps(nsw)
    cells_local = sim.shape[0] * ny * nx
    launch_s = ms * 1e-3 / nsw           # one launch per full sweep
    esz_scale = 2.0 if a.dtype == "f64" else 1.0   # 8-byte faces: 2 x (24 + 24) + 2 flag bytes ~ 98 B
    sweep_bytes = (48.0 * esz_scale + 2.0)
    achieved = sweep_bytes * cells_local / launch_s / 1e9
    barrier()

    if rank == 0:
        cells = nx * ny * nz
        ms_step = elapsed / a.steps * 1e3
        sweeps_per_step = sweeps / a.steps
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
                "residual_norm": "sqrt(sum (div/dt)^2) / #interior cells, the norm of 3_fluid/FluidSim.cpp:25-43",
                "device_ms_until_last_projection": st.ms_project, "device_ms_last_advection": st.ms_advect,
            },
            "roofline": {
                "bound": "hbm", "kernel": f"k_rb_rows<{'double' if a.dtype == 'f64' else 'float'}> (one-pass red-black sweep)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "bytes_per_launch": sweep_bytes * cells_local, "launch_us": launch_s * 1e6,
                "blocking_factor": 2,
                "hbm_model_bytes_per_launch": 0.5 * sweep_bytes * cells_local,
                "hbm_model_rate": 0.5 * sweep_bytes * cells_local / launch_s / 1e9,
                # the whole step against the same peak: advection 25 + reflection 36 + advection 45 + 50 per sweep
                # algorithmic bytes per cell (fp32), over the measured step time
                "full_step_bytes_per_cell": 25 + 36 + 45 + 50 * sweeps_per_step,
                "full_step_frac": (25 + 36 + 45 + 50 * sweeps_per_step) * esz_scale * cells / (elapsed / a.steps) / 1e9
                                  / HBM_PEAK_GBS / world,
            },
        }
        pmc = os.path.join(ROOT, "profiles", "sweep_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc)).get(str(n))
                if rec and world == 1 and a.dtype == "f32" and not a.grid:
                    line["roofline"]["traffic"] = rec["hbm_bytes_per_launch"]
            except Exception:
                pass
        if world == 1 and a.solver == "sweeps" and a.local_slabs == 1 and not a.no_multigrid_leg:
            line["solver_multigrid"] = multigrid_leg(fs, scenes, a, nx, ny, nz, device)
        big = (512, 512, 512)
        if not a.no_512_leg and a.scene == "vortex_ring" and not a.n and not a.grid and a.dtype == "f32" and a.solver == "sweeps" \
                and a.advection == "semi_lagrange":
            # the configuration the multi-GPU lines are quoted on, on this rank's GPU alone (the other ranks wait):
            # with N = 1 it sits beside the 256^3 headline, with N > 1 it is the base of the strong-scaling ratio
            leg = extra_leg(fs, scenes, a, *big, device, 0, "vortex ring 512^3 on one GPU, same steps and stopping rule")
            leg["hbm_roofline_frac_full_step"] = ((25 + 36 + 45 + 50 * leg["sweeps_per_step"]) * big[0] * big[1] * big[2]
                                                  / (leg["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS)
            line["one_gpu_512"] = leg
            if world > 1 and (nx, ny, nz) == big:
                line["speedup_vs_one_gpu"] = line["value"] / leg["value"]
        if world == 1 and not a.no_cpu_baseline and a.scene == "vortex_ring" and not a.grid and a.dtype == "f32":
            line["cpu_baseline"] = cpu_baseline(n, sweeps_per_step)
        print(json.dumps(line), flush=True)
    sim.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
