"""One rank of a multi-process z-slab run (started by tests/test_gpu_slab_ranks.py, one process per rank, all
on GPU 0 through the shared-memory stand-in for RCCL).  Runs the scenario named on the command line, gathers
the global fields on rank 0 and writes them to an .npz for the parent to compare with its single-domain run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np


def main():
    scenario, out_path = sys.argv[1], sys.argv[2]
    import fsamd_loader
    fs = fsamd_loader.load()
    from fluid_simulation_amd import dist as fd
    from tests import slab_scenarios as sc
    rank, world = fd.init_control_plane("gloo")
    comm_id = fd.broadcast_bytes(fs.comm_unique_id() if rank == 0 else None)
    record = sc.run(fs, scenario, rank=rank, nranks=world, comm_id=comm_id)
    gathered = {}
    for key, val in record.items():
        if isinstance(val, np.ndarray) and val.ndim == 3:
            g = fd.gather_planes(val)
            if rank == 0:
                gathered[key] = g
        elif rank == 0:
            gathered[key] = np.asarray(val)
    if rank == 0:
        np.savez(out_path, **gathered)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} done", flush=True)


if __name__ == "__main__":
    main()
