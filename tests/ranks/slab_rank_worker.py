"""One rank of a multi-process z-slab run (started by tests/test_gpu_slab_ranks.py, one process per rank, all
on GPU 0 through the shared-memory stand-in for RCCL).  Runs the scenario named on the command line and writes
its owned planes and scalars to <out>.rank<r>.npz for the parent to assemble and compare with its single-domain
run.  The communicator id travels through a file: no torch in these processes (the torch.distributed control
plane has its own tests, and the benchmark's launch line is rehearsed separately)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np


def main():
    scenario, out_path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import fsamd_loader
    fs = fsamd_loader.load()
    from tests import slab_scenarios as sc
    def comm_id_for(i):
        """The i-th communicator id of this run: rank 0 makes it, the others wait for its file."""
        id_path = f"{out_path}.id{i}"
        if rank == 0:
            with open(id_path + ".tmp", "wb") as f:
                f.write(fs.comm_unique_id())
            os.replace(id_path + ".tmp", id_path)
        deadline = time.time() + 120
        while not os.path.exists(id_path):
            if time.time() > deadline:
                raise SystemExit("no communicator id from rank 0")
            time.sleep(0.005)
        return open(id_path, "rb").read()

    os.environ["FSAMD_TEST_RESTART_PATH"] = out_path + ".restart"
    if scenario.startswith("random:"):
        record = sc.run(fs, scenario, rank=rank, nranks=world, comm_ids=comm_id_for)
    else:
        record = sc.run(fs, scenario, rank=rank, nranks=world, comm_id=comm_id_for(0))
    np.savez(f"{out_path}.rank{rank}.npz", **{k: np.asarray(v) for k, v in record.items()})
    print(f"rank {rank} done", flush=True)


if __name__ == "__main__":
    main()
