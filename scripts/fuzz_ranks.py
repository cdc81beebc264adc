"""Random configurations of the rank-to-rank slab path (2, 3 or 4 processes on one GPU through the shared-memory
stand-in for RCCL) against the single domain, bit for bit. Each batch of configurations runs in one set of
processes, a communicator per configuration.  usage: fuzz_ranks.py <seed> <seconds>"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fsamd_loader
fs = fsamd_loader.load()
from tests import slab_scenarios
from tests.test_gpu_slab_ranks import run_ranks

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
BATCH = 12
t0, runs, arrays, planes = time.time(), 0, 0, 0
tmp = tempfile.mkdtemp()
while time.time() - t0 < budget:
    world, seed = int(rng.integers(2, 5)), int(rng.integers(1 << 30))
    name = f"random:{seed}:{world}:{BATCH}"
    out = os.path.join(tmp, f"b{runs}")
    run_ranks([os.path.join(ROOT, "tests", "ranks", "slab_rank_worker.py"), name, out], world)
    parts = [np.load(f"{out}.rank{r}.npz") for r in range(world)]
    want = slab_scenarios.run(fs, name)
    assert sorted(parts[0].files) == sorted(want)
    for key, val in want.items():
        val = np.asarray(val)
        if val.ndim == 3:
            got = np.concatenate([p[key] for p in parts], axis=0)
            same = np.array_equal(got, val)
            arrays += 1
            planes += sum(p[key].shape[0] > 0 for p in parts)
        elif key.endswith("_sweeps"):
            same = all(np.array_equal(p[key], val) for p in parts)
        else:
            same = all(np.array_equal(p[key], parts[0][key]) for p in parts) and \
                (np.allclose(parts[0][key], val, rtol=1e-9, atol=0) or (np.isnan(val) and np.isnan(parts[0][key])))
        if not same:
            print("MISMATCH", name, key, flush=True)
            sys.exit(1)
    runs += BATCH
    print("  ...", runs, "configurations", flush=True)
print("fuzz ok:", runs, "configurations,", arrays, "fields assembled from", planes, "rank pieces", flush=True)
