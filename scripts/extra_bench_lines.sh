set -x
cd $GRAFT_REPO_ROOT
X="--no-cpu-baseline --no-512-leg --no-steady-leg --no-matched-leg"
timeout -k 10 200 python bench.py --scene smoke_plume --grid 256,512,256 $X > gpurun_out/x_plume.json 2>gpurun_out/x_plume.err
timeout -k 10 300 python bench.py --scene ring_collision $X > gpurun_out/x_collision.json 2>gpurun_out/x_collision.err
timeout -k 10 300 python bench.py --dtype f64 $X > gpurun_out/x_f64.json 2>gpurun_out/x_f64.err
timeout -k 10 300 python bench.py --n 512 --advection maccormack $X --no-multigrid-leg > gpurun_out/x_mac512.json 2>gpurun_out/x_mac512.err
timeout -k 10 500 python bench.py --n 1024 --scene ring_collision --solver multigrid $X > gpurun_out/x_1024.json 2>gpurun_out/x_1024.err
timeout -k 10 100 python scripts/time_pair_shapes.py 1024,1024,128 2>&1 | tail -1 > gpurun_out/x_slab1024.txt
python - <<'PY'
import json
for n in ["plume","collision","f64","mac512","1024"]:
    try:
        d=json.loads(open(f"gpurun_out/x_{n}.json").read())
        mg=d.get("solver_multigrid")
        print(n, round(d["ms_per_step"],3), "ms", round(d["value"]), "Mcells/s", d["config"]["sweeps_per_step"], "sweeps/step", ("| multigrid leg %.3f ms %.1f sweeps-eq" % (mg["ms_per_step"], mg["sweeps_per_step"])) if mg else "")
    except Exception as e:
        print(n, "failed", e)
print(open("gpurun_out/x_slab1024.txt").read())
PY
