"""Reads a bench.py JSON line on stdin and prints its main numbers on one line."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = d["config"]
print(" ".join(sys.argv[1:]), c["grid"], c["parallelism"], "Mcells/s", round(d["value"]), "ms/step", round(d["ms_per_step"], 2),
      "sweeps/step", c["sweeps_per_step"], "resid", c["last_residual"], "sweep_us", round(d["roofline"]["launch_us"], 1))
