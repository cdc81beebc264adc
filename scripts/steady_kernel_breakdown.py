"""Per-kernel device time of the steady 512^3 step, from a rocprofv3 kernel trace.

Run as   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 scripts/steady_kernel_breakdown.py run N [slabs]
then     python3 scripts/steady_kernel_breakdown.py read DIR
The run makes 8 warm-up steps, sleeps, and makes 20 more; the reader takes the kernels after the longest idle gap of the
trace (the sleep) and prints, per kernel name, launches and microseconds per step, their sum and the span per step."""
import csv, glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    import fsamd_loader
    fs = fsamd_loader.load()
    n = int(sys.argv[2]); slabs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    g = fs.FluidSim3D(n, n, n, stop_mode=fs.STOP_L2_RESIDUAL, tol=1e-6, max_iters=1000, step_mode=1, check_every=-1, local_slabs=slabs)
    g.sphere_influence(n / 10.0)
    for it in range(8):
        g.fill_sphere_influence(n / 10.0, it); g.advance(1 / 30.)
    g.synchronize(); time.sleep(1.0)
    for it in range(8, 28):
        g.fill_sphere_influence(n / 10.0, it); g.advance(1 / 30.)
    g.synchronize()
    g.close()
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(f))))
    gaps = [(rows[i + 1][0] - rows[i][1], i) for i in range(len(rows) - 1)]
    cut = max(gaps)[1] + 1
    tail = rows[cut:]
    steps = 20.0
    per = {}
    for a, b, k in tail:
        c, t = per.get(k, (0, 0)); per[k] = (c + 1, t + (b - a))
    tot = 0
    for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[:80]:80s} {c / steps:7.1f} launches  {t / steps / 1e3:9.1f} us per step")
        tot += t
    print(f"sum of kernels {tot / steps / 1e3:.1f} us per step; span {(tail[-1][1] - tail[0][0]) / steps / 1e3:.1f} us per step")
