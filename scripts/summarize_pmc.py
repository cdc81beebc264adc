"""Summarises rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, one pass each) per kernel.

usage: summarize_pmc.py <dir with FETCH_SIZE_<n>/ and WRITE_SIZE_<n>/> <n> [<n> ...]
Prints a table and writes profiles/sweep_traffic.json for bench.py.  FETCH_SIZE is doubled before it
is compared with a byte count (on gfx950 it tallies 128-byte requests at 64 bytes); the calibration
rows show what that gives on kernels whose traffic is known exactly.
"""
import collections, csv, glob, json, os, sys

import datetime

root = sys.argv[1]
collected = os.environ.get("FSAMD_COLLECTED") or datetime.date.today().isoformat()
out = {}
for n in map(int, sys.argv[2:]):
    per = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(root, f"{c}_{n}", "**", "*_counter_collection.csv"), recursive=True)
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            vals[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]) * 1024)
        for k, v in vals.items():
            per[k][c] = (sum(v) / len(v), len(v))
    cells = n ** 3
    print(f"\n{n}^3  ({cells} cells)  bytes per launch / per cell")
    print(f"{'kernel':45s} {'launches':>8s} {'FETCH raw':>12s} {'FETCH x2':>12s} {'WRITE':>12s} {'HBM B/cell':>10s}")
    for k, d in sorted(per.items()):
        f, nf = d.get("FETCH_SIZE", (0, 0)); w, _ = d.get("WRITE_SIZE", (0, 0))
        print(f"{k[:45]:45s} {nf:8d} {f:12.4g} {2 * f:12.4g} {w:12.4g} {(2 * f + w) / cells:10.2f}")
        if "k_rb_rows2" in k:                        # one launch = two full red-black sweeps
            out[f"{n}_pairs"] = {"kernel": k, "fetch_size_raw_bytes": f, "write_size_bytes": w, "hbm_bytes_per_launch": 2 * f + w,
                                 "hbm_bytes_per_cell": (2 * f + w) / cells, "algorithmic_bytes_per_cell": 100.0,
                                 "one_pass_model_bytes_per_cell": 25.0, "collected": collected}
        elif "k_rb_rows" in k or "k_rb_fused" in k:   # one launch = one full red-black sweep
            out[str(n)] = {"kernel": k, "fetch_size_raw_bytes": f, "write_size_bytes": w, "hbm_bytes_per_launch": 2 * f + w,
                           "hbm_bytes_per_cell": (2 * f + w) / cells, "algorithmic_bytes_per_cell": 50.0,
                           "one_pass_model_bytes_per_cell": 25.0, "collected": collected}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "sweep_traffic.json"), "w"), indent=1)
