"""Prints the top rows of a rocprofv3 --stats kernel_stats.csv.  usage: kernel_stats.py <dir> [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in list(csv.DictReader(open(f)))[:n]:
    name = r["Name"].split("(")[0][-56:]
    print(f"{name:56s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs']) / 1e6:8.2f} ms avg {float(r['AverageNs']) / 1e3:8.1f} us {r['Percentage']}%")
