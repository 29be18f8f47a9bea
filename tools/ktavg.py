"""Average duration per kernel name from a `rocprofv3 --kernel-trace --output-format csv` directory (all dispatches)."""
import collections, csv, glob, os, sys
files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
agg = collections.defaultdict(lambda: [0, 0])
for f in files:
    for r in csv.DictReader(open(f, newline="")):
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if pat in n:
        print(f"{t / c / 1e3:9.1f} us x {c:4d}  {n[:100]}")
