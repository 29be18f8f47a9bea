"""Summarise a `rocprofv3 --kernel-trace` CSV of `bench.py` into a per-kernel table over the steady-state steps.

    python tools/kstats.py <dir with *_kernel_trace.csv> [--skip-last 1] [--steps 10] > profiles/rNN_....md

Steps are delimited by `embedding_fwd_kernel` launches (one per step); the last `--skip-last` steps (the event-traced eager step
of bench.py) are dropped and the `--steps` steps before them are averaged."""
import argparse
import collections
import csv
import glob
import os
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--skip-last", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--marker", default="embedding_fwd_kernel")
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv under " + a.dir)
    rows = []
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    if len(marks) < a.steps + a.skip_last + 1:
        sys.exit(f"only {len(marks)} step markers found")
    hi = marks[len(marks) - a.skip_last] if a.skip_last else len(rows)
    lo = marks[len(marks) - a.skip_last - a.steps]
    sel = rows[lo:hi]
    wall = (rows[hi][0] if hi < len(rows) else sel[-1][1]) - sel[0][0]
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in sel:
        agg[n][0] += 1
        agg[n][1] += e - s
    tot = sum(v[1] for v in agg.values())
    print(f"steady-state window: {a.steps} steps, wall {wall / a.steps / 1e6:.2f} ms/step, sum of kernel time {tot / a.steps / 1e6:.2f} ms/step, "
          f"{sum(v[0] for v in agg.values()) / a.steps:.0f} launches/step\n")
    print("| kernel | calls/step | ms/step | avg us | % |")
    print("|---|---|---|---|---|")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{short(n)}` | {c / a.steps:.1f} | {t / a.steps / 1e6:.3f} | {t / c / 1e3:.1f} | {100 * t / tot:.2f} |")


if __name__ == "__main__":
    main()
