"""Per-kernel PMC summary of one bench.py step: MFMA-pipe busy fraction, and HBM-side bytes per launch / GB/s against the 8 TB/s peak.

    python tools/pmc_summary.py <sq_dir> <fetch_dir> <write_dir> <kernel_trace_dir> > profiles/rNN_pmc_summary.md

sq_dir: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE ; fetch_dir / write_dir: --pmc FETCH_SIZE / WRITE_SIZE (separate
passes, MI355X_MICROARCH.md); kernel_trace_dir: --kernel-trace of the same command (un-profiled durations).  gfx950 corrections: read bytes
= 2 x FETCH_SIZE KiB; GRBM_GUI_ACTIVE is the sum over the 8 XCDs; 1024 SIMDs."""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[:90]


def counters(d):
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"])
                out[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[(k, r["Counter_Name"])] += 1
    return out, cnt


def durations(d):
    tot, n = collections.Counter(), collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"])
                tot[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                n[k] += 1
    return {k: tot[k] / n[k] / 1e3 for k in tot}  # avg us


def main():
    sq, sqn = counters(sys.argv[1])
    ft, ftn = counters(sys.argv[2])
    wt, wtn = counters(sys.argv[3])
    dur = durations(sys.argv[4])
    rows = []
    for k in sorted(set(sq) | set(ft)):
        if k.startswith("void at::") or k.startswith("__amd") or "at::native" in k:
            continue
        n = max(sqn.get((k, "GRBM_GUI_ACTIVE"), 0), ftn.get((k, "FETCH_SIZE"), 0))
        if n == 0:
            continue
        mf = sq[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        gui = sq[k].get("GRBM_GUI_ACTIVE", 0.0)
        busy = 100.0 * mf / (gui / 8.0 * 1024.0) if gui else 0.0
        nf, nw = ftn.get((k, "FETCH_SIZE"), 0), wtn.get((k, "WRITE_SIZE"), 0)
        rd = 2.0 * ft[k].get("FETCH_SIZE", 0.0) * 1024.0 / nf if nf else 0.0
        wr = wt[k].get("WRITE_SIZE", 0.0) * 1024.0 / nw if nw else 0.0
        us = dur.get(k, 0.0)
        gbs = (rd + wr) / us / 1e3 if us else 0.0
        rows.append((k, n, us, busy, rd / 1e6, wr / 1e6, gbs))
    print("| kernel | launches | avg us (kernel trace) | MFMA pipe busy % | read MB / launch | written MB / launch | GB/s | % of 8 TB/s |")
    print("|---|---|---|---|---|---|---|---|")
    for k, n, us, busy, rd, wr, gbs in sorted(rows, key=lambda r: -r[2] * r[1]):
        print(f"| `{k}` | {n} | {us:.1f} | {busy:.1f} | {rd:.1f} | {wr:.1f} | {gbs:.0f} | {gbs / 80:.1f} |")


if __name__ == "__main__":
    main()
