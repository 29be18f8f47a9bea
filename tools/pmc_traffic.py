"""HBM bytes per launch of the GEMM kernels from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately, as
MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2 - they do not fit one pass).

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <config> <seq> <out_dir>

Writes profiles-style JSON files  r<NN>_<config>_s<seq>_<kind>_gemm_hbm_traffic.json (round prefix from $LLX_ROUND, default r03)  for kind in {bf16, i8} that ran, with the gfx950
correction (FETCH_SIZE reports half the bytes of wide coalesced reads: x2; WRITE_SIZE exact; both in KiB)."""
import collections
import csv
import glob
import json
import os
import re
import sys

KIND = [("i8", re.compile(r"gemm_nt_kernel<\d+, true")), ("bf16", re.compile(r"gemm_nt_kernel<\d+, false"))]


def collect(d, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                for kind, rx in KIND:
                    if rx.search(r["Kernel_Name"]):
                        tot[kind] += float(r["Counter_Value"])
                        cnt[kind] += 1
                        break
    return tot, cnt


def main():
    fetch_dir, write_dir, config, seq, out_dir = sys.argv[1:6]
    ft, fc = collect(fetch_dir, "FETCH_SIZE")
    wt, wc = collect(write_dir, "WRITE_SIZE")
    for kind in ft:
        n = fc[kind]
        if n == 0 or wc[kind] != n:
            print(f"{kind}: launch counts differ between the passes ({n} vs {wc[kind]}): skipped", file=sys.stderr)
            continue
        per = (2.0 * ft[kind] + wt[kind]) * 1024.0 / n
        out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over python3 bench.py --config {config} --seq {seq} "
                         "--steps 1 --warmup 1 --no-graph --no-extras --no-cpu-baseline; " + ("i8" if kind == "i8" else "bf16") + " gemm_nt_kernel launches only",
               "gemm_launches": n, "FETCH_SIZE_KB_sum": ft[kind], "WRITE_SIZE_KB_sum": wt[kind],
               "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request: read bytes = 2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
               "hbm_bytes_per_launch": per}
        path = os.path.join(out_dir, f"{os.environ.get('LLX_ROUND', 'r03')}_{config}_s{seq}_{kind}_gemm_hbm_traffic.json")
        with open(path, "w") as fh:
            json.dump(out, fh, indent=1)
        print(path, f"{per / 1e6:.1f} MB per launch over {n} launches")


if __name__ == "__main__":
    main()
