#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03h}
rm -rf $O && mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attn or attention" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 100 python tools/attn_stamps.py
for nw in ${NWS:-8}; do
  LLX_ATTN_FWD_NW=$nw rocprofv3 --kernel-trace --output-format csv -d $O/kt_$nw -o t -- python3 tools/attn_fwd_only.py all > $O/bench_$nw.log 2>&1
  echo "== NW $nw"
  python - <<PY
import csv, glob
for f in glob.glob("$O/kt_$nw/**/*kernel_trace.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if "attn_fwd" in r["Kernel_Name"]]
    for i in range(0,len(rows),10):
        d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows[i:i+10]]
        print("  group", i//10, "min %.1f med %.1f us" % (min(d), sorted(d)[len(d)//2]))
PY
done
find $O -name "*kernel_trace.csv" -delete
