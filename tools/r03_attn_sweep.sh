#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03g}
rm -rf $O && mkdir -p $O
for pr in 0 1 2 3; do
  LLX_DQ2_PROBE=$pr rocprofv3 --kernel-trace --output-format csv -d $O/kt_$pr -o t -- python3 tools/attn_bwd_bench.py > $O/bench_$pr.log 2>&1
  echo "== probe $pr"; python tools/ktavg.py $O/kt_$pr attn_bwd_dq2
done
find $O -name "*kernel_trace.csv" -delete
