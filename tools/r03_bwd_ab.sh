#!/bin/bash
# attention tests, then the backward under rocprofv3 for each value of the env knob named in $KNOB (values $VALS)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03i}
rm -rf $O && mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attn or attention" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for v in ${VALS:-8}; do
  env ${KNOB:-LLX_X}=$v rocprofv3 --kernel-trace --output-format csv -d $O/kt_$v -o t -- python3 tools/attn_bwd_bench.py > $O/bench_$v.log 2>&1
  echo "== ${KNOB} $v"; cat $O/bench_$v.log | grep attn; python tools/ktavg.py $O/kt_$v attn_
done
find $O -name "*kernel_trace.csv" -delete
