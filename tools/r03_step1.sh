#!/bin/bash
# round-3 GPU session: decode tests + profile, attention route tests, training bench with / without the dS route, kernel tables
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03d}
rm -rf $O && mkdir -p $O
python -m pytest tests/test_decode_gpu.py -x -q > $O/decode_tests.log 2>&1; echo "rc=$?" >> $O/decode_tests.log; tail -4 $O/decode_tests.log
python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > $O/attn_tests.log 2>&1; echo "rc=$?" >> $O/attn_tests.log; tail -4 $O/attn_tests.log
python bench.py --no-extras --no-cpu-baseline --steps 10 > $O/bench_text.json 2> $O/bench_text.err; tail -c 300 $O/bench_text.json
LLX_ATTN_BWD_DS=0 python bench.py --no-extras --no-cpu-baseline --steps 10 > $O/bench_text_nods.json 2> $O/bench_text_nods.err; tail -c 300 $O/bench_text_nods.json
rocprofv3 --kernel-trace --output-format csv -d $O/kt_decode -o decode -- python3 bench.py --config decode --steps 20 --warmup 5 > $O/bench_decode.json 2> $O/bench_decode.err
python tools/kstats.py $O/kt_decode --skip-last 0 --steps 20 > $O/kstats_decode.md; head -14 $O/kstats_decode.md
python bench.py --config decode --steps 20 --warmup 5 > $O/bench_decode_noprof.json 2>> $O/bench_decode.err; cut -c1-400 $O/bench_decode_noprof.json
rocprofv3 --kernel-trace --output-format csv -d $O/kt_text -o text -- python3 bench.py --no-extras --no-cpu-baseline --steps 10 > $O/bench_text_prof.json 2> $O/bench_text_prof.err
python tools/kstats.py $O/kt_text > $O/kstats_text.md; head -30 $O/kstats_text.md
find $O -name "*kernel_trace.csv" -delete
