#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03e}
rm -rf $O && mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or rope_epilogue" > $O/attn_tests.log 2>&1; echo "rc=$?" >> $O/attn_tests.log; tail -4 $O/attn_tests.log
python -m pytest tests/test_decode_gpu.py -x -q > $O/decode_tests.log 2>&1; echo "rc=$?" >> $O/decode_tests.log; tail -3 $O/decode_tests.log
python bench.py --no-extras --no-cpu-baseline --steps 10 > $O/bench_text.json 2> $O/bench_text.err; python -c "
import json; d=json.load(open('$O/bench_text.json')); print(d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d.get('roofline_attn'))"
LLX_ATTN_BWD_DS=0 python bench.py --no-extras --no-cpu-baseline --steps 10 > $O/bench_text_nods.json 2> $O/bench_text_nods.err; python -c "
import json; d=json.load(open('$O/bench_text_nods.json')); print(d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d.get('roofline_attn'))"
rocprofv3 --kernel-trace --output-format csv -d $O/kt_text -o text -- python3 bench.py --no-extras --no-cpu-baseline --steps 10 > $O/bench_text_prof.json 2> $O/bench_text_prof.err
python tools/kstats.py $O/kt_text > $O/kstats_text.md; grep -E "attn|wall" $O/kstats_text.md
find $O -name "*kernel_trace.csv" -delete
python bench.py --config decode --steps 20 --warmup 5 > $O/bench_decode.json 2> $O/bench_decode.err; python -c "
import json; d=json.load(open('$O/bench_decode.json'))['decode']; print({k:(v['ms_per_token'], v['roofline']['frac']) for k,v in d.items() if k.startswith('ctx')})"
