#!/bin/bash
# decode bench under launch knobs, alternating in one box:  VARS="A=1,B=2 A=3" bash tools/r03_gemv_ab.sh
cd "$GRAFT_REPO_ROOT"
run() {
  echo "== $*"
  env "$@" python bench.py --config decode --steps 30 2>gpurun_out/decode_ab.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
dd=d.get('decode') or d['configs']['decode']
for k in ('ctx4096','ctx8192'):
    print(k, dd[k]['ms_per_token'], dd[k]['roofline']['frac'])
"
}
for rep in 1 2; do
  for v in ${VARS:-LLX_DECODE_WARM_MB=0 LLX_DECODE_WARM_MB=32 LLX_DECODE_WARM_MB=96 LLX_DECODE_WARM_MB=160}; do
    run ${v//,/ }
  done
done
