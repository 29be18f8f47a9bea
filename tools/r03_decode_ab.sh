#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for v in ${VALS:-256 512 1024}; do
  echo "== LLX_DECODE_WGS=$v"
  LLX_DECODE_WGS=$v python bench.py --config decode --steps 30 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('ctx4096','ctx8192'):
    print(k, d['decode'][k]['ms_per_token'] if 'decode' in d else d['configs']['decode'][k]['ms_per_token'], (d.get('decode') or d['configs']['decode'])[k]['roofline']['frac'])
"
done
