#!/bin/bash
# kernel table of one bench config under rocprofv3 (kernel trace + stats) -> gpurun_out/$2/kstats_$3.md ; extra bench args in $XARGS
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cfg=${1:-text}; O=gpurun_out/${2:-r03kt}; tag=${3:-$cfg}
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -o $tag -- python3 bench.py --config $cfg --no-extras --no-cpu-baseline --steps 10 $XARGS > $O/bench_$tag.json 2> $O/bench_$tag.err
python tools/kstats.py $O/kt_$tag > $O/kstats_$tag.md
cp $O/kt_$tag/*/${tag}_kernel_stats.csv $O/kstats_$tag.csv 2>/dev/null || cp $O/kt_$tag/${tag}_kernel_stats.csv $O/kstats_$tag.csv
find $O -name "*kernel_trace.csv" -delete
cat $O/bench_$tag.json | tail -1 | cut -c1-300
