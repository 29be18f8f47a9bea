#!/bin/bash
# kernel table of one bench config under rocprofv3 (kernel trace + stats) -> gpurun_out/$2/kstats_$1.md
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cfg=${1:-text}; O=gpurun_out/${2:-r03kt}
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$cfg -o $cfg -- python3 bench.py --config $cfg --no-extras --no-cpu-baseline --steps 10 > $O/bench_$cfg.json 2> $O/bench_$cfg.err
python tools/kstats.py $O/kt_$cfg > $O/kstats_$cfg.md
cp $O/kt_$cfg/*/${cfg}_kernel_stats.csv $O/kstats_$cfg.csv 2>/dev/null || cp $O/kt_$cfg/${cfg}_kernel_stats.csv $O/kstats_$cfg.csv
find $O -name "*kernel_trace.csv" -delete
cat $O/bench_$cfg.json | tail -1 | cut -c1-400
