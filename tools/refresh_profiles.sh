#!/bin/bash
# Regenerates the judged artefacts under profiles/ on a GPU box (run through gpurun from the repo root; results land in gpurun_out/refresh,
# copy them into profiles/ afterwards with the round prefix):  kernel tables of the workloads (text, int8, audio, text with the reference's
# default trainable set, single-token decode), the PMC passes (separate, as MI355X_MICROARCH.md prescribes) for the GEMM traffic files and
# the per-kernel counter summary.  Steps run with && semantics: a failed GPU step ends the script.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export LLX_ROUND=${LLX_ROUND:-r03}
O=gpurun_out/refresh
rm -rf $O && mkdir -p $O
kt() {  # kt <tag> <bench args...>
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -o $tag -- python3 bench.py "$@" --no-extras --no-cpu-baseline --steps 10 > $O/bench_$tag.json 2> $O/bench_$tag.err
  python tools/kstats.py $O/kt_$tag > $O/kstats_$tag.md
  cp $O/kt_$tag/*/${tag}_kernel_stats.csv $O/kstats_$tag.csv 2>/dev/null || cp $O/kt_$tag/${tag}_kernel_stats.csv $O/kstats_$tag.csv
  echo "[refresh] kernel table $tag done"
}
kt text --config text
kt int8 --config int8
kt audio --config audio
kt text_reference_trainable --config text --trainable reference
kt decode --config decode
for cfg in text int8 audio; do
  for pmc in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $pmc --output-format csv -d $O/pmc_${cfg}_$pmc -o p -- python3 bench.py --config $cfg --steps 1 --warmup 1 --no-graph --no-extras --no-cpu-baseline > /dev/null 2> $O/pmc_${cfg}_$pmc.err
    echo "[refresh] pmc $cfg $pmc done"
  done
  python tools/pmc_traffic.py $O/pmc_${cfg}_FETCH_SIZE $O/pmc_${cfg}_WRITE_SIZE $cfg 4096 $O
done
for cfg in text int8; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${cfg}_SQ -o p -- python3 bench.py --config $cfg --steps 1 --warmup 1 --no-graph --no-extras --no-cpu-baseline > /dev/null 2> $O/pmc_${cfg}_SQ.err
  rocprofv3 --kernel-trace --output-format csv -d $O/ktng_$cfg -o p -- python3 bench.py --config $cfg --steps 1 --warmup 1 --no-graph --no-extras --no-cpu-baseline > /dev/null 2> $O/ktng_$cfg.err
  python tools/pmc_summary.py $O/pmc_${cfg}_SQ $O/pmc_${cfg}_FETCH_SIZE $O/pmc_${cfg}_WRITE_SIZE $O/ktng_$cfg > $O/pmc_summary_$cfg.md
  echo "[refresh] pmc summary $cfg done"
done
# keep the merge small: the raw traces stay on the box
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
ls -la $O
