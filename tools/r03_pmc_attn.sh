#!/bin/bash
# PMC counters of the attention kernels (separate passes, counters only): LDS conflicts, wait reasons, instruction mix
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03pmc}; rm -rf $O; mkdir -p $O
prog=${PROG:-tools/attn_fwd_only.py}
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- python3 $prog > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in agg.items():
    if "attn" not in k: continue
    print(k)
    for c, (v, n) in sorted(d.items()): print(f"   {c:28s} {v / n:16.0f}  (avg of {n} dispatches)")
PY
find $O -name "*counter_collection.csv" -delete
