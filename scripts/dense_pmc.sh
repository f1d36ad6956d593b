#!/bin/bash
# SQ counters of k_dense_gemm_bf3 on one compute-bound shape (separate passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/dense_pmc
rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- python3 $R/scripts/dense_one.py > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_dense_gemm_bf3" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    v = v[5:] if len(v) > 5 else v
    print(f"{k:32s} {sum(v) / len(v):16.0f}  ({len(v)} launches)")
PY
