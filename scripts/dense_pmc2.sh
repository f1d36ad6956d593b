#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/dense_pmc2
rm -rf $O; mkdir -p $O
i=0
for set in "MeanOccupancyPerCU MeanOccupancyPerActiveCU" "OccupancyPercent" "MfmaUtil" "LdsUtil" "VALUBusy" "MemUnitStalled" "SQ_LEVEL_WAVES SQ_WAVES" "SPI_RA_RES_STALL_CSN SPI_RA_WVLIM_STALL_CSN SPI_RA_TMP_STALL_CSN"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- python3 $R/scripts/dense_one.py > $O/p$i.log 2>&1 || echo "pass $i ($set) failed"
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
    print(f"{k:32s} {sum(v) / len(v):16.3f}  ({len(v)} launches)")
PY
