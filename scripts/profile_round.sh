#!/bin/bash
# Round artefacts for profiles/: default bench line, rocprofv3 kernel stats of the default and the single-stream run,
# and the HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE in SEPARATE runs, kernel-trace only).  Run on the GPU box:
#   gpurun -- scripts/profile_round.sh r01        -> gpurun_out/profile_r01/*  (copy what is judged into profiles/)
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "[profile] default bench"; 
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err </dev/null || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
echo "[profile] rocprof stats, default run"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/bench_prof.err </dev/null || { echo "rocprof default failed"; exit 1; }
echo "[profile] rocprof stats, single stream"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o b -- python3 $R/bench.py --streams 1 --steps 60 --no-cpu-baseline > $OUT/bench_s1.json 2> $OUT/bench_s1.err </dev/null || { echo "rocprof s1 failed"; exit 1; }
for C in FETCH_SIZE WRITE_SIZE; do
  echo "[profile] pmc $C"
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o p -- python3 $R/bench.py --steps 4 --warmup 2 --streams 1 --no-cpu-baseline --no-roofline > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err </dev/null || { echo "pmc $C failed"; exit 1; }
done
python3 $R/scripts/pmc_summary.py $OUT > $OUT/pmc_spconv_summary.json && cat $OUT/pmc_spconv_summary.json | head -30
# keep the merge small: traces are large, the stats and counter tables are what is judged
find $OUT -name "*kernel_trace.csv" -size +20M -delete
echo "[profile] done"
