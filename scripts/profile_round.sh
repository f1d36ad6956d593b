#!/bin/bash
# Round artefacts for profiles/: the driver's exact command (3 runs, per-step log), the default bench line, rocprofv3
# kernel stats of the default and the single-stream run, the HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE in
# SEPARATE runs, kernel-trace only), and the Predator pair (kernel stats + KPConv PMC).  Run on the GPU box:
#   gpurun -- scripts/profile_round.sh r02        -> gpurun_out/profile_r02/*  (scripts/collect_profiles.py copies
#   what is judged into profiles/)
set -o pipefail
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PART=${PART:-all}      # a: the plain runs, b: the rocprof / PMC passes (gpurun limits a call to 20 minutes)
if [ "$PART" != b ]; then
for i in 1 2 3; do
  echo "[profile] driver command, run $i"
  APR_BENCH_STEPLOG=1 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-workloads > $OUT/driver_cmd_$i.json 2> $OUT/driver_cmd_$i.log </dev/null || { echo "driver cmd failed"; tail -5 $OUT/driver_cmd_$i.log; exit 1; }
done
echo "[profile] default bench"
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err </dev/null || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
echo "[profile] one stream: plain, and with the host three steps ahead + the matching batch on three lanes"
python3 $R/bench.py --streams 1 --depth 1 --match-lanes 1 --steps 120 --no-cpu-baseline --no-roofline --no-workloads > $OUT/bench_s1_plain.json 2> $OUT/bench_s1_plain.err </dev/null || { echo "s1 plain failed"; exit 1; }
python3 $R/bench.py --streams 1 --depth 3 --match-lanes 3 --steps 120 --no-cpu-baseline --no-roofline --no-workloads > $OUT/bench_s1_d3_l3.json 2> $OUT/bench_s1_d3_l3.err </dev/null || { echo "s1 depth 3 lanes 3 failed"; exit 1; }
echo "[profile] bench.py --gpus 2 with no launcher around it (gloo, both ranks on this GPU): weak line and config-4 mode"
APR_BENCH_BACKEND=gloo APR_BENCH_SINGLE_DEVICE=1 python3 $R/bench.py --gpus 2 --steps 20 --warmup 5 --no-workloads --no-cpu-baseline --no-roofline > $OUT/selflaunch_gloo2.json 2> $OUT/selflaunch_gloo2.err </dev/null || { echo "self-launch failed"; tail -5 $OUT/selflaunch_gloo2.err; exit 1; }
APR_BENCH_BACKEND=gloo APR_BENCH_SINGLE_DEVICE=1 python3 $R/bench.py --gpus 2 --pairs-total 30 --warmup 5 --no-workloads --no-cpu-baseline --no-roofline > $OUT/selflaunch_gloo2_config4.json 2> $OUT/selflaunch_gloo2_config4.err </dev/null || { echo "self-launch config 4 failed"; exit 1; }
echo "[profile] host CPU of a rank: where the three worker threads spend their time"
python3 $R/scripts/host_cpu_split.py 150 > $OUT/host_cpu_split_poll.log 2>&1 </dev/null || { echo "host split failed"; exit 1; }
APR_FETCH_WAIT=sync python3 $R/scripts/host_cpu_split.py 150 > $OUT/host_cpu_split_sync.log 2>&1 </dev/null || { echo "host split (sync) failed"; exit 1; }
echo "[profile] one pair per call: host time between the fetches against time blocked on the GPU"
python3 $R/scripts/one_pair_split.py > $OUT/one_pair_split.log 2>&1 </dev/null || { echo "one pair split failed"; exit 1; }
APR_ENCODE_PLAN=0 APR_FRONT_END_CALL=0 python3 $R/scripts/one_pair_split.py > $OUT/one_pair_split_python_plan.log 2>&1 </dev/null || { echo "one pair split (python plan) failed"; exit 1; }
fi
if [ "$PART" = a ]; then echo "[profile] part a done"; exit 0; fi
echo "[profile] matching at 0 / 30 % true matches under rocprof (the regime a trained checkpoint puts the matcher in)"
SHARES=0.0,0.3 REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/match_stats -o m -- python3 $R/scripts/match_load_bench.py > $OUT/match_load.log 2>&1 </dev/null || { echo "match stats failed"; exit 1; }
echo "[profile] rocprof stats, default run"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python3 $R/bench.py --no-cpu-baseline --no-workloads > $OUT/bench_prof.json 2> $OUT/bench_prof.err </dev/null || { echo "rocprof default failed"; exit 1; }
echo "[profile] rocprof stats, single stream"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o b -- python3 $R/bench.py --streams 1 --depth 1 --match-lanes 1 --steps 60 --no-cpu-baseline --no-workloads > $OUT/bench_s1.json 2> $OUT/bench_s1.err </dev/null || { echo "rocprof s1 failed"; exit 1; }
for C in FETCH_SIZE WRITE_SIZE; do
  echo "[profile] pmc $C"
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o p -- python3 $R/bench.py --steps 4 --warmup 2 --streams 1 --depth 1 --match-lanes 1 --no-cpu-baseline --no-roofline --no-workloads > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err </dev/null || { echo "pmc $C failed"; exit 1; }
done
echo "[profile] predator pair: kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pred_stats -o p -- python3 $R/scripts/predator_profile.py > $OUT/predator_profile.log 2>&1 </dev/null || { echo "predator stats failed"; exit 1; }
for C in FETCH_SIZE WRITE_SIZE; do
  echo "[profile] predator kpconv pmc $C"
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pred_pmc_$C -o p -- python3 $R/scripts/kpconv_bench.py > $OUT/pred_pmc_$C.log 2>&1 </dev/null || { echo "kpconv pmc $C failed"; exit 1; }
done
echo "[profile] APR training iteration: stage split + kernel stats"
ITERS=10 python3 $R/scripts/apr_train_step.py > $OUT/train_step.json 2> $OUT/train_step.log </dev/null || { echo "train step failed"; tail -5 $OUT/train_step.log; exit 1; }
ITERS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_stats -o t -- python3 $R/scripts/apr_train_step.py > $OUT/train_step_prof.json 2> $OUT/train_step_prof.log </dev/null || { echo "train step rocprof failed"; exit 1; }
python3 $R/scripts/pmc_summary.py $OUT > $OUT/pmc_spconv_summary.json && head -40 $OUT/pmc_spconv_summary.json
# keep the merge small: traces are large, the stats and counter tables are what is judged
find $OUT -name "*kernel_trace.csv" -size +20M -delete
echo "[profile] done"
