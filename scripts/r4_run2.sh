#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_hash_gpu.py tests/test_spconv_gpu.py tests/test_match_pose_gpu.py tests/test_fullsize_gpu.py tests/test_resunet_gpu.py -m gpu -x -q > $O/r4_gputest2.log 2>&1 || { tail -40 $O/r4_gputest2.log; exit 1; }
tail -2 $O/r4_gputest2.log
timeout -k 10 300 python scripts/host_cpu_split.py 150 > $O/r4_hostcpu_block.log 2>&1 || { tail -20 $O/r4_hostcpu_block.log; exit 2; }
APR_BLOCKING_EVENTS=0 timeout -k 10 300 python scripts/host_cpu_split.py 150 > $O/r4_hostcpu_spin.log 2>&1 || exit 3
tail -8 $O/r4_hostcpu_block.log; tail -8 $O/r4_hostcpu_spin.log
for rep in 1 2; do
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-workloads --no-cpu-baseline > $O/r4_drv2_$rep.json 2> $O/r4_drv2_$rep.log || exit 4
done
timeout -k 10 300 python bench.py --steps 200 --no-workloads --no-cpu-baseline > $O/r4_bench2.json 2> $O/r4_bench2.log || exit 5
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_s1 -o s1 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --streams 1 --depth 1 --match-lanes 1 --steps 40 --warmup 5 --no-workloads --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$O/r4_prof_s1.json 2> $GRAFT_REPO_ROOT/$O/r4_prof_s1.log || exit 6
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_drv2_*.json'))+['gpurun_out/r4_bench2.json','gpurun_out/r4_prof_s1.json']:
    d=json.loads(open(f).read().strip().splitlines()[-1]); c=d['config']
    print(f, round(d['value'],1), 'cpu/step', round(c['host_cpu_s_per_step']*1e3,2),'ms busy',round(c['host_cpus_busy'],2))
PY
find gpurun_out/prof_s1 -name '*kernel_stats.csv' | head
