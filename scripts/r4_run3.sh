#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_match_pose_gpu.py tests/test_hash_gpu.py tests/test_predator_pose_mining_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > $O/r4_gputest3.log 2>&1 || { tail -40 $O/r4_gputest3.log; exit 1; }
tail -2 $O/r4_gputest3.log
timeout -k 10 300 python scripts/host_cpu_split.py 150 > $O/r4_hostcpu_poll.log 2>&1 || { tail -20 $O/r4_hostcpu_poll.log; exit 2; }
tail -8 $O/r4_hostcpu_poll.log
APR_FETCH_POLL_US=20 timeout -k 10 300 python scripts/host_cpu_split.py 150 > $O/r4_hostcpu_poll20.log 2>&1 || exit 2
tail -8 $O/r4_hostcpu_poll20.log
for v in "poll:1" "sync:1" "poll:0" ; do
W=${v%%:*}; S=${v##*:}
for rep in 1 2; do
APR_FETCH_WAIT=$W APR_RANSAC_SCREEN=$S timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-workloads --no-cpu-baseline --no-roofline > $O/r4_drv3_${W}_s${S}_$rep.json 2> $O/r4_drv3_${W}_s${S}_$rep.log || exit 4
done
APR_FETCH_WAIT=$W APR_RANSAC_SCREEN=$S timeout -k 10 300 python bench.py --steps 200 --no-workloads --no-cpu-baseline --no-roofline > $O/r4_bench3_${W}_s${S}.json 2> $O/r4_bench3_${W}_s${S}.log || exit 5
done
timeout -k 10 300 python scripts/match_load_bench.py > $O/r4_matchload3.log 2>&1 || exit 6
tail -5 $O/r4_matchload3.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_drv3_*.json'))+sorted(glob.glob('gpurun_out/r4_bench3_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); c=d['config']
    print(f, round(d['value'],1), 'cpu/step', round(c['host_cpu_s_per_step']*1e3,2),'ms busy',round(c['host_cpus_busy'],2))
PY
