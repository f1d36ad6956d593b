#!/bin/bash
# round-4 first GPU pass: parity suite, the driver's command with blocking / spinning fetch events, the self-launched
# 2-rank rehearsal (gloo, one device), then the full default bench line
set -o pipefail
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r4_gputest1.log 2>&1 || { tail -30 $O/r4_gputest1.log; exit 1; }
tail -2 $O/r4_gputest1.log
for rep in 1 2; do
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-workloads --no-cpu-baseline > $O/r4_drv_block_$rep.json 2> $O/r4_drv_block_$rep.log || exit 2
APR_BLOCKING_EVENTS=0 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-workloads --no-cpu-baseline > $O/r4_drv_spin_$rep.json 2> $O/r4_drv_spin_$rep.log || exit 3
done
APR_BENCH_BACKEND=gloo APR_BENCH_SINGLE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-workloads --no-cpu-baseline --no-roofline > $O/r4_gloo2.json 2> $O/r4_gloo2.log || exit 4
APR_BENCH_BACKEND=gloo APR_BENCH_SINGLE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --pairs-total 30 --warmup 5 --no-workloads --no-cpu-baseline --no-roofline > $O/r4_gloo2_cfg4.json 2> $O/r4_gloo2_cfg4.log || exit 5
timeout -k 10 600 python bench.py > $O/r4_bench_full.json 2> $O/r4_bench_full.log || exit 6
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_drv_*.json'))+['gpurun_out/r4_gloo2.json','gpurun_out/r4_gloo2_cfg4.json','gpurun_out/r4_bench_full.json']:
    d=json.loads(open(f).read().strip().splitlines()[-1]); c=d['config']
    print(f, round(d['value'],1), d['n_gpus'], 'cpu/step', round(c['host_cpu_s_per_step']*1e3,2),'ms busy',round(c['host_cpus_busy'],2), c.get('per_rank_pairs_per_s'))
PY
