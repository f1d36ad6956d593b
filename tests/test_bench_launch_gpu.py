"""bench.py --gpus N with no launcher around it starts its own N ranks (BASELINE config 4's launch shape).
Rehearsed on the one GPU of the test box: gloo backend, both ranks on cuda:0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(APR_BENCH_BACKEND="gloo", APR_BENCH_SINGLE_DEVICE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--warmup", "1", "--pool", "3",
                        "--no-workloads", "--no-cpu-baseline", "--no-roofline"] + extra,
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                      # ONE JSON line, from rank 0
    return json.loads(lines[0]), p.stderr


def test_bench_gpus_2_starts_two_ranks_and_reports_both(dev):
    out, err = _run(["--steps", "4"])
    assert "launching 2 ranks" in err
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "weak"
    cfg = out["config"]
    assert len(cfg["per_rank_pairs_per_s"]) == 2 and all(v > 0 for v in cfg["per_rank_pairs_per_s"])
    assert len(cfg["per_rank_host_cpus_busy"]) == 2 and cfg["host_cpu_s_per_step"] > 0
    # whole-job value = pairs of BOTH ranks over the max-over-ranks time
    assert abs(out["value"] - 2 * 4 * cfg["pairs_per_step"] / (out["ms_per_step"] * 4e-3)) < 1e-6 * out["value"]


def test_bench_config4_mode_shards_the_pair_list(dev):
    out, _ = _run(["--pairs-total", "14"])
    cfg = out["config"]
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and cfg["pairs_total"] == 14
    assert cfg["pairs_this_rank"] == 7 and cfg["poses_gathered"] == [14, 4, 4]


def test_bench_under_torch_distributed_run_the_drivers_launch_form(dev):
    """The driver's N > 1 command line: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ... -- here with 2 ranks over gloo on the one GPU of the box: ONE JSON line from rank 0
    with both ranks in it, and no second launch by bench.py itself (WORLD_SIZE is set)."""
    from apr_amd.shard import free_port
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(APR_BENCH_BACKEND="gloo", APR_BENCH_SINGLE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--pool", "3", "--no-workloads", "--no-cpu-baseline", "--no-roofline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and len(out["config"]["per_rank_pairs_per_s"]) == 2
    assert "launching 2 ranks" not in p.stderr
