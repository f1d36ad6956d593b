"""N > 1 path on CPU: two gloo ranks shard the pair list, gather stats and agree on the aggregate."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from apr_amd import shard


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(11, rank, world)
    seeds = shard.rank_seeds(rank, hi - lo)
    secs = 1.0 + rank                       # rank 1 is slower
    m = shard.gather_stats([hi - lo, secs, 1e-6 * (rank + 1)], torch.device("cpu"))
    t = shard.max_over_ranks(secs, torch.device("cpu"))
    # config-4 pose gather: every pose carries its GLOBAL pair index, blocks of unequal size (6 and 5)
    mine = torch.eye(4, dtype=torch.float64).repeat(hi - lo, 1, 1)
    mine[:, 0, 3] = torch.arange(lo, hi, dtype=torch.float64)
    allp = shard.gather_poses(mine, 11, torch.device("cpu"))
    assert allp.shape == (11, 4, 4) and allp[:, 0, 3].tolist() == [float(i) for i in range(11)]
    assert torch.equal(allp[:, :3, :3], torch.eye(3, dtype=torch.float64).expand(11, 3, 3))
    dist.barrier()
    q.put((rank, lo, hi, seeds, m.tolist(), t, shard.aggregate_throughput(m)))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(30) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    (r0, lo0, hi0, s0, m0, t0, a0), (r1, lo1, hi1, s1, m1, t1, a1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 6, 6, 11)                 # contiguous, covers all 11 pairs once
    assert s0 == [0, 1, 2, 3, 4, 5] and s1 == [64, 65, 66, 67, 68]
    assert m0 == m1 and m0[0][:2] == [6.0, 1.0] and m0[1][:2] == [5.0, 2.0]
    assert t0 == t1 == 2.0                                        # max over ranks
    assert abs(a0 - 11 / 2.0) < 1e-12 and a0 == a1


def test_shard_range_edge_cases():
    assert shard.shard_range(0, 0, 4) == (0, 0)
    got = [shard.shard_range(10, r, 4) for r in range(4)]
    assert got == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard.shard_range(512, 7, 8) == (448, 512)
    assert shard.gather_stats([1, 2], torch.device("cpu")).shape == (1, 2)   # no process group: world of one
    assert shard.gather_poses(torch.eye(4).repeat(3, 1, 1), 3, torch.device("cpu")).shape == (3, 4, 4)


# ---- bench.py --gpus N starts its own ranks (apr_amd.shard.launch_ranks) ----

def test_launch_ranks_env_plumbing(tmp_path, capfd):
    import json
    import sys
    code = ("import os, json, sys; r = os.environ['RANK'];"
            "json.dump({k: os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT',"
            " 'HSA_ENABLE_IPC_MODE_LEGACY')}, open(os.path.join(sys.argv[1], r + '.json'), 'w'));"
            "print('line from rank ' + r)")
    assert shard.launch_ranks([sys.executable, "-c", code, str(tmp_path)], 3, timeout_s=60) == 0
    envs = [json.load(open(tmp_path / f"{r}.json")) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert {e["WORLD_SIZE"] for e in envs} == {"3"} and {e["MASTER_ADDR"] for e in envs} == {"127.0.0.1"}
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and envs[0]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out, err = capfd.readouterr()
    assert out.strip() == "line from rank 0"                      # only rank 0 owns stdout (the JSON line)
    assert "line from rank 1" in err and "line from rank 2" in err


def test_launch_ranks_failure_propagates_and_stragglers_are_killed():
    import sys
    import time
    code = "import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(3)\ntime.sleep(120)"
    t0 = time.monotonic()
    assert shard.launch_ranks([sys.executable, "-c", code], 2, grace_s=0.5) == 3
    assert time.monotonic() - t0 < 30


def test_bench_gpus_flag_plumbing():
    """bench.py --gpus 2 with no launcher around it starts two ranks itself (here they stop at "needs a GPU", non-zero);
    a launcher whose WORLD_SIZE disagrees with --gpus is refused before anything else happens."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if not torch.cuda.is_available():
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode != 0 and "launching 2 ranks" in p.stderr and p.stderr.count("needs a GPU") == 2, p.stderr
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in p.stderr, p.stderr


def test_launcher_leaves_no_rank_behind_when_it_is_terminated(tmp_path):
    """SIGTERM to the launcher (a driver's `timeout`) must take every rank with it, and a rank that never exits
    must not hold the launcher past its deadline."""
    import signal
    import subprocess
    import sys
    import time
    pidfile = tmp_path / "pids"
    child = ("import os, time\n"
             f"open({str(pidfile)!r}, 'a').write(str(os.getpid()) + '\\n')\n"
             "time.sleep(600)\n")
    launcher = ("import sys\n"
                f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
                "from apr_amd import shard\n"
                f"sys.exit(shard.launch_ranks([sys.executable, '-c', {child!r}], 3, timeout_s=float(sys.argv[1])))\n")

    def alive(pid):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        try:                                       # a zombie still answers kill(0): look at its state
            return open(f"/proc/{pid}/stat").read().split(") ")[1][0] != "Z"
        except FileNotFoundError:
            return False

    def wait_pids():
        for _ in range(200):
            if pidfile.exists() and len(pidfile.read_text().split()) == 3:
                return [int(v) for v in pidfile.read_text().split()]
            time.sleep(0.05)
        raise AssertionError("ranks did not start")

    p = subprocess.Popen([sys.executable, "-c", launcher, "600"])
    pids = wait_pids()
    p.send_signal(signal.SIGTERM)
    assert p.wait(30) == 128 + signal.SIGTERM
    assert not any(alive(q) for q in pids)
    pidfile.unlink()
    p = subprocess.Popen([sys.executable, "-c", launcher, "1.0"])      # deadline: ranks sleep for ever
    pids = wait_pids()
    assert p.wait(30) == 124
    assert not any(alive(q) for q in pids)
