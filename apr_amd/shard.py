"""Multi-GPU sharding of independent scan pairs.

Pairs are independent units: rank r of W owns a contiguous block of the pair
list and never exchanges data-path tensors.  The only collectives are the
barrier around the timed region and one end-of-run gather of a few floats per
rank (pairs done, seconds, max errors) -- RCCL over xGMI on the GPU box
(backend "nccl"), gloo in the CPU tests.
"""
from __future__ import annotations

import os
import sys

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block [lo, hi) of rank `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rank_seeds(rank: int, count: int, per_rank: int = 64):
    """Synthetic pair seeds of a rank: 64*rank + i (BASELINE config 4: 512 pairs over 8 GPUs)."""
    return [per_rank * rank + i for i in range(count)]


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_stats(stats, device):
    """all_gather a short list of floats per rank -> [world, len(stats)] tensor (CPU)."""
    t = torch.tensor(list(stats), dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        return t.cpu().unsqueeze(0)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu()


def aggregate_throughput(stats_matrix):
    """Whole-job pairs/s = total pairs / max seconds over ranks (columns: pairs, seconds, ...)."""
    pairs = float(stats_matrix[:, 0].sum())
    secs = float(stats_matrix[:, 1].max())
    return pairs / secs if secs > 0 else 0.0


def gather_poses(poses, n_total: int, device):
    """Poses [n_rank, 4, 4] of this rank's block of the pair list -> [n_total, 4, 4] float64 on EVERY rank, in pair-list
    order ("optional gather of poses [64,4,4] to rank 0"; BASELINE config 4).  Blocks follow shard_range,
    so their sizes differ by at most one: each rank pads its block to the largest one and a single all_gather moves
    world x ceil(n_total / world) x 16 doubles."""
    poses = torch.as_tensor(poses, dtype=torch.float64).reshape(-1, 4, 4)
    if not (dist.is_available() and dist.is_initialized()):
        if poses.shape[0] != n_total:
            raise ValueError("gather_poses: a world of one holds the whole pair list")
        return poses.cpu()
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(n_total, rank, world)
    if poses.shape[0] != hi - lo:
        raise ValueError(f"gather_poses: rank {rank} owns pairs [{lo}, {hi}) but passed {poses.shape[0]} poses")
    width = -(-n_total // world)
    mine = torch.zeros((width, 4, 4), dtype=torch.float64, device=device)
    mine[:hi - lo] = poses.to(device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    parts = []
    for r in range(world):
        a, b = shard_range(n_total, r, world)
        parts.append(out[r][:b - a].cpu())
    return torch.cat(parts)


def free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank: int, world: int, port: int, base=None):
    """Environment of rank `rank` of a one-node job of `world` ranks: the variables torch.distributed.run would set."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts (RCCL needs it)
    return env


def launch_ranks(cmd, world: int, timeout_s: float = None, grace_s: float = 20.0) -> int:
    """Start `world` child processes of `cmd` (one per GPU; rank r gets rank_env(r)), wait for all of them and return 0
    if every one exited 0, else the first non-zero exit code.  Rank 0 inherits stdout (its JSON line is the job's
    output); the other ranks' stdout is folded into stderr.  The caller must not have initialised the GPU: children
    are fresh processes (subprocess.Popen), nothing is exec'ed over a process that holds a HIP context.  When a rank
    fails, the others get `grace_s` seconds (they usually fail the same way or hang in a collective) and are then
    killed by PID.

    No rank outlives the launcher: every child leads its own session (start_new_session), SIGTERM / SIGINT / SIGHUP to
    the launcher -- a driver's `timeout`, Ctrl-C -- and any exception on the way out terminate every live child's process
    group (SIGTERM, then SIGKILL after 5 s) before the launcher returns 128 + signal.  `timeout_s` None reads
    APR_LAUNCH_TIMEOUT_S (default 3600 s): a rank wedged in a collective cannot hold the job for ever.  OMP_NUM_THREADS
    defaults to 1 as under torch.distributed.run, so host-CPU figures of the two launch forms are comparable."""
    import signal
    import subprocess
    import time
    if world < 1:
        raise ValueError("launch_ranks: world must be >= 1")
    if timeout_s is None:
        timeout_s = float(os.environ.get("APR_LAUNCH_TIMEOUT_S", "3600"))
    port = free_port()
    procs = []

    def stop_all(sig_first=signal.SIGTERM, wait_s=5.0):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig_first)          # the child leads its own group: pgid == pid (ours, by PID)
                except (ProcessLookupError, PermissionError):
                    pass
        t_end = time.monotonic() + wait_s
        for p in procs:
            try:
                p.wait(max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass
                p.wait()

    class _Signalled(Exception):
        pass

    def on_signal(signum, _frame):
        raise _Signalled(signum)

    old = {}
    for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        try:
            old[sg] = signal.signal(sg, on_signal)
        except ValueError:                               # not the main thread: the try/finally below still cleans up
            pass
    first_bad, bad_at = 0, None
    try:
        for r in range(world):
            env = rank_env(r, world, port)
            env.setdefault("OMP_NUM_THREADS", "1")
            procs.append(subprocess.Popen(list(cmd), env=env, stdout=None if r == 0 else sys.stderr,
                                          start_new_session=True))
        t0 = time.monotonic()
        live = set(range(world))
        while live:
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0 and first_bad == 0:
                    first_bad, bad_at = rc, time.monotonic()
                    print(f"[launch_ranks] rank {r} exited with code {rc}", file=sys.stderr, flush=True)
            now = time.monotonic()
            expired = now - t0 > timeout_s
            if live and (expired or (bad_at is not None and now - bad_at > grace_s)):
                if expired:
                    print(f"[launch_ranks] {len(live)} rank(s) still running after {timeout_s:.0f} s: stopping them",
                          file=sys.stderr, flush=True)
                stop_all()
                live.clear()
                if first_bad == 0:
                    first_bad = 124
            if live:
                time.sleep(0.05)
        return first_bad
    except _Signalled as e:
        print(f"[launch_ranks] signal {e.args[0]}: stopping {sum(p.poll() is None for p in procs)} rank(s)",
              file=sys.stderr, flush=True)
        return 128 + int(e.args[0])
    finally:
        stop_all()
        for sg, h in old.items():
            signal.signal(sg, h)
