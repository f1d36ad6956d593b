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
    killed by PID."""
    import subprocess
    import time
    if world < 1:
        raise ValueError("launch_ranks: world must be >= 1")
    port = free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen(list(cmd), env=rank_env(r, world, port),
                                      stdout=None if r == 0 else sys.stderr))
    t0 = time.monotonic()
    first_bad, bad_at = 0, None
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
        expired = timeout_s is not None and now - t0 > timeout_s
        if live and (expired or (bad_at is not None and now - bad_at > grace_s)):
            for r in sorted(live):
                procs[r].kill()
                procs[r].wait()
            live.clear()
            if first_bad == 0:
                first_bad = 124
        if live:
            time.sleep(0.05)
    return first_bad
