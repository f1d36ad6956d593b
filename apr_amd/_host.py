"""Host-side housekeeping shared by the pipelines and bench.py."""
import os

import torch

_done = False


def cpu_quota():
    """CPUs this process may use: the cgroup v2 / v1 CFS quota if one is set, else the affinity mask."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return max(1, int(quota) // int(period))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, q // p)
    except (OSError, ValueError):
        pass
    return len(os.sched_getaffinity(0))


def limit_cpu_threads():
    """torch sizes its intra-op pool from the HOST's core count (128 threads on a 256-core box) even when a cgroup
    quota allows 16 CPUs: the idle pool threads spin after every small CPU op (a 4x4 pose product is enough), the
    quota runs out and the kernel throttles the whole process for the rest of the 100 ms period -- measured on the
    MI355X box as 20-70 ms stalls every few pairs (cpu.stat nr_throttled 82 in a 9 s run, 0 with one thread).  The hot
    path has no CPU tensor work worth a pool: cap it at a quarter of the quota (once per process, announced with a
    warning).  A process-wide side effect of building a pipeline object: APR_CPU_THREADS=0 opts out, APR_CPU_THREADS=n
    picks the cap."""
    global _done
    if _done:
        return
    _done = True
    want = os.environ.get("APR_CPU_THREADS")          # "0": leave torch's pool alone; "n": cap at n
    if want == "0":
        return
    cap = max(1, int(want)) if want else max(1, min(8, cpu_quota() // 4))
    if torch.get_num_threads() > cap:
        import warnings
        warnings.warn(f"apr_amd: capping torch's intra-op CPU pool at {cap} threads (was {torch.get_num_threads()}; cgroup "
                      f"quota {cpu_quota()} CPUs); APR_CPU_THREADS=0 leaves it alone", RuntimeWarning, stacklevel=2)
        torch.set_num_threads(cap)


def cgroup_throttle():
    """(nr_throttled, throttled_usec) of this process' cgroup so far (cpu.stat; cgroup v2, then v1), or None.  A quota that
    runs out stalls EVERY thread of the cgroup until the 100 ms period ends: a benchmark window that overlaps such a period
    -- its own start-up burst, or another process of the same container -- shows it as a gap of up to 0.1 s."""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat", "/sys/fs/cgroup/cpu,cpuacct/cpu.stat"):
        try:
            kv = dict(line.split()[:2] for line in open(path) if len(line.split()) >= 2)
        except OSError:
            continue
        if "nr_throttled" in kv:
            t = kv.get("throttled_usec")
            if t is None and "throttled_time" in kv:       # v1: nanoseconds
                t = int(kv["throttled_time"]) // 1000
            return int(kv["nr_throttled"]), int(t or 0)
    return None
