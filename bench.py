#!/usr/bin/env python3
"""bench.py -- pairs/sec of the FCGF_APR hot path on MI355X (BASELINE.json metric).

A "step" = one synthetic KITTI-shaped scan pair (2 x ~118 k points) through
voxel hash -> sparse ResUNet encode (both frames) -> feature NN -> RANSAC(4 M
iterations, the reference's criteria) + Kabsch, with the raw xyz of the pair
already resident in HBM when the timed region starts (BASELINE config[1]).

    python bench.py --gpus N --steps K --warmup W        (N > 1: starts its N ranks itself)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, independent pairs per rank (no data-path
collective; RCCL only for the barrier and the max-over-ranks time), weak scaling.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=18)
    ap.add_argument("--model", default="ResUNetBN2C")
    ap.add_argument("--n-out", type=int, default=32)
    ap.add_argument("--ransac-iters", type=int, default=4000000)
    ap.add_argument("--pool", type=int, default=12, help="distinct synthetic pairs cycled through")
    ap.add_argument("--streams", type=int, default=3,
                    help="pairs in flight per GPU: each on its own HIP stream + host thread (a 28 k-voxel pair "
                         "cannot fill 256 CUs alone, so independent pairs overlap)")
    ap.add_argument("--depth", type=int, default=None,
                    help="steps in flight PER STREAM: depth d puts d host workers (or scheduler slots) on every stream, so "
                         "the next steps' kernels are already queued behind the running one and the stream never waits "
                         "for the host between a step's device->host fetch and the following launch (--streams 1 "
                         "--depth 3: one HIP stream, the host up to three steps ahead).  Default: 1, except 3 with "
                         "--streams 1")
    ap.add_argument("--match-lanes", type=int, default=None,
                    help="deal the pairs of a step's matching + RANSAC batch over this many streams forked from the "
                         "step's stream inside libapr_hip (APR_MATCH_LANES; 0 = leave the library default, 1 lane).  Helps "
                         "--streams 1 (1613 -> 1896 pairs/s together with --depth 3), costs throughput once several steps "
                         "are in flight.  Default: 0, except 3 with --streams 1 (the plain one-stream loop: --streams 1 "
                         "--depth 1 --match-lanes 1)")
    ap.add_argument("--host", choices=["pipelined", "threads"], default="threads",
                    help="how the --streams steps in flight are driven: one Python thread per stream (default: the step's "
                         "host work is mostly inside library calls, which release the GIL, so three threads enqueue three "
                         "streams side by side: +6 %% over the single scheduler), or ONE host thread resuming each step "
                         "when its device->host fetch has landed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-workloads", action="store_true",
                    help="skip the extra workloads block (Predator config 3, FatBN-128, config 5), measured after the "
                         "headline loop")
    ap.add_argument("--pairs-per-step", type=int, default=1,
                    help="independent pairs batched into one encoder call per step (value counts pairs, not steps)")
    ap.add_argument("--pairs-total", type=int, default=0,
                    help="BASELINE config 4: register THIS MANY distinct pairs in all (seeds = global pair index), the pair "
                         "list cut into contiguous blocks over the ranks (shard.shard_range), poses gathered to every "
                         "rank at the end; value = pairs_total / max-over-ranks seconds, strong scaling.  --steps is "
                         "ignored (a rank runs ceil(block / pairs-per-step) steps)")
    ap.set_defaults(pairs_per_step=6)
    args = ap.parse_args()
    # a single stream gets the two remedies for a caller with one step in flight unless told otherwise (both are
    # reported in config: steps_in_flight_per_stream, match_lanes)
    one = args.streams == 1 and not args.pairs_total
    if args.depth is None:
        args.depth = 3 if one else 1
    if args.match_lanes is None:
        args.match_lanes = 3 if one else 0
    return args


def build_model(name, n_out, dev):
    from apr_amd.fcgf.model import load_model
    torch.manual_seed(0)
    m = load_model(name)(1, n_out, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    g = torch.Generator().manual_seed(0)
    for mod in m.modules():            # non-trivial eval-mode BN
        if isinstance(mod, torch.nn.BatchNorm1d):
            with torch.no_grad():
                mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
                mod.running_var.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
    return m.to(dev).eval()


def pmc_traffic(path):
    """(HBM bytes per launch of the kernel family `path` = "tile" | "ws" | "os", source file) from the committed PMC run of
    this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 x2 fetch correction; produced
    by scripts/profile_round.sh): the counters cannot be collected from inside the process, so bench.py reports the
    profiled value and names the file it came from."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_spconv_summary.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        ks = json.load(f).get("kernels", {})
    src = os.path.relpath(files[-1], ROOT)
    try:
        if path == "tile":
            return ks["k_spconv_pairs"]["hbm_bytes_per_launch"], src
        if path == "os":
            return ks["k_os_conv"]["hbm_bytes_per_launch"], src
        fam = json.load(open(files[-1])).get("ws_family")
        if fam:       # gemm (per-offset pairs or x-triple entries) + reduce, bytes of the family per layer
            return fam["hbm_bytes_per_layer"], src
        return ks["k_ws_gemm"]["hbm_bytes_per_launch"] + ks["k_ws_reduce"]["hbm_bytes_per_launch"], src
    except KeyError:
        return None, src


def cpu_baseline(state_dict, name, n_out, pairs, ransac_iters):
    """The oracle (a CPU port of the reference path) timed on this box's host cores over a bounded sample of the same
    workload: up to 8 of the bench's pairs, RANSAC on half of the iterations and scaled (about 10 s of CPU work)."""
    from oracle import match_pose_oracle as MO
    from oracle import me_oracle as OME
    from oracle import resunet_oracle as OR
    cores = MO.host_threads()      # this process's CPU share, not the host's core count
    torch.set_num_threads(cores)
    om = OR.MODELS[name](1, n_out, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    om.load_state_dict({k: v.cpu() for k, v in state_dict.items()})
    om.eval()
    pairs = pairs[:8]
    iters = min(ransac_iters, 2000000)
    t_enc = t_nn = t_rs = 0.0
    for xyz0, xyz1 in pairs:
        t0 = time.perf_counter()
        feats, pts = [], []
        for xyz in (xyz0, xyz1):
            c, sel = OME.sparse_quantize(xyz / np.float32(0.3), return_index=True)
            C = OME.batched_coordinates([c])
            with torch.no_grad():
                feats.append(om(OME.SparseTensor(np.ones((len(C), 1), np.float32), coordinates=C)).F.numpy())
            pts.append(xyz[sel])
        t1 = time.perf_counter()
        corr, _ = MO.feature_nn(feats[0], feats[1], nthreads=cores)
        t2 = time.perf_counter()
        MO.ransac_feature_matching(pts[0], pts[1], corr, 0.3, 0.9, max_iter=iters, seed=0)
        t3 = time.perf_counter()
        t_enc, t_nn, t_rs = t_enc + (t1 - t0), t_nn + (t2 - t1), t_rs + (t3 - t2)
    scale = ransac_iters / iters
    total = t_enc + t_nn + t_rs * scale
    n = len(pairs)
    return {"value": n / total, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": (f"{n} pairs of the timed workload: voxelise+encode 2 frames {t_enc / n:.2f}s, feature NN "
                       f"{t_nn / n:.2f}s, RANSAC {iters} of {ransac_iters} iterations {t_rs / n:.2f}s scaled "
                       f"x{scale:.0f} (per pair; {t_enc + t_nn + t_rs:.1f}s of CPU work in all)")}

PREDATOR_LIMITS = [58, 59, 58, 57]     # calibrate_neighbors (80th percentile) on this generator at full size


BF3_EQ_PEAK_TFLOPS = 2500.0 / 6.0     # dense bf16 MFMA peak / the 6 bf16 products of one fp32-equivalent product (bf3.h)


def conv_roofline(s, mfma_peak=None, mfma_peak_note=None):
    """Roofline entry of a profile summary (bytes / flops / ms / launches): the tighter of the HBM roof and the MFMA roof OF
    THE INSTRUCTIONS THE KERNELS ISSUE.  `mfma_peak` (fp32-equivalent TFLOP/s): a number, or None = weight the two MFMA
    forms by the time the summary's `by_path` spent on them (tile kernel: exact-fp32 MFMA, 157.3; weight-stationary /
    output-stationary / dense-bf3 kernels: bf16 MFMA in the 3-way split, 2.5 PFLOP/s / 6 = 417).  Both fractions are
    always reported (round-4 verdict: a bf16-split kernel priced against the fp32 peak read 2.6x too high)."""
    if mfma_peak is None:
        by = s.get("by_path") or {}
        tot = sum(d["ms"] for d in by.values())
        if tot > 0:
            w_tile = sum(d["ms"] for k, d in by.items() if k == "tile") / tot
            mfma_peak = 1.0 / (w_tile / MFMA_F32_PEAK_TFLOPS + (1.0 - w_tile) / BF3_EQ_PEAK_TFLOPS)
            mfma_peak_note = (f"time-weighted over the kernels' own instruction mix: {100 * w_tile:.0f} % of the time in the "
                              f"exact-fp32 tile kernel ({MFMA_F32_PEAK_TFLOPS} TF), the rest in bf16-split kernels "
                              f"({BF3_EQ_PEAK_TFLOPS:.0f} fp32-equivalent TF)")
        else:
            mfma_peak = MFMA_F32_PEAK_TFLOPS
    t_hbm = s["bytes"] / (HBM_PEAK_GBS * 1e9)
    t_mfma = s["flops"] / (mfma_peak * 1e12)
    sec = s["ms"] * 1e-3
    gbs, tf = s["bytes"] / sec / 1e9, s["flops"] / sec / 1e12
    if t_mfma > t_hbm:
        r = {"bound": "mfma", "achieved": tf, "peak": mfma_peak, "unit": "TFLOP/s", "frac": tf / mfma_peak}
    else:
        r = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
    r.update(traffic=None, hbm_gbs=gbs, hbm_frac=gbs / HBM_PEAK_GBS, mfma_tflops=tf, mfma_peak_of_issued_instructions=mfma_peak,
             mfma_frac=tf / mfma_peak, launches=s["launches"], avg_launch_us=1000.0 * s["ms"] / max(s["launches"], 1))
    if mfma_peak_note:
        r["mfma_peak_note"] = mfma_peak_note
    return r


def extra_workloads(dev, log, cpu_baselines=True):
    """The other single-GPU configurations of BASELINE.json, measured AFTER the headline loop (they do not touch it):
    config 3 (Predator_APR pair), APR's own encoder (ResUNetFatBN, 128 features) through the headline pipeline, and
    config 5 (distant pair: APG aggregation + FatBN encode + NPR loss).  Each with the roofline of its dominant
    kernel from HIP events on the launch stream."""
    from apr_amd import MinkowskiEngine as ME
    from apr_amd import ops, synth
    from apr_amd.fcgf.lib import apg


    from apr_amd.fcgf.pipeline import PairRegistration
    from apr_amd.predator import kp_ops
    from apr_amd.predator.configs.models import kitti_config
    from apr_amd.predator.models.architectures import KPFCNN
    from apr_amd.predator.pipeline import PredatorRegistration
    out = {}

    def sync():
        torch.cuda.synchronize()
        return time.perf_counter()

    a, b, _ = synth.make_pair(0)
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)

    # ---- config 3: Predator_APR pair, one pair at a time (grid subsample -> collate -> KPFCNN -> sampling -> RANSAC)
    np.random.seed(0)
    torch.manual_seed(0)
    cfg = kitti_config()
    pred = PredatorRegistration(KPFCNN(cfg).to(dev).eval(), cfg, PREDATOR_LIMITS)
    for i in range(3):
        pred(ta, tb, seed=i)
    reps = 10
    t0 = sync()
    for i in range(reps):
        T, info = pred(ta, tb, seed=i)
    t1 = sync()
    kp_ops.PROFILE = []
    src, tgt, feats, ov, sal = pred.encode(ta, tb)
    kp_records = list(kp_ops.PROFILE)
    ks = kp_ops.kpconv_profile_summary(kp_records)
    kp_ops.PROFILE = None
    # k_kpconv_weighted_mfma issues exact-fp32 MFMA (2 N H 15 (3 + cin) FLOP), k_dense_gemm_bf3 the bf16 split (2 N 15 cin cout):
    # the roof is the FLOP-weighted harmonic mean of the two peaks
    f1 = sum(2.0 * n_ * h_ * k_ * (3 + ci_) for (n_, h_, ci_, co_, k_, *_rest) in kp_records)
    f2 = sum(2.0 * n_ * k_ * ci_ * co_ for (n_, h_, ci_, co_, k_, *_rest) in kp_records)
    kp_peak = (f1 + f2) / (f1 / MFMA_F32_PEAK_TFLOPS + f2 / BF3_EQ_PEAK_TFLOPS)
    kr = conv_roofline(ks, mfma_peak=kp_peak, mfma_peak_note=(
        f"FLOP-weighted: {100 * f1 / (f1 + f2):.0f} % of the FLOP in the exact-fp32 correlation ({MFMA_F32_PEAK_TFLOPS} TF), the "
        f"rest in the bf16-split contraction ({BF3_EQ_PEAK_TFLOPS:.0f} fp32-equivalent TF)"))
    # counter traffic of the kernel-point correlation (k_kpconv_weighted_mfma: the gather + the [N, 15 cin] intermediate it
    # writes for step 2), per launch, from the committed PMC passes over scripts/kpconv_bench.py (the level mix of one pair)
    try:
        import glob
        f_ = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_spconv_summary.json")))[-1]
        kr["traffic"] = json.load(open(f_))["predator_kpconv"]["kernels"]["k_kpconv_weighted_mfma"]["hbm_bytes_per_launch"]
        kr["traffic_source"] = os.path.relpath(f_, ROOT) + " (k_kpconv_weighted_mfma per launch, one pair per call)"
    except (IndexError, KeyError, OSError):
        pass
    kr["kernel"] = ("KPConv layer = k_row_sums + k_kpconv_weighted_mfma (kernel-point correlation) + "
                    "k_dense_gemm_bf3 ([N, 15*cin] x [15*cin, cout], bf16 3-way split, fp32-equivalent); "
                    f"{ks['launches']} layers of one KPFCNN forward")
    # the same pipeline the way a registration service runs it: 8 pairs stacked per collate / KPFCNN forward
    # (per-pair InstanceNorm statistics and overlap attention: every pair gets its batch-of-one result), ONE host
    # thread keeping 4 batches in flight on 4 streams (register_batch_phases resumed when its fetches land), so that
    # one batch's host-RNG sampling and launch calls overlap the others' kernels
    from apr_amd.fcgf.pipeline import run_pipelined
    pool = [(ta, tb)]
    for sd in range(1, 8):
        pa, pb, _ = synth.make_pair(sd)
        pool.append((torch.from_numpy(pa).to(dev), torch.from_numpy(pb).to(dev)))
    # 8 batches in flight: 318 vs 295 pairs/s with 4 (the path is GPU-bound; more streams fill its gaps).  48 batches per run:
    # with 16 (rounds 2-4) filling and draining the 8-deep pipeline was a third of the run -- 384 pairs/s against 393 / 395 with
    # 48 / 96 batches on one box (scripts/predator_stacked_rate.py NBATCH=...)
    # 8 pairs per forward (round 5; 4 before): same box 407 / 419 / 435 / 433 / 439 pairs/s with 4 / 6 / 8 / 12 / 16 pairs stacked
    B, S, nbatch = 8, 8, 24
    batches = [[pool[(i * B + j) % len(pool)] for j in range(B)] for i in range(nbatch)]
    pstreams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    mk = lambda i: pred.register_batch_phases(batches[i], seeds=range(i * B, i * B + B))
    rates, host_ms = [], []
    for rep in range(4):                # rep 0: every stream sees the batch compositions (allocator, caches)
        t0s = sync()
        run_pipelined(mk, range(nbatch), pstreams)
        t1s = sync()
        if rep:
            rates.append(nbatch * B / (t1s - t0s))
            host_ms.append(1e3 * run_pipelined.last_host_busy_s / (nbatch * B))
    rates.sort()
    out["predator_config3"] = {
        "workload": "Predator_APR KPConv encoder + overlap attention + score sampling + RANSAC(50000, 1000) on "
                    "2 x 118 k-point pairs (8 distinct synthetic pairs); 8 pairs stacked per forward, one host thread "
                    "keeping 8 batches in flight on 8 streams; median of 3 runs of 192 pairs",
        "value": rates[1], "unit": "pairs/s", "ms_per_pair": 1e3 / rates[1], "runs_pairs_per_s": rates,
        "host_enqueue_ms_per_pair": sorted(host_ms)[1],
        "one_pair_at_a_time": {"value": reps / (t1 - t0), "unit": "pairs/s", "ms_per_pair": 1e3 * (t1 - t0) / reps},
        "points_after_0.3m_grid": [int(len(src)), int(len(tgt))], "neighbor_limits": PREDATOR_LIMITS,
        "roofline": kr}
    log(f"workloads: predator {out['predator_config3']['value']:.1f} pairs/s stacked, pipelined, "
        f"{reps / (t1 - t0):.1f} one pair at a time")
    if cpu_baselines:
        # the CPU path beside it (round-4 verdict): ONE pair of the workload -- the reference's own C++ (oracle/_ref) for the
        # 0.3 m grid, the 10 neighbour tables and the 3 sub-samplings, then the oracle's KPFCNN forward (torch CPU ops,
        # arithmetic-identical to the imported reference, tests/test_predator_oracle_cpu.py) on all host threads
        try:
            from oracle import kpfcnn_oracle as KO
            from oracle import match_pose_oracle as MO
            from oracle import predator_points_oracle as PREF
            if not PREF.available():
                raise RuntimeError("oracle/_ref not built")
            nthr = MO.host_threads()
            torch.set_num_threads(nthr)
            sd_cpu = {k: v.cpu() for k, v in pred.model.state_dict().items()}
            c0 = time.perf_counter()
            cp, cl = PREF.subsample_batch(np.concatenate([a, b]), np.array([len(a), len(b)], np.int32), sampleDl=0.3)
            bo = KO.collate(cp[:cl[0]], cp[cl[0]:], cfg, PREDATOR_LIMITS)
            c1 = time.perf_counter()
            with torch.no_grad():
                KO.kpfcnn_forward(sd_cpu, cfg, bo)
            c2 = time.perf_counter()
            out["predator_config3"]["cpu_baseline"] = {
                "value": 1.0 / (c2 - c0), "unit": "pairs/s", "cores": nthr, "kind": "reference+port",
                "sample": (f"1 pair of the timed workload: index build with the reference's own C++ core (single thread, as the "
                           f"reference runs it) {c1 - c0:.2f}s + KPFCNN forward of the CPU oracle on {nthr} threads {c2 - c1:.2f}s; "
                           "score sampling and RANSAC(50000, 1000) not included (open3d absent)")}
            log(f"workloads: predator cpu baseline {1.0 / (c2 - c0):.3f} pairs/s ({c1 - c0:.2f}s index build + {c2 - c1:.2f}s forward)")
        except Exception as e:      # noqa: BLE001 -- the baseline is a report, never a reason to lose the bench line
            out["predator_config3"]["cpu_baseline"] = {"value": None, "error": repr(e)}

    # ---- APR's encoder (FatBN, 128 features) through the headline pipeline: 6 pairs per call, 3 steps in flight (one host
    # thread resuming each step when its fetch has landed: the same steps as the headline's three threads), and one stream
    torch.manual_seed(0)
    fat = build_model("ResUNetFatBN", 128, dev)
    pipe = PairRegistration(fat, voxel_size=0.3, ransac_iters=4000000)
    pool6 = [(ta, tb)] + [tuple(torch.from_numpy(x).to(dev) for x in synth.make_pair(s)[:2]) for s in range(1, 6)]
    fstreams = [torch.cuda.Stream(device=dev) for _ in range(3)]

    def pipelined_rate(p, nsteps=48, B6=6):
        """pairs/s of `p` over nsteps steps of B6 pairs, 3 steps in flight; median of 3 runs after a priming run (48 steps: fill
        and drain of the pipeline are ~3 % of a run; they were 5-6 % of the 24-step runs of rounds 2-4)."""
        mk6 = lambda i: p.register_batch_phases([pool6[(i * B6 + j) % len(pool6)] for j in range(B6)],
                                                seeds=[i * B6 + j for j in range(B6)])
        rates6 = []
        for rep in range(4):
            t0s = sync()
            res6, _ = run_pipelined(mk6, range(nsteps), fstreams)
            t1s = sync()
            if rep:
                rates6.append(nsteps * B6 / (t1s - t0s))
        return sorted(rates6)[1], res6

    batch = [pool6[j % len(pool6)] for j in range(6)]
    for i in range(3):
        pipe.register_batch(batch, seeds=list(range(6)))
    steps = 10
    t0 = sync()
    for i in range(steps):
        pipe.register_batch(batch, seeds=[6 * i + j for j in range(6)])
    t1 = sync()
    fat_rate, _ = pipelined_rate(pipe)
    prof = ops.SpconvProfile()
    ops.PROFILE = prof
    for i in range(3):
        pipe.encode_batch(pipe.voxelize_batch([c for p in batch for c in p])[0])
    ops.PROFILE = None
    fs = prof.summary()
    fr = conv_roofline(fs)
    fr["kernel"] = "all MFMA conv layers of the ResUNetFatBN encode (12 frames per call)"
    fr["by_path_us_per_encode"] = {k: 1e3 * d["ms"] / 3 for k, d in fs["by_path"].items()}
    out["fcgf_fatbn128"] = {
        "workload": "FCGF_APR encode+match+SVD with APR's encoder (ResUNetFatBN, 128-d features, "
                    "scripts/train_apr_kitti.sh:12-13), 6 pairs per step, 3 steps in flight (one host thread)",
        "value": fat_rate, "unit": "pairs/s", "ms_per_step": 6e3 / fat_rate,
        "one_stream": {"value": steps * 6 / (t1 - t0), "unit": "pairs/s", "ms_per_step": 1e3 * (t1 - t0) / steps},
        "roofline": fr}
    log(f"workloads: fatbn128 {fat_rate:.1f} pairs/s (3 in flight), {steps * 6 / (t1 - t0):.1f} one stream")

    # ---- the headline pipeline with descriptors that MATCH: the encoder (BN2C / 32, the headline's) runs, then its output
    # rows are overwritten with descriptors carrying 30 % ground-truth matches (what a trained checkpoint's features look
    # like to the matcher): NN + RANSAC now have ~32 k surviving hypotheses per pair instead of ~200
    from apr_amd.fcgf.registration import rte_rre
    bn2c = build_model("ResUNetBN2C", 32, dev)
    ppipe = PairRegistration(bn2c, voxel_size=0.3, ransac_iters=4000000)
    planted, gts = {}, {}
    gen = torch.Generator(device="cpu").manual_seed(0)
    for s_, (pa, pb) in enumerate(pool6):
        T_s = synth.make_pair(s_)[2]
        _, q0, q1, m0, m1 = ppipe.voxelize_pair(pa, pb)
        gtp = apg.get_matching_indices(q0, q1, T_s, 0.3, K=1)
        G1 = torch.nn.functional.normalize(torch.randn(m1, 32, generator=gen), dim=1).to(dev)
        G0 = torch.nn.functional.normalize(torch.randn(m0, 32, generator=gen), dim=1).to(dev)
        pk = gtp[torch.randperm(len(gtp), generator=gen)[:int(0.3 * m0)].to(dev)]
        G0[pk[:, 0]] = torch.nn.functional.normalize(G1[pk[:, 1]] + 0.02 * torch.randn(len(pk), 32, generator=gen).to(dev), dim=1)
        planted[(pa.data_ptr(), pb.data_ptr())] = (G0, G1)
        gts[(pa.data_ptr(), pb.data_ptr())] = T_s

    def plant(F, counts, prs):
        o_ = 0
        for i_, (pa, pb) in enumerate(prs):
            G0, G1 = planted[(pa.data_ptr(), pb.data_ptr())]
            assert counts[2 * i_] == len(G0) and counts[2 * i_ + 1] == len(G1)
            F[o_:o_ + len(G0)] = G0
            F[o_ + len(G0):o_ + len(G0) + len(G1)] = G1
            o_ += len(G0) + len(G1)
        return F

    base_rate, _ = pipelined_rate(ppipe)
    # ---- BASELINE config 2 read literally (FCGF_APR/scripts/test_apr.py:111-163 runs batch size 1): ONE pair per call, one
    # stream, nothing else in flight -- voxelise both frames, one encoder call on the two frames, NN, RANSAC(4 M), pose
    # fetched before the next pair starts
    def one_at_a_time(n1=60):
        for i_ in range(6):
            ppipe.register_batch([pool6[i_]], seeds=[i_])
        t0 = sync()
        for i_ in range(n1):
            ppipe.register_batch([pool6[i_ % 6]], seeds=[i_])
        return n1 / (sync() - t0)

    r_poll = one_at_a_time()            # the process default: sleeping poll (what the headline's ranks use)
    ppipe.fetch_wait = "sync"           # a caller with one step in flight: hipEventSynchronize, one spinning core
    r_sync = one_at_a_time()
    ppipe.fetch_wait = None
    n1, t0, t1 = 60, 0.0, 60 / r_sync
    out["fcgf_one_pair"] = {
        "workload": "BASELINE config 2 literally: FCGF_APR encode+match+SVD (ResUNetBN2C / 32, RANSAC 4 M), ONE 2 x 118 k-point "
                    "pair per call on one stream, the pose on the host before the next pair starts (the reference loop's "
                    "shape, test_apr.py:111-163); 6 distinct pairs cycled",
        "value": n1 / (t1 - t0), "unit": "pairs/s", "ms_per_pair": 1e3 * (t1 - t0) / n1,
        "fetch_wait": "hipEventSynchronize (PairRegistration.fetch_wait = 'sync': one spinning core, lowest wake-up latency)",
        "with_sleeping_poll": {"value": r_poll, "unit": "pairs/s", "ms_per_pair": 1e3 / r_poll}}
    log(f"workloads: one pair at a time {n1 / (t1 - t0):.1f} pairs/s ({1e3 * (t1 - t0) / n1:.3f} ms per pair)")
    ppipe.feature_hook = plant
    plant_rate, res_p = pipelined_rate(ppipe)
    errs = []
    for i_, lst in res_p.items():
        for j_, (T_, info_) in enumerate(lst):
            pa, pb = pool6[(i_ * 6 + j_) % len(pool6)]
            errs.append(rte_rre(T_, gts[(pa.data_ptr(), pb.data_ptr())]) + (info_["n_valid"],))
    errs = np.array(errs)
    out["fcgf_planted_30pct"] = {
        "workload": "the headline pipeline (ResUNetBN2C / 32, 6 pairs per step, 3 steps in flight, one host thread) with the "
                    "encoder's output rows overwritten by descriptors carrying 30 % ground-truth matches: the matcher "
                    "works as it would behind a trained checkpoint",
        "value": plant_rate, "unit": "pairs/s", "same_harness_random_init_features": base_rate,
        "mean_valid_hypotheses": float(errs[:, 2].mean()), "max_rte_m": float(errs[:, 0].max()),
        "max_rre_deg": float(errs[:, 1].max())}
    log(f"workloads: planted 30 % {plant_rate:.1f} pairs/s vs {base_rate:.1f} with random-init features (same harness)")

    # ---- matching + pose under load: correspondences with a CONTROLLED share of true matches.  A random-init encoder
    # (the headline loop) gives collapsed features: nearly every one of the 4 M hypotheses dies in the edge-length
    # check and the scoring kernels idle.  Trained features put 10-60 % true matches into the correspondence set;
    # the survivors (and the cost of scoring them, survivors x correspondences) grow with the 4th power of that share.
    a_h, b_h, T_gt = synth.make_pair(0)
    base = PairRegistration(fat, voxel_size=0.3, ransac_iters=4000000)
    _, pts0, pts1, n0, n1 = base.voxelize_pair(ta, tb)
    pairs_gt = apg.get_matching_indices(pts0, pts1, T_gt, 0.3, K=1)           # true matches: nearest within a voxel
    g = torch.Generator(device="cpu").manual_seed(0)
    F1 = torch.nn.functional.normalize(torch.randn(n1, 32, generator=g), dim=1).to(dev)
    sweep = []
    for share in (0.1, 0.3, 0.5):
        F0 = torch.nn.functional.normalize(torch.randn(n0, 32, generator=g), dim=1).to(dev)
        k = int(share * n0)
        pick = pairs_gt[torch.randperm(len(pairs_gt), generator=g)[:k].to(dev)]
        F0[pick[:, 0]] = torch.nn.functional.normalize(
            F1[pick[:, 1]] + 0.02 * torch.randn(len(pick), 32, generator=g).to(dev), dim=1)
        ops.match_pose_batch([F0], [F1], [pts0], [pts1], 0.3, 0.9, 4000000, seeds=[0])
        t0 = sync()
        for r_ in range(3):
            (T_est, info), = ops.match_pose_batch([F0], [F1], [pts0], [pts1], 0.3, 0.9, 4000000, seeds=[r_])
        t1 = sync()
        rte, rre = rte_rre(T_est, T_gt)
        sweep.append({"true_match_share": len(pick) / n0, "ms_per_pair_nn_plus_ransac": 1e3 * (t1 - t0) / 3,
                      "valid_hypotheses": info["n_valid"], "inliers": info["inliers"], "rte_m": rte, "rre_deg": rre})
    out["matching_under_load"] = {
        "workload": "feature NN + RANSAC(4 M) + Kabsch on one pair's ~14 k x 14 k voxels, synthetic 32-d descriptors "
                    "with a controlled share of ground-truth matches (the headline's random-init features: ~0 %)",
        "sweep": sweep}
    log(f"workloads: matching under load {[round(x['ms_per_pair_nn_plus_ransac'], 2) for x in sweep]} ms")

    # ---- config 5: distant pair (16-beam source vs 64-beam target, 40 m), APG over 10 complement frames + NPR loss
    xyz0, _, _ = synth.make_pair(0, n_beams1=16, dist=40.0)
    rng = np.random.default_rng(1000)
    scene = synth.make_scene(0)
    frames, poses = [], []
    for j in list(range(-5, 0)) + list(range(1, 6)):
        frames.append(torch.from_numpy(synth.raycast(scene, (6.0 * j, 0.0, 0.0), 0.0, rng, 16, 1875)).to(dev))
        M = np.eye(4)
        M[0, 3] = 6.0 * j
        poses.append(M)
    key = torch.from_numpy(xyz0).to(dev)
    gen = apg.GenerativeMLP_98(in_channel=128, out_points=4).to(dev).eval()
    rows = []
    for rep in range(4):
        t0 = sync()
        nghb, sel = apg.aggregate_frames(key, frames, poses, 0.3)
        cloud = nghb[sel]
        t1 = sync()
        m = ops.build_map(ops.voxelize(key, 0.3, 0), want_first=True)
        ops.finalize_maps([m])
        with torch.no_grad():
            F = fat(ME.SparseTensor(torch.ones((m.n, 1), device=dev), coordinates=m.coords)).F
        t2 = sync()
        loss = apg.npr_reconstruction_loss(gen, F, m.coords[:, 1:], cloud, 0.3, 4)
        t3 = sync()
        if rep:
            rows.append([t1 - t0, t2 - t1, t3 - t2])
    # the distant pair itself: 16-beam source (~30 k points) against the 64-beam target 40 m away, both encodes + NN +
    # RANSAC(4 M); then with geometry-derived descriptors planted (30 % of the source voxels' true matches) the pose
    xyz0_h, xyz1_h, T5 = synth.make_pair(0, n_beams1=16, dist=40.0)
    d0, d1 = torch.from_numpy(xyz0_h).to(dev), torch.from_numpy(xyz1_h).to(dev)
    p5 = PairRegistration(fat, voxel_size=0.3, ransac_iters=4000000)
    for _ in range(2):
        p5.register_batch([(d0, d1)], seeds=[0])
    t0 = sync()
    for r_ in range(5):
        p5.register_batch([(d0, d1)], seeds=[r_])
    t_pair = (sync() - t0) / 5
    _, q0, q1, m0, m1 = p5.voxelize_pair(d0, d1)
    gt5 = apg.get_matching_indices(q0, q1, T5, 0.3, K=1)
    H1 = torch.nn.functional.normalize(torch.randn(m1, 128, generator=g), dim=1).to(dev)
    H0 = torch.nn.functional.normalize(torch.randn(m0, 128, generator=g), dim=1).to(dev)
    pk5 = gt5[torch.randperm(len(gt5), generator=g)[:int(0.3 * m0)].to(dev)]
    H0[pk5[:, 0]] = torch.nn.functional.normalize(H1[pk5[:, 1]] + 0.02 * torch.randn(len(pk5), 128, generator=g).to(dev), dim=1)

    def plant5(F, counts, prs):
        F[:m0] = H0
        F[m0:m0 + m1] = H1
        return F

    p5.feature_hook = plant5
    p5.register_batch([(d0, d1)], seeds=[0])
    t0 = sync()
    for r_ in range(5):
        (T5e, info5), = p5.register_batch([(d0, d1)], seeds=[r_])
    t_pair_m = (sync() - t0) / 5
    rte5, rre5 = rte_rre(T5e, T5)
    r = np.array(rows).mean(0) * 1e3
    out["config5_distant_pair"] = {
        "registration": {
            "workload": "the distant pair end to end (FatBN-128 encode of both frames + feature NN + RANSAC 4 M + Kabsch), one "
                        "pair per call, one stream",
            "ms_per_pair_random_init_features": 1e3 * t_pair, "ms_per_pair_30pct_true_matches": 1e3 * t_pair_m,
            "voxels": [int(m0), int(m1)], "density_ratio": float(m1) / float(m0), "true_matches_planted": int(len(pk5)),
            "valid_hypotheses": int(info5["n_valid"]), "inliers": int(info5["inliers"]), "rte_m": rte5, "rre_deg": rre5},
        "workload": "LoNuScenes-shaped distant pair, source side: APG aggregation of 10 complement frames + FatBN-128 "
                    "encode of the 16-beam key frame + NPR decoder (GenerativeMLP_98, ratio 4) reconstruction loss",
        "value": float(r.sum()), "unit": "ms", "higher_is_better": False,
        "stages_ms": {"apg_aggregate": float(r[0]), "voxelise+encode": float(r[1]), "npr_mlp+chamfer+reg": float(r[2])},
        "key_points": int(len(xyz0)), "complement_points": int(sum(len(f) for f in frames)),
        "apg_points": int(len(cloud)), "key_voxels": int(m.n), "loss": float(loss)}
    log(f"workloads: config5 {r.sum():.2f} ms")
    # ---- the APR training iteration (round-4 verdict item 2): ms per iteration + stage split
    from apr_amd.fcgf.lib import complement_trainer as CT
    del fat, p5
    torch.cuda.empty_cache()
    out["apr_train_step"] = CT.measure(dev, iters=8, log=log)
    log(f"workloads: apr_train_step {out['apr_train_step']['value']:.2f} ms per iteration")
    return out


def main():
    args = parse()
    # --gpus N without a launcher around it: start the N ranks ourselves, one fresh child process of this script per GPU
    # (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run would set them), relay rank 0's JSON line and
    # fail if any rank does.  Nothing has touched the GPU in this process yet, and it never will: children only, no exec.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from apr_amd import shard
        print(f"[bench] --gpus {args.gpus} without WORLD_SIZE: launching {args.gpus} ranks of this script",
              file=sys.stderr, flush=True)
        raise SystemExit(shard.launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must "
                         "agree (the JSON line reports n_gpus = WORLD_SIZE)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # rehearsal switches (1-GPU box): APR_BENCH_BACKEND=gloo + APR_BENCH_SINGLE_DEVICE=1 run N ranks on cuda:0
    backend = os.environ.get("APR_BENCH_BACKEND", "nccl")
    if os.environ.get("APR_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)     # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    from apr_amd import _host, ops, shard, synth
    if args.match_lanes > 0:
        ops.set_match_lanes(args.match_lanes)
    from apr_amd.fcgf.pipeline import PairRegistration

    model = build_model(args.model, args.n_out, dev)
    pipe = PairRegistration(model, voxel_size=0.3, ransac_iters=args.ransac_iters)

    # synthetic pairs of this rank (seeds 64*rank + i); upload before timing.  The pool size does NOT
    # depend on --warmup: every distinct batch composition is run on every worker stream during the untimed setup
    # below, whatever the driver passes.
    cfg4 = args.pairs_total > 0
    if cfg4:     # config 4: this rank's contiguous block of the global pair list, every pair distinct
        blk_lo, blk_hi = shard.shard_range(args.pairs_total, rank, world)
        seeds4 = list(range(blk_lo, blk_hi))
        if not seeds4:
            raise SystemExit("--pairs-total smaller than the number of ranks")
    npool = len(seeds4) if cfg4 else max(1, args.pool)
    host_pairs = [synth.make_pair(s)[:2] for s in (seeds4 if cfg4 else shard.rank_seeds(rank, npool))]
    pairs = [(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)) for a, b in host_pairs]
    n_pts = float(np.mean([len(a) + len(b) for a, b in host_pairs])) / 2

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    # one-off allocator warm-up (setup, not a step): cache a large HBM segment so the timed region
    # never falls into hipMalloc when a pair's voxel counts differ from the previous pair's
    _reserve = torch.empty(4 << 30, dtype=torch.uint8, device=dev)
    del _reserve
    # ... and prime the HIP runtime's launch queue (kernarg / signal pools grow once, ~40 ms, after
    # a few hundred un-synchronised launches): 2000 trivial launches on a 4-element tensor
    _tiny = torch.zeros((4, 1), dtype=torch.float32, device=dev)
    for _ in range(2000):
        ops.affine_act(_tiny, relu=True, out=_tiny)
    torch.cuda.synchronize()

    B = max(1, args.pairs_per_step)

    poses4 = {}        # config 4: local pair index -> 4x4 pose

    def step(i):
        if cfg4:        # step i of the block: pairs [i B, min((i + 1) B, block)); priming / warm-up steps wrap around
            n4 = len(pairs)
            nsteps4 = -(-n4 // B)
            j0 = (i % nsteps4) * B
            idx = list(range(j0, min(j0 + B, n4)))
            res = pipe.register_batch([pairs[j] for j in idx], seeds=[seeds4[j] for j in idx])
            for j, (T_, _) in zip(idx, res):
                poses4[j] = T_
            return res[-1]
        if B == 1:
            a, b = pairs[i % len(pairs)]
            return pipe(a, b, seed=i)
        batch = [pairs[(i * B + j) % len(pairs)] for j in range(B)]
        return pipe.register_batch(batch, seeds=[i * B + j for j in range(B)])[-1]

    # S independent steps in flight: worker w owns HIP stream w and runs steps w, w+S, w+2S, ...  The worker threads
    # are PERSISTENT (created once, here in the setup): no thread, stream, HSA queue or allocator pool is first touched
    # inside the timed region.
    import threading
    if os.environ.get("APR_BENCH_SWITCH"):
        sys.setswitchinterval(float(os.environ["APR_BENCH_SWITCH"]))
    nhip = max(1, args.streams)
    depth = max(1, args.depth)
    hip_streams = [torch.cuda.Stream(device=dev) for _ in range(nhip)]
    # worker / slot w enqueues on stream w % nhip.  Workers that share a stream interleave their launches on it: every
    # step's scratch comes from torch's per-stream caching allocator at call time (no cached workspaces, no static device
    # state in libapr_hip), so FIFO order on the stream keeps a block's reuse behind its last reader.
    nstreams = nhip * depth
    streams = [hip_streams[w % nhip] for w in range(nstreams)]
    results = {}
    step_log = []      # (step, worker, host start, host end) of every step
    job = {"first": 0, "last": 0, "stop": False, "err": None, "next": 0, "stagger": 0.0}
    take = threading.Lock()
    go = threading.Barrier(nstreams + 1)
    done = threading.Barrier(nstreams + 1)

    def worker(w):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[w]):
            while True:
                go.wait()
                if job["stop"]:
                    return
                try:
                    # Staggered start: released together, the workers run their first steps in the same phase -- three
                    # front ends, then three encoders, then three matchers, and three Python phases queueing on the GIL
                    # (step log of round 2's driver runs: the first three steps took 10-12 ms each against 7.5 ms for
                    # three steps in steady state, and the convoy lasted two more rounds).  In steady state steps
                    # complete one interval apart; worker w therefore starts w intervals late, which is where the
                    # pipeline would have it anyway.  The wait is inside the timed region.
                    if job["stagger"] > 0 and w > 0:
                        time.sleep(w * job["stagger"])
                    while True:      # the next unclaimed step (a static deal leaves workers idle at the tail)
                        with take:
                            i = job["next"]
                            job["next"] = i + 1
                        if i >= job["last"]:
                            break
                        ta = time.perf_counter()
                        results[i] = step(i)
                        step_log.append((i, w, ta, time.perf_counter()))
                    streams[w].synchronize()
                except BaseException as e:      # surface worker failures instead of hanging the barrier
                    job["err"] = e
                done.wait()

    if cfg4:
        args.steps = -(-len(pairs) // B)
    pipelined = args.host == "pipelined" and B > 1 and not cfg4
    threads = [] if pipelined else [threading.Thread(target=worker, args=(w,), daemon=True) for w in range(nstreams)]
    [t.start() for t in threads]

    host_busy = {"s": None, "steps": 0}     # time the scheduling thread spent enqueueing (inside the generators)

    def make_step(i):
        batch = [pairs[(i * B + j) % len(pairs)] for j in range(B)]
        return pipe.register_batch_phases(batch, seeds=[i * B + j for j in range(B)])

    def run_steps(first, last):
        if pipelined:       # one host thread, `nstreams` steps in flight (apr_amd.fcgf.pipeline.run_pipelined)
            from apr_amd.fcgf.pipeline import run_pipelined
            res, done_at = run_pipelined(make_step, range(first, last), streams)
            host_busy["s"], host_busy["steps"] = run_pipelined.last_host_busy_s, last - first
            for i, r in res.items():
                results[i] = r[-1]
            step_log.extend((i, 0, t, t) for i, t in done_at)
            for st in hip_streams:
                st.synchronize()
            return
        job["first"], job["last"], job["next"] = first, last, first
        go.wait()
        done.wait()
        if job["err"] is not None:
            raise job["err"]

    # untimed setup, independent of --warmup: lazily built caches (packed weights, folded BN) on one thread, then every
    # distinct batch composition (the pool repeats with period `ncomp` steps) at least twice on EVERY worker stream
    step(0)
    torch.cuda.synchronize()
    ncomp = max(1, npool // int(np.gcd(B, npool)))
    nprime = 2 * nstreams * (min(ncomp, 2) if cfg4 else ncomp)
    if cfg4:            # the timed region must start at step 0 of the block: whole passes over the block before it
        nprime = -(-nprime // args.steps) * args.steps
        args.warmup = -(-args.warmup // args.steps) * args.steps if args.warmup else 0
    tp = time.perf_counter()
    run_steps(0, nprime)
    log(f"setup: {nprime} priming steps on {nstreams} worker stream(s) in {time.perf_counter() - tp:.3f}s")
    # the steady-state interval between step completions, from the priming run (the start stagger of the workers)
    ends0 = sorted(tb for _, _, _, tb in step_log)
    gaps0 = np.diff(np.array(ends0)) if len(ends0) > 2 else np.array([0.0])
    stagger_env = os.environ.get("APR_BENCH_STAGGER_MS")
    job["stagger"] = (float(stagger_env) * 1e-3 if stagger_env is not None
                      else (float(np.median(gaps0)) if not pipelined and nstreams > 1 else 0.0))
    log(f"setup: worker start stagger {1e3 * job['stagger']:.3f} ms")
    # ... then the W warm-up steps the contract asks for
    run_steps(nprime, nprime + args.warmup)
    step_log.clear()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    first = nprime + args.warmup
    poses4.clear()
    barrier()
    thr0 = _host.cgroup_throttle()           # the container's CPU-quota throttling so far (diagnostic: see config)
    cpu0 = time.process_time()               # CPU seconds of ALL threads of this rank (user + system)
    samp0 = ops.ransac_sampling_launches()
    t0 = time.perf_counter()
    run_steps(first, first + args.steps)     # EXACTLY `steps` steps, `streams` in flight
    T, info = results[first + args.steps - 1]
    torch.cuda.synchronize()
    barrier()
    elapsed_local = time.perf_counter() - t0
    samp1 = ops.ransac_sampling_launches()
    n_chk, n_scr = samp1[0] - samp0[0], samp1[1] - samp0[1]      # what the library launched inside the timed region
    sampling_kernel_ran = ("k_sample_screen" if n_chk == 0 and n_scr > 0 else "k_sample_check" if n_scr == 0 and n_chk > 0
                           else f"k_sample_screen x{n_scr} + k_sample_check x{n_chk}")
    host_cpu_s = time.process_time() - cpu0
    thr1 = _host.cgroup_throttle()
    elapsed = shard.max_over_ranks(elapsed_local, dev if backend == "nccl" else torch.device("cpu"))   # all-reduce(MAX)
    if not pipelined:
        job["stop"] = True
        go.wait()

    log(f"timed loop: {args.steps} steps in {elapsed:.3f}s with {nstreams} stream(s)")
    all_poses = None
    if cfg4:            # poses of the whole pair list on every rank (one all_gather of [ceil(N / world), 4, 4] per rank)
        mine = np.stack([np.asarray(poses4[j], dtype=np.float64) for j in range(len(pairs))])
        all_poses = shard.gather_poses(mine, args.pairs_total, dev if backend == "nccl" else torch.device("cpu"))
        assert tuple(all_poses.shape) == (args.pairs_total, 4, 4)
    # steady state: median interval between step completions (each completion = one step of B pairs), next to the
    # mean over the whole timed region that `ms_per_step` reports
    ends = sorted(tb for _, _, _, tb in step_log)
    gaps = np.diff(np.array([t0] + ends))
    steady_ms = 1e3 * float(np.median(gaps)) if len(gaps) else float("nan")
    if os.environ.get("APR_BENCH_STEPLOG"):
        for i, w, ta, tb in sorted(step_log):
            log(f"  step {i - first:4d} worker {w} start {1e3 * (ta - t0):8.2f} ms  host {1e3 * (tb - ta):7.2f} ms")
    log(f"steady state: median step-completion interval {steady_ms:.3f} ms vs mean {1e3 * elapsed / args.steps:.3f} ms")
    # end-of-run stats of every rank: one all_gather of a few floats
    stats = shard.gather_stats([len(pairs) if cfg4 else args.steps * B, elapsed_local, float(info["n_valid"]), host_cpu_s],
                               dev if backend == "nccl" else torch.device("cpu"))
    out = {
        "metric": "point-cloud pairs/sec (120k-pt KITTI frame, FCGF encode+match+SVD)",
        "value": (args.pairs_total if cfg4 else world * args.steps * B) / elapsed,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        "steady_ms_per_step": steady_ms,
        "higher_is_better": True,
        "scaling": "strong" if cfg4 else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"FCGF_APR encode+match+SVD on 120k-point KITTI-shaped pairs, voxel_size=0.3; {B} independent pair(s) "
                               f"per step share one batched encoder call (2x{B} frames), then NN + RANSAC per pair",
                   "encoder": args.model, "feature_dim": args.n_out, "points_per_frame": int(n_pts),
                   "voxels_per_frame": int((info["n0"] + info["n1"]) / 2), "ransac_iterations": args.ransac_iters,
                   "pairs_per_step": B, "streams_per_gpu": nhip, "steps_in_flight_per_stream": depth,
                   "match_lanes": args.match_lanes or int(os.environ.get("APR_MATCH_LANES", "1")),
                   "ransac_sampling_kernel": sampling_kernel_ran,
                   "ransac_sampling_kernel_requested": ("k_sample_check" if os.environ.get("APR_RANSAC_SCREEN") == "0" else "k_sample_screen"),
                   "host": "one thread, steps resumed on fetch completion" if pipelined else f"{nstreams} threads",
                   "sharding": f"{world} ranks x independent pairs",
                   "host_enqueue_ms_per_step": (None if host_busy["s"] is None
                                                else 1e3 * host_busy["s"] / max(host_busy["steps"], 1)),
                   "pool_pairs": npool, "ransac_valid_hypotheses_last_pair": int(info["n_valid"]),
                   **({"pairs_total": args.pairs_total, "pairs_this_rank": len(pairs),
                       "poses_gathered": list(all_poses.shape),
                       "mode": "BASELINE config 4: the pair list cut into contiguous blocks over the ranks, seeds = global "
                               "pair index, every pair registered once, poses gathered at the end"} if cfg4 else {}),
                   # host cost of a rank: CPU seconds (all threads, user + system) per step and as a share of one CPU over
                   # the timed region; 8 ranks need 8 x host_cpus_busy CPUs of the node's quota
                   "host_cpu_s_per_step": host_cpu_s / args.steps,
                   "host_cpus_busy": host_cpu_s / elapsed_local,
                   # a timed window can be hit by something that is not this program: the container's CPU quota running out
                   # (every thread of the cgroup stalls until the 100 ms period ends).  The median step-completion interval
                   # next to ms_per_step's mean, and the throttling the cgroup saw inside the window, say whether it was
                   "step_interval_median_ms": steady_ms,
                   "cgroup_throttled_in_window": (None if thr0 is None or thr1 is None else
                                                  {"periods": thr1[0] - thr0[0], "ms": (thr1[1] - thr0[1]) / 1e3}),
                   "host_cpu_quota": _host.cpu_quota(), "fetch_wait": (f"poll {1e6 * ops.FETCH_POLL_S:.0f} us" if ops.FETCH_WAIT != "sync" else "hipEventSynchronize"),
                   "per_rank_host_cpus_busy": [float(r[3] / r[1]) for r in stats.tolist()],
                   "per_rank_pairs_per_s": [float(r[0] / r[1]) for r in stats.tolist()]},
    }

    if rank == 0 and not args.no_roofline:
        # instrumented pass: HIP events around every sparse-conv launch, on the launch stream
        prof = ops.SpconvProfile()
        ops.PROFILE = prof
        nprof = max(1, min(args.steps, 5))
        for i in range(nprof):       # the same encoder call as the timed steps: B pairs = 2B frames per call
            batch = [pairs[(i * B + j) % len(pairs)] for j in range(B)]
            cm = pipe.voxelize_batch([c for p in batch for c in p])[0]
            pipe.encode_batch(cm)
        ops.PROFILE = None
        s = prof.summary()
        log(f"roofline pass: {s}")
        gbs = s["bytes"] / (s["ms"] * 1e-3) / 1e9

        def leg(d):
            g = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            return {"launches_per_encode": d["launches"] // nprof, "achieved": g, "frac": g / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                    "avg_launch_us": 1000.0 * d["ms"] / d["launches"],
                    "mfma_tflops": d["flops"] / (d["ms"] * 1e-3) / 1e12}

        names = {"tile": "k_spconv_pairs", "ws": "k_ws_gemm_bf3|k_ws3_gemm_bf3+k_ws_reduce", "os": "k_os_conv"}
        what = {"tile": "pair-compacted gather -> fp32 MFMA -> fused epilogue, one kernel",
                "ws": "weight-stationary gather -> bf16 MFMA in an exact 3-way split (k_ws3_gemm_bf3 over x-triple entries: "
                      "one product row per entry; k_ws_gemm_bf3 over per-offset pairs for 256 input channels), product rows "
                      "through HBM, then per-row sum + fused epilogue (k_ws_reduce)",
                "os": "output-stationary: tile accumulators in LDS, gather -> bf16 MFMA in an exact 3-way split -> fused "
                      "epilogue, no product rows"}
        # the instructions a family issues decide which roof prices it: the tile kernel runs exact-fp32 MFMA (157.3 TFLOP/s
        # dense), the ws / os families bf16 MFMA (6 per fp32 product in the split: 2.5 PFLOP/s / 6 = 417 fp32-equivalent
        # TFLOP/s, never the binding roof at these channel counts) -> they are priced against HBM
        mfma_peak = {"tile": MFMA_F32_PEAK_TFLOPS, "ws": 2500.0 / 6.0, "os": 2500.0 / 6.0}
        legs = {k: leg(d) for k, d in s["by_path"].items()}
        dom = max(s["by_path"], key=lambda k: s["by_path"][k]["ms"])      # dominant kernel family by summed time
        dd, dl = s["by_path"][dom], legs[dom]
        t_hbm = dd["bytes"] / (HBM_PEAK_GBS * 1e9)
        t_mfma = dd["flops"] / (mfma_peak[dom] * 1e12)
        if t_mfma > t_hbm:
            roof = {"bound": "mfma", "achieved": dl["mfma_tflops"], "peak": mfma_peak[dom], "unit": "TFLOP/s",
                    "frac": dl["mfma_tflops"] / mfma_peak[dom]}
        else:
            roof = {"bound": "hbm", "achieved": dl["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": dl["frac"]}
        traffic, traffic_src = pmc_traffic(dom)

        def all_layers(d, n):
            g = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            return {"launches_per_encode": d["launches"] // n, "hbm_gbs": g, "hbm_frac": g / HBM_PEAK_GBS,
                    "mfma_tflops": d["flops"] / (d["ms"] * 1e-3) / 1e12, "avg_launch_us": 1000.0 * d["ms"] / d["launches"],
                    "encode_conv_us": 1000.0 * d["ms"] / n}

        # the same aggregate at other batch sizes (roofline per batch size): one pair and the bench's B pairs
        by_batch = {}
        for nb in sorted({1, B}):
            pb = ops.SpconvProfile()
            ops.PROFILE = pb
            for i in range(3):
                batch = [pairs[(i * nb + j) % len(pairs)] for j in range(nb)]
                pipe.encode_batch(pipe.voxelize_batch([c for p in batch for c in p])[0])
            ops.PROFILE = None
            sb = pb.summary()
            by_batch[f"{nb}_pairs_per_launch"] = dict(all_layers(sb, 3), frames_per_launch=2 * nb)
        out["roofline"] = {
            "kernel": f"{names[dom]} ({what[dom]}); {dl['launches_per_encode']} of the {s['launches'] // nprof} "
                      f"MFMA conv layers of one encode",
            **roof,
            "traffic": traffic, "traffic_source": traffic_src,
            "launches_per_encode": dl["launches_per_encode"],
            "algorithmic_bytes_per_launch": dl["algorithmic_bytes_per_launch"],
            "algorithmic_flops_per_launch": dd["flops"] / dd["launches"],
            "avg_launch_us": dl["avg_launch_us"],
            "hbm_gbs": dl["achieved"], "hbm_frac": dl["frac"],
            "mfma_tflops": dl["mfma_tflops"], "mfma_frac_of_f32_peak": dl["mfma_tflops"] / MFMA_F32_PEAK_TFLOPS,
            "all_conv_layers": all_layers(s, nprof),
            "by_batch": by_batch,
            "by_kernel": {names[k]: v for k, v in sorted(legs.items())},
        }
    if rank == 0 and world == 1 and not args.no_workloads:
        if args.match_lanes > 1:
            ops.set_match_lanes(1)      # the side workloads keep several steps in flight themselves
        out["workloads"] = extra_workloads(dev, log, cpu_baselines=not args.no_cpu_baseline)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("cpu baseline (oracle on host cores) ...")
        out["cpu_baseline"] = cpu_baseline(model.state_dict(), args.model, args.n_out, host_pairs,
                                           args.ransac_iters)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
