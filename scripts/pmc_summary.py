"""Summarise the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_round.sh into HBM bytes per launch of the sparse-conv
kernels (and of the KPConv kernel of the Predator pair).  gfx950 correction: FETCH_SIZE counts 64 B per 128-B request ->
x2; WRITE_SIZE exact; both in KB (MI355X_MICROARCH.md, HBM).

Every encoder call of the profiled bench command runs the FULL-SIZE workload (12 frames per call): since round 2 the
bench primes its streams with the timed workload itself, not with a small pair, so the per-launch averages are per
full-size launch (round 1's figure was diluted 2.8x by 180 small priming launches)."""
import csv, glob, json, os, sys
out = sys.argv[1]
KERNELS = ("k_spconv_pairs", "k_ws_gemm", "k_ws3_gemm", "k_ws_reduce", "k_os_conv", "k_os_build", "k_pairs_build",
           "k_pairs3_build", "k_dense_gemm", "k_dense_rows", "k_spconv_smallcin")


def per_kernel(counter, sub, names):
    files = glob.glob(os.path.join(out, sub.format(counter), "**", "*counter_collection.csv"), recursive=True)
    acc = {k: [0.0, 0] for k in names}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            for k in names:
                if k in r["Kernel_Name"]:
                    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc


def table(sub, names):
    fetch, write = per_kernel("FETCH_SIZE", sub, names), per_kernel("WRITE_SIZE", sub, names)
    res = {}
    for k in names:
        nf, nw = fetch[k][1], write[k][1]
        if not nf or not nw:
            continue
        f_kb, w_kb = fetch[k][0] / nf, write[k][0] / nw
        res[k] = {"launches": nf, "FETCH_SIZE_avg_KB": f_kb, "WRITE_SIZE_avg_KB": w_kb,
                  "hbm_bytes_per_launch": (2 * f_kb + w_kb) * 1024}
    return res


res = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 4 "
                  "--warmup 2 --streams 1 --depth 1 --match-lanes 1 --no-cpu-baseline --no-roofline --no-workloads   (6 pairs = 12 frames per "
                  "encoder call, every call full size)",
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE "
                     "exact; units KB",
       "kernels": table("pmc_{}", KERNELS)}
ks = res["kernels"]
if "k_ws_reduce" in ks and ("k_ws_gemm" in ks or "k_ws3_gemm" in ks):
    # one weight-stationary conv layer = one gemm launch (k_ws_gemm_bf3 over per-offset pairs, or k_ws3_gemm_bf3 over x-triple
    # entries) + one k_ws_reduce launch: bytes of the family / layers (= reduce launches)
    tot = sum(ks[k]["hbm_bytes_per_launch"] * ks[k]["launches"] for k in ("k_ws_gemm", "k_ws3_gemm", "k_ws_reduce") if k in ks)
    res["kernel"] = "k_ws_gemm_bf3 | k_ws3_gemm_bf3 + k_ws_reduce (one weight-stationary conv layer = one gemm + one reduce launch)"
    res["ws_family"] = {"layers": ks["k_ws_reduce"]["launches"], "hbm_bytes_per_layer": tot / ks["k_ws_reduce"]["launches"]}
    res["hbm_bytes_per_launch"] = res["ws_family"]["hbm_bytes_per_layer"]


def whole_step(sub):
    """Fabric traffic of EVERY kernel of the profiled command, per step (12 frames): which kernels move the bytes.  Steps =
    launches of conv1's occupancy kernel (one per encode = one per step)."""
    import re
    tot = {}
    for counter, mul in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        for f in glob.glob(os.path.join(out, sub.format(counter), "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") != counter:
                    continue
                name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
                t = tot.setdefault(name, {"fetch_MB": 0.0, "write_MB": 0.0, "launches": 0})
                t["fetch_MB" if counter == "FETCH_SIZE" else "write_MB"] += mul * float(r["Counter_Value"]) / 1024.0
                if counter == "FETCH_SIZE":
                    t["launches"] += 1
    steps = max(1, sum(v["launches"] for k, v in tot.items() if k.startswith("k_occ_conv")))
    rows = sorted(tot.items(), key=lambda kv: -(kv[1]["fetch_MB"] + kv[1]["write_MB"]))
    return {"steps_profiled": steps, "total_MB_per_step": sum(v["fetch_MB"] + v["write_MB"] for _, v in rows) / steps,
            "kernels": [{"kernel": k, "MB_per_step": (v["fetch_MB"] + v["write_MB"]) / steps, "fetch_MB_per_step": v["fetch_MB"] / steps,
                         "write_MB_per_step": v["write_MB"] / steps, "launches_per_step": v["launches"] / steps} for k, v in rows[:20]]}


res["whole_step"] = whole_step("pmc_{}")
pk = table("pred_pmc_{}", ("k_kpconv_weighted_mfma", "k_kpconv_weighted_generic"))
if pk:
    res["predator_kpconv"] = {"command": "same counters -- python3 scripts/kpconv_bench.py (KPConv step 1 per pyramid "
                                         "level of one full-size pair, 13 launches per shape)", "kernels": pk}
print(json.dumps(res, indent=1))
