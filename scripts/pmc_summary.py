"""Summarise the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_round.sh into HBM bytes per launch of the sparse-conv
kernels (gfx950 correction: FETCH_SIZE counts 64 B per 128-B request -> x2; WRITE_SIZE exact; both in KB)."""
import csv, glob, json, os, sys
out = sys.argv[1]
KERNELS = ("k_spconv_pairs", "k_ws_gemm", "k_ws_reduce")


def per_kernel(counter):
    files = glob.glob(os.path.join(out, f"pmc_{counter}", "**", "*counter_collection.csv"), recursive=True)
    acc = {k: [0.0, 0] for k in KERNELS}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            for k in KERNELS:
                if k in r["Kernel_Name"]:
                    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc


fetch, write = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
res = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 4 "
                  "--warmup 2 --streams 1 --no-cpu-baseline --no-roofline   (6 pairs per encoder call, as in the timed "
                  "run; averages include the priming calls on a small pair)",
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE "
                     "exact; units KB",
       "kernels": {}}
for k in KERNELS:
    nf, nw = fetch[k][1], write[k][1]
    if not nf or not nw:
        continue
    f_kb, w_kb = fetch[k][0] / nf, write[k][0] / nw
    res["kernels"][k] = {"launches": nf, "FETCH_SIZE_avg_KB": f_kb, "WRITE_SIZE_avg_KB": w_kb,
                         "hbm_bytes_per_launch": (2 * f_kb + w_kb) * 1024}
ks = res["kernels"]
if "k_ws_gemm" in ks and "k_ws_reduce" in ks:
    res["kernel"] = "k_ws_gemm+k_ws_reduce (one weight-stationary conv layer = one launch of each)"
    res["hbm_bytes_per_launch"] = ks["k_ws_gemm"]["hbm_bytes_per_launch"] + ks["k_ws_reduce"]["hbm_bytes_per_launch"]
print(json.dumps(res, indent=1))
