"""Copy the judged artefacts of a scripts/profile_round.sh run from gpurun_out/profile_<tag>/ into profiles/ as r01_*.
usage: python scripts/collect_profiles.py <tag> [prefix=r01]"""
import csv, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else "r05"
src = os.path.join(root, "gpurun_out", f"profile_{tag}")
dst = os.path.join(root, "profiles")
pairs = [("bench.json", f"{prefix}_bench.json"), ("bench_s1.json", f"{prefix}_bench_streams1.json"),
         ("stats/b_kernel_stats.csv", f"{prefix}_bench_kernel_stats.csv"),
         ("stats1/b_kernel_stats.csv", f"{prefix}_bench_streams1_kernel_stats.csv"),
         ("pmc_spconv_summary.json", f"{prefix}_pmc_spconv_summary.json"),
         ("pred_stats/p_kernel_stats.csv", f"{prefix}_predator_kernel_stats.csv"),
         ("predator_profile.log", f"{prefix}_predator_host_profile.log")]
pairs += [("bench_s1_plain.json", f"{prefix}_bench_streams1_plain.json"),
          ("bench_s1_d3_l3.json", f"{prefix}_bench_streams1_depth3_lanes3.json")]
pairs += [("selflaunch_gloo2.json", f"{prefix}_selflaunch_gloo2.json"),
          ("selflaunch_gloo2_config4.json", f"{prefix}_selflaunch_gloo2_config4.json"),
          ("host_cpu_split_poll.log", f"{prefix}_host_cpu_split_poll.log"),
          ("host_cpu_split_sync.log", f"{prefix}_host_cpu_split_sync.log"),
          ("match_stats/m_kernel_stats.csv", f"{prefix}_matching_0_30pct_kernel_stats.csv"),
          ("match_load.log", f"{prefix}_matching_0_30pct.log")]
pairs += [("one_pair_split.log", f"{prefix}_one_pair_split.log"),
          ("one_pair_split_python_plan.log", f"{prefix}_one_pair_split_python_plan.log")]
pairs += [("train_step.json", f"{prefix}_train_step.json"), ("train_step.log", f"{prefix}_train_step.log"),
          ("train_stats/t_kernel_stats.csv", f"{prefix}_train_step_kernel_stats.csv")]
pairs += [(f"driver_cmd_{i}.json", f"{prefix}_driver_cmd_{i}.json") for i in (1, 2, 3)]
pairs += [(f"driver_cmd_{i}.log", f"{prefix}_driver_cmd_{i}.log") for i in (1, 2, 3)]
for a, b in pairs:
    shutil.copyfile(os.path.join(src, a), os.path.join(dst, b))
    print("copied", b)
cols = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Counter_Name",
        "Counter_Value"]
for c, name in (("FETCH_SIZE", "fetch_size"), ("WRITE_SIZE", "write_size")):
    rows = list(csv.DictReader(open(os.path.join(src, f"pmc_{c}", "p_counter_collection.csv"))))
    keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_spconv", "k_ws_", "k_os_", "k_dense_rows"))]   # the sparse-conv kernels
    prows = list(csv.DictReader(open(os.path.join(src, f"pred_pmc_{c}", "p_counter_collection.csv"))))
    keep += [r for r in prows if "k_kpconv" in r["Kernel_Name"]]
    with open(os.path.join(dst, f"{prefix}_pmc_{name}_spconv.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(cols)
        for r in keep:
            w.writerow([r[k][:60] if k == "Kernel_Name" else r[k] for k in cols])
    print("wrote", f"{prefix}_pmc_{name}_spconv.csv", len(keep), "rows")
