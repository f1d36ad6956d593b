"""GPU idle time in a rocprofv3 --kernel-trace csv: union of kernel intervals against the span, and the idle gaps grouped by
the kernel that ends before the gap (which kernel the GPU waits AFTER).  python scripts/trace_gaps.py <kernel_trace.csv> [skip_fraction]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")) for r in rows)
ev = ev[int(len(ev) * skip):]          # drop the warm-up
span = ev[-1][1] - ev[0][0]
busy, cur_end, gaps = 0, ev[0][0], collections.defaultdict(lambda: [0, 0])
last_name = None
for s, e, n in ev:
    if s > cur_end:
        if last_name is not None:
            g = gaps[(last_name, n)]; g[0] += 1; g[1] += s - cur_end
        busy += e - s; cur_end, last_name = e, n
    else:
        if e > cur_end:
            busy += e - cur_end; cur_end, last_name = e, n
print(f"span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), kernels {len(ev)}, sum of durations {sum(e - s for s, e, _ in ev) / 1e6:.2f} ms")
tot_gap = span - busy
print(f"idle {tot_gap / 1e6:.2f} ms; largest gap classes (after kernel -> before kernel: count, total us, mean us):")
for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {a[:38]:38s} -> {b[:38]:38s} {c:6d} {t / 1e3:9.1f} {t / c / 1e3:7.1f}")
