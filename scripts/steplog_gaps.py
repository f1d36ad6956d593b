"""Largest gaps between step completions and longest steps in a bench.py log written with APR_BENCH_STEPLOG=1 (a stall that is
not the program's shows as one long gap): python scripts/steplog_gaps.py bench.log"""
import re, sys
rows = []
for line in open(sys.argv[1]):
    m = re.search(r"step\s+(\d+) worker (\d) start\s+([\d.]+) ms\s+host\s+([\d.]+) ms", line)
    if m:
        rows.append((float(m.group(3)), float(m.group(4)), int(m.group(1))))
    if "priming steps" in line or "timed loop" in line or "steady state" in line:
        print(line.strip())
rows.sort()
ends = sorted(a + b for a, b, _ in rows)
gaps = [(ends[i + 1] - ends[i], ends[i]) for i in range(len(ends) - 1)]
gaps.sort(reverse=True)
print("largest completion gaps (ms, at ms):", [(round(g, 1), round(t, 1)) for g, t in gaps[:4]])
print("longest steps (host ms, start ms, step):", sorted(((round(b, 1), round(a, 1), i) for a, b, i in rows), reverse=True)[:4])
