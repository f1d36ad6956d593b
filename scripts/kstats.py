"""Print calls / average / min duration of the kernels whose name contains PATTERN from rocprofv3 kernel_stats.csv files.
usage: python scripts/kstats.py PATTERN file.csv [file.csv ...]"""
import csv, sys
pat = sys.argv[1]
for f in sys.argv[2:]:
    try:
        rows = list(csv.DictReader(open(f)))
    except OSError as e:
        print(f, "unreadable:", e); continue
    for r in rows:
        if pat in r["Name"]:
            name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            print(f"{f.split('/')[-2]:18s} {name:24s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f} us")
