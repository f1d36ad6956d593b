import sys, collections
rows = [list(map(int, l.split())) for l in open(sys.argv[1])]
t0 = min(r[1] for r in rows); t1 = max(r[2] for r in rows)
print("WGs", len(rows), "span", (t1 - t0) / 100.0, "us (100 MHz clock assumed)", "units", rows[0][4])
st = sorted((r[1] - t0) / 100.0 for r in rows); en = sorted((r[2] - t0) / 100.0 for r in rows)
dur = sorted((r[2] - r[1]) / 100.0 for r in rows)
q = lambda a, f: a[min(len(a) - 1, int(f * len(a)))]
print("start  p0/50/90/100:", q(st, 0), q(st, .5), q(st, .9), st[-1])
print("end    p0/10/50/90/100:", en[0], q(en, .1), q(en, .5), q(en, .9), en[-1])
print("dur    p0/10/50/90/100:", dur[0], q(dur, .1), q(dur, .5), q(dur, .9), dur[-1])
percu = collections.Counter(((r[3] >> 32) & 0xf, (r[3] >> 8) & 0xf, (r[3] >> 13) & 0x7) for r in rows)   # xcc, cu_id, se_id
print("distinct (xcc, cu, se):", len(percu), "WGs per CU histogram:", sorted(collections.Counter(percu.values()).items()))
