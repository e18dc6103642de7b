"""Launch-gap analysis of one rocprofv3 kernel trace (needs an MI355X to produce the trace):
   cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -- python bench.py --no-extras --steps 3 --warmup 1
   python tools/gap_analysis.py <dir>
Prints, for the busiest stream window, kernel-busy time, idle time between consecutive kernels and a histogram of the gaps - what a
hipGraph capture of the denoise loop could win at most."""
import csv, glob, sys
import numpy as np

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
s = np.array([int(r["Start_Timestamp"]) for r in rows]); e = np.array([int(r["End_Timestamp"]) for r in rows])
o = np.argsort(s); s, e = s[o], e[o]
names = [rows[i]["Kernel_Name"] for i in o]
gap = s[1:] - np.maximum.accumulate(e)[:-1]
# the timed region = the last 60 % of the trace (warm-up + weight binding come first)
lo = int(len(s) * 0.4)
g = gap[lo:]
busy = (e[lo + 1:] - s[lo + 1:]).sum() / 1e6
wall = (e[-1] - s[lo + 1]) / 1e6
print(f"{len(s) - lo - 1} kernels, wall {wall:.1f} ms, busy {busy:.1f} ms, idle {np.clip(g, 0, None).sum() / 1e6:.1f} ms, overlap {-np.clip(g, None, 0).sum() / 1e6:.1f} ms")
for a, b in ((-1e12, 0), (0, 1000), (1000, 2000), (2000, 4000), (4000, 10000), (10000, 1e5), (1e5, 1e12)):
    m = (g >= a) & (g < b)
    print(f"gap [{a / 1e3:>8.0f}, {b / 1e3:>8.0f}) us: {m.sum():7d} launches  {g[m].sum() / 1e6:9.2f} ms")
big = np.argsort(g)[-8:]
for i in big:
    print(f"  {g[i] / 1e3:8.1f} us before {names[lo + 1 + i][:70]}")

import collections, re
by = collections.defaultdict(lambda: [0, 0.0, 0.0])
for i, gg in enumerate(g):
    if gg < 4000: continue
    nm = re.sub(r"\(.*", "", names[lo + 1 + i])[:60]
    prev = re.sub(r"\(.*", "", names[lo + i])[:40]
    k = prev + "  ->  " + nm
    by[k][0] += 1; by[k][1] += gg / 1e6; by[k][2] += (e[lo + i] - s[lo + i]) / 1e3
print("gaps >= 4 us by (previous kernel -> kernel): count, idle ms, mean duration of the previous kernel (us)")
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{v[0]:6d} {v[1]:8.2f} ms  prev {v[2] / v[0]:7.1f} us  {k}")

gg4 = g[g >= 4000] / 1e3
print("gaps >= 4 us percentiles (us):", {q: round(float(np.percentile(gg4, q)), 2) for q in (1, 10, 25, 50, 75, 90, 99)})
print("launches in window:", len(g), " distinct kernels:", len(set(names[lo:])))

pairs = collections.defaultdict(lambda: [0, 0])
for i, gg in enumerate(g):
    nm = re.sub(r"\(.*", "", names[lo + 1 + i])[:44]
    prev = re.sub(r"\(.*", "", names[lo + i])[:44]
    pairs[prev + "  ->  " + nm][0 if gg >= 4000 else 1] += 1
print("transition: launches with a >= 4 us gap / without")
for k, v in sorted(pairs.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:70]:
    print(f"{v[0]:6d} / {v[1]:6d}  {k}")
