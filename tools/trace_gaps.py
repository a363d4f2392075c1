"""Timeline reading of a rocprofv3 --kernel-trace CSV of `bench.py --lanes 1`: inside the hipGraph replays of the UNet evaluation, how much of
the wall time is kernels and how much is the space between them (dependent-kernel boundaries), per kernel symbol:
    python tools/trace_gaps.py <kernel_trace.csv>  > profiles/<round>/trace_gaps_one_lane.txt
A 'run' is a maximal sequence of dispatches whose gaps stay below 50 us (one sampling pass of 51 evaluations + decode)."""
import csv
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
runs, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - a[1] > 50_000:
        runs.append(cur)
        cur = []
    cur.append(b)
runs.append(cur)
runs = [r for r in runs if len(r) > 5000]
print(f"{len(rows)} dispatches, {len(runs)} long runs (>5000 dispatches each)")
run = runs[-1]
t0, t1 = run[0][0], run[-1][1]
busy = sum(e - s for s, e, _ in run)
gaps = [max(0, b[0] - a[1]) for a, b in zip(run, run[1:])]
overlap = sum(max(0, a[1] - b[0]) for a, b in zip(run, run[1:]))
print(f"last run: {len(run)} dispatches, wall {(t1 - t0) / 1e6:.3f} ms, sum of kernel durations {busy / 1e6:.3f} ms, sum of gaps {sum(gaps) / 1e6:.3f} ms "
      f"({sum(gaps) / (t1 - t0) * 100:.1f} % of the wall), overlap {overlap / 1e6:.3f} ms")
print(f"per dispatch: mean duration {busy / len(run) / 1e3:.2f} us, mean gap {sum(gaps) / len(gaps) / 1e3:.2f} us, median gap {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us")
short = lambda n: re.sub(r"\(.*", "", n).replace("void gl::", "").replace("(anonymous namespace)::", "")[:70]
by = defaultdict(lambda: [0, 0, 0])   # calls, duration, gap BEHIND this kernel (to the next one)
for (s, e, n), g in zip(run, gaps + [0]):
    d = by[short(n)]
    d[0] += 1; d[1] += e - s; d[2] += g
print(f"{'kernel':72s} {'calls':>7s} {'us/call':>9s} {'gap after':>10s} {'total ms':>9s} {'gap ms':>8s}")
for k, (c, d, g) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:72s} {c:7d} {d / c / 1e3:9.2f} {g / c / 1e3:10.2f} {d / 1e6:9.3f} {g / 1e6:8.3f}")
