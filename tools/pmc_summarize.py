#!/usr/bin/env python
"""Fold rocprofv3 --pmc counter_collection CSVs into per-kernel-symbol averages, and (with --launch-log) per-PROBLEM rows.

usage: pmc_summarize.py OUT.csv DIR [DIR ...] [--launch-log LOG --per-problem OUT2.csv]
  each DIR = one --pmc pass (any counters). Output rows: kernel, counter, dispatches, mean, sum.
  LOG = the engine's GL_LAUNCH_LOG of the same command ("symbol|M|N|K|mode|algorithmic bytes" per GEMM / conv launch, in launch
  order): the i-th dispatch of a gemm_u_kernel symbol in a pass is the i-th log line with that symbol, which gives every counter
  row its problem. OUT2 rows: kernel, M, N, K, mode, launches, algorithmic_bytes, <counter> mean per launch ...
Template arguments are kept, parameter lists dropped."""
import csv, glob, os, re, sys
from collections import defaultdict

args = sys.argv[1:]
log = per_problem = None
if "--launch-log" in args:
    i = args.index("--launch-log"); log = args[i + 1]; del args[i:i + 2]
if "--per-problem" in args:
    i = args.index("--per-problem"); per_problem = args[i + 1]; del args[i:i + 2]
out, dirs = args[0], args[1:]


def clean(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name).replace("gl::", "")


acc = defaultdict(lambda: [0, 0.0])
rows_by_pass = []
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = []
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                name = clean(r.get("Kernel_Name") or r.get("Kernel Name") or "")
                a = acc[(name, r["Counter_Name"])]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
                rows.append((int(r.get("Dispatch_Id") or r.get("Dispatch Id") or 0), name, r["Counter_Name"], float(r["Counter_Value"])))
        rows_by_pass.append(rows)
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "counter", "dispatches", "mean", "sum"])
    for (k, c), (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, c, n, f"{s / n:.3f}", f"{s:.1f}"])
print(f"{out}: {len(acc)} rows")

if log and per_problem:
    launches = defaultdict(list)   # symbol -> [(M, N, K, mode, bytes)] in launch order
    for line in open(log):
        sym, M, N, K, mode, b = line.rstrip("\n").split("|")
        launches[sym.split(" + ")[0]].append((int(M), int(N), int(K), int(mode), float(b)))
    prob = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    alg = {}
    for rows in rows_by_pass:
        per_sym = defaultdict(lambda: defaultdict(list))   # symbol -> counter -> [(dispatch id, value)]
        for did, name, ctr, val in rows:
            if name in launches:
                per_sym[name][ctr].append((did, val))
        for name, ctrs in per_sym.items():
            for ctr, vals in ctrs.items():
                vals.sort()
                # a pass may contain more dispatches than the log (warm-up evaluations repeat the same launch sequence): cycle
                seq = launches[name]
                for i, (_, val) in enumerate(vals):
                    M, N, K, mode, b = seq[i % len(seq)]
                    a = prob[(name, M, N, K, mode)][ctr]
                    a[0] += 1
                    a[1] += val
                    alg[(name, M, N, K, mode)] = b
    ctrs = sorted({c for v in prob.values() for c in v})
    with open(per_problem, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "M", "N", "K", "mode(0 rows,1 conv3x3,2 attention: M=Nq N=Nk K=d)", "dispatches", "algorithmic_bytes"] + [f"{c}_mean" for c in ctrs])
        for key, v in sorted(prob.items(), key=lambda kv: -alg[kv[0]] * max(x[0] for x in kv[1].values())):
            n = max(x[0] for x in v.values())
            w.writerow(list(key) + [n, f"{alg[key]:.0f}"] + [f"{v[c][1] / v[c][0]:.3f}" if c in v else "" for c in ctrs])
    print(f"{per_problem}: {len(prob)} problems")
