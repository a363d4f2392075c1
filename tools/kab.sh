#!/bin/bash
# Same-box A/B of variant libraries (tools/build_variant.sh) on one kbench shapes file, alternated ROUNDS times:
#     gpurun -- 'bash tools/kab.sh SHAPES ROUNDS REPS base var_u1 var_u2 ...'      -> gpurun_out/$GL_OUT/kab_<shapes>.txt
# "base" is gligen_amd/build/kbench (the shipped library); var_X is gligen_amd/build/var_X/kbench. Per shape: the minimum over the rounds.
export TMPDIR=/tmp GL_DEV_SWITCHES=1
R=$PWD; O=$R/gpurun_out/${GL_OUT:-run}; mkdir -p $O
f=$1; rounds=$2; reps=$3; shift 3
b=$(basename $f .shapes)
for r in $(seq $rounds); do for v in "$@"; do
  k=gligen_amd/build/$v/kbench; [ $v = base ] && k=gligen_amd/build/kbench
  timeout 300 $k $f $reps - check > $O/kab_${b}_${v}_$r.txt 2>&1
done; done
python - "$O" "$b" "$rounds" "$@" <<'PY' | tee $O/kab_$b.txt
import sys, re
O, b, rounds, vs = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4:]
tab, order, chk = {}, [], {}
for v in vs:
    for r in range(1, rounds + 1):
        for line in open(f"{O}/kab_{b}_{v}_{r}.txt"):
            m = re.match(r"^((?:gemm|conv|attn|gn|ln)(?: \d+)+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s+(\S+)", line)
            if m:
                key, us, cnt = m.group(1).strip(), float(m.group(2)), int(m.group(5))
                if key not in order: order.append(key)
                d = tab.setdefault(key, {})
                d[v] = min(d.get(v, 1e30), us); d["cnt"] = cnt; d.setdefault("cfg_" + v, m.group(6))
            if line.startswith("CHECK") or "MISMATCH" in line: chk[v] = line.strip()[:100]
print(f"{'shape':34s} {'cnt':>4s} " + " ".join(f"{v:>10s}" for v in vs))
tot = {v: 0.0 for v in vs}
for k in order:
    d = tab[k]
    print(f"{k:34s} {d['cnt']:4d} " + " ".join(f"{d.get(v, float('nan')):10.1f}" for v in vs) + "   " + d.get("cfg_" + vs[0], ""))
    for v in vs: tot[v] += d.get(v, 0) * d["cnt"] / 1e3
print(f"{'TOTAL ms (count-weighted)':39s} " + " ".join(f"{tot[v]:10.3f}" for v in vs))
for v in vs: print(v, chk.get(v, "no check line"))
PY
