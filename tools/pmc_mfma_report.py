#!/usr/bin/env python
"""Per-kernel MFMA-pipe and LDS figures from the per-symbol counter means of tools/gpu_mfma_util.sh (pmc_mfma.csv).

  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE)   (both counters are sums over the 8 XCDs: GRBM_GUI_ACTIVE is 8 x the
               launch's cycles -- 1.59 M for a 76 us launch at 2.3 GHz -- and each XCD has 32 CUs x 4 SIMDs of MFMA pipes)
  lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
usage: pmc_mfma_report.py pmc_mfma.csv"""
import csv, sys
from collections import defaultdict

d = defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["kernel"]][r["counter"]] = (float(r["mean"]), int(r["dispatches"]), float(r["sum"]))
rows = []
for k, c in d.items():
    if "GRBM_GUI_ACTIVE" not in c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
        continue
    act = c["GRBM_GUI_ACTIVE"][0]
    mf = c["SQ_VALU_MFMA_BUSY_CYCLES"][0]
    busy = mf / (128.0 * act) if act else 0.0
    conf = c.get("SQ_LDS_BANK_CONFLICT", (0, 0, 0))[0]
    idx = c.get("SQ_LDS_IDX_ACTIVE", (0, 0, 0))[0]
    rows.append((c["GRBM_GUI_ACTIVE"][2], k, c["GRBM_GUI_ACTIVE"][1], act, busy, (conf / idx) if idx else 0.0, c.get("SQ_INSTS_MFMA", (0, 0, 0))[0]))
print(f"{'kernel':58s} {'launches':>8s} {'8 x cycles':>14s} {'MFMA pipe busy':>15s} {'LDS conflict share':>19s} {'MFMA instr/launch':>18s}")
for tot, k, n, act, busy, conf, ni in sorted(rows, reverse=True)[:24]:
    print(f"{k[:58]:58s} {n:8d} {act:14.0f} {busy:15.3f} {conf:19.3f} {ni:18.0f}")
