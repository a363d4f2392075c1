#!/bin/bash
# round 4 (h): chained row-local launches in the engine: op tests, full-size parity, whole-path A/B (GL_FF_CHAIN = 0 / 1 / 2)
export GL_DEV_SWITCHES=1
O=gpurun_out/r4h; mkdir -p $O
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "ff_chain or feedforward" ) > $O/pytest_ops.log 2>&1; tail -3 $O/pytest_ops.log | cut -c1-200
( timeout 900 python -m pytest tests/test_configs_gpu.py -m gpu -q -k "full_size_pair or c1_end" ) > $O/pytest_cfg.log 2>&1; tail -3 $O/pytest_cfg.log | cut -c1-200
grep -E "^FAILED|^ERROR" $O/pytest_ops.log $O/pytest_cfg.log | cut -c1-250
for r in 1 2; do
  for ch in 0 1 2; do
    GL_FF_CHAIN=$ch timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_ch$ch.$r.json 2> $O/bench_ch$ch.$r.err
  done
  python - <<PY
import json
for n in ("ch0","ch1","ch2"):
    try:
        d=json.loads(open("$O/bench_%s.$r.json"%n).read().strip().splitlines()[-1])
        print(n, "img/s %.3f"%d["value"], "one lane %.3f"%d["value_one_lane"], "unet_step_ms %.3f"%d.get("unet_step_ms"), "launches", d.get("launches_per_unet_eval"), "eager_sum", d["roofline"].get("eager_sum_ms"))
    except Exception as e: print(n, "failed", e)
PY
done
