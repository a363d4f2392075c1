#!/bin/bash
# round 4 (f): engine forks (shared packed weights), virtual-memory arena, lanes fixes: probe + tests + bench line
export GL_DEV_SWITCHES=1
O=gpurun_out/r4f; mkdir -p $O
timeout 60 gligen_amd/build/vmmtest > $O/vmmtest.txt 2>&1; cat $O/vmmtest.txt
( timeout 900 python -m pytest tests/test_configs_gpu.py -m gpu -q -x -k "lanes or fork or two_prompts or spatial_sampler or run_entry" ) > $O/pytest_cfg.log 2>&1; tail -15 $O/pytest_cfg.log | cut -c1-220
( timeout 600 python -m pytest tests/test_path_gpu.py -m gpu -q -x ) > $O/pytest_path.log 2>&1; tail -4 $O/pytest_path.log | cut -c1-220
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","value_one_lane","per_gpu_weight_bytes","arena_reserved_bytes","arena_high_water_gb","unet_step_ms","launches_per_unet_eval")})
PY
tail -3 $O/bench.err | cut -c1-300
