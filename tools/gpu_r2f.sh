#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call F: fresh on-device autotune with the deep-ring candidates (kbench on the UNet shapes, tolerance-checked against the
# v2 kernel) + the op-level GPU tests; the log feeds tools/make_tuned_table.py
export TMPDIR=/tmp
mkdir -p gpurun_out
K=gligen_amd/build/kbench
GL_GEMM_NO_TABLE=1 GL_GEMM_TUNE_LOG=1 GL_GEMM_TUNE_REPS=10 timeout 900 $K tools/unet_b8.shapes 10 - check > gpurun_out/kb_r2f.txt 2> gpurun_out/tune_r2f_kb.log
grep "^TOTAL\|CHECK\|MISMATCH" gpurun_out/kb_r2f.txt | cut -c1-160
grep -c "gemm autotune" gpurun_out/tune_r2f_kb.log; grep -c "cfg [4-7] " gpurun_out/tune_r2f_kb.log; grep "cfg [4-7] " gpurun_out/tune_r2f_kb.log | cut -c1-200 | head -40
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q ) > gpurun_out/pytest_ops.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_ops.log | cut -c1-250
GL_GEMM_NO_TABLE=1 GL_GEMM_TUNE_LOG=1 GL_GEMM_TUNE_REPS=10 timeout 600 python bench.py --steps 2 --warmup 1 --lanes 1 --no-cpu-baseline > gpurun_out/tune_bench2.json 2> gpurun_out/tune_r2f.log
grep -c "gemm autotune" gpurun_out/tune_r2f.log; grep -c "cfg [4-7] " gpurun_out/tune_r2f.log; cut -c1-160 gpurun_out/tune_bench2.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/tune_bench2.json')); print(d['value'], d['unet_step_ms'], d['roofline']['eager_sum_ms'], d['gpu_clocks'])
PY
