#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call D: spatial-modality tests first (new ConvNeXt path), then the whole GPU suite, a fresh GEMM autotune log (for the
# shipped tile table) and the in-situ profile
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_configs_gpu.py -m gpu -q -k spatial ) > gpurun_out/pytest_spatial.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|Error|assert" gpurun_out/pytest_spatial.log | cut -c1-300 | head -20
( timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_gpu.log | cut -c1-250
GL_GEMM_NO_TABLE=1 GL_GEMM_TUNE_LOG=1 GL_GEMM_TUNE_REPS=10 timeout 600 python bench.py --steps 1 --warmup 1 --lanes 1 --no-cpu-baseline > gpurun_out/tune_bench.json 2> gpurun_out/tune_r2.log
grep -c "gemm autotune" gpurun_out/tune_r2.log; cut -c1-160 gpurun_out/tune_bench.json
timeout 300 python tools/insitu.py > gpurun_out/insitu_r2d.txt 2> gpurun_out/insitu_r2d.err
head -1 gpurun_out/insitu_r2d.txt
