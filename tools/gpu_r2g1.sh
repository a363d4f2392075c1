#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call G1: full GPU suite + bench lines on every BASELINE config (+ the reference demo's alpha schedule)
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_gpu.log | cut -c1-250
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
cut -c1-250 gpurun_out/bench.json; tail -2 gpurun_out/bench.err | cut -c1-200
for c in C3 C4 C5; do
  timeout 400 python bench.py --config $c --steps 2 --no-cpu-baseline > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
  cut -c1-120 gpurun_out/bench_$c.json; tail -1 gpurun_out/bench_$c.err | cut -c1-200
done
timeout 400 python bench.py --alpha-type 0.3,0,0.7 --steps 2 --no-cpu-baseline > gpurun_out/bench_alpha.json 2> gpurun_out/bench_alpha.err
cut -c1-120 gpurun_out/bench_alpha.json; tail -1 gpurun_out/bench_alpha.err | cut -c1-200
timeout 400 python bench.py --lanes 1 --steps 2 --no-cpu-baseline > gpurun_out/bench_l1.json 2> gpurun_out/bench_l1.err
cut -c1-120 gpurun_out/bench_l1.json
