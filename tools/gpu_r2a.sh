#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call A: GPU tests (all, no -x), bench on every BASELINE config, lanes sweep, in-situ per-problem profile
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 600 python bench.py > gpurun_out/bench_C2.json 2> gpurun_out/bench_C2.err
cut -c1-600 gpurun_out/bench_C2.json; tail -2 gpurun_out/bench_C2.err
for c in C3 C4 C5; do
  timeout 400 python bench.py --config $c --steps 2 --no-cpu-baseline > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
  cut -c1-330 gpurun_out/bench_$c.json; tail -2 gpurun_out/bench_$c.err | cut -c1-300
done
for l in 1 3; do
  timeout 400 python bench.py --lanes $l --steps 3 --no-cpu-baseline > gpurun_out/bench_l$l.json 2> gpurun_out/bench_l$l.err
  cut -c1-200 gpurun_out/bench_l$l.json
done
timeout 300 python tools/insitu.py > gpurun_out/insitu.txt 2> gpurun_out/insitu.err
head -3 gpurun_out/insitu.txt
