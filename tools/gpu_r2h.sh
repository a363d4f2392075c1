#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call H: chunk-major conv K order -- op tests, same-box A/B of the gather order (kbench, timing only), suite, in-situ,
# bench, PMC traffic
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "conv or vae or linear" ) > gpurun_out/pytest_conv.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_conv.log | cut -c1-250
K=gligen_amd/build/kbench
for d in 16 0; do
  echo "== GL_GEMM_DBG=$d  (16 = old tap-major K order, timing only; 0 = chunk-major)"
  GL_GEMM_DBG=$d timeout 200 $K tools/unet_b8.shapes 10 conv | grep "^conv\|^TOTAL conv" | cut -c1-110
  GL_GEMM_DBG=$d timeout 200 $K tools/vae_b4.shapes 5 conv | grep "^conv\|^TOTAL conv" | cut -c1-110
done > gpurun_out/conv_korder_ab.txt 2>&1
grep "TOTAL\|==" gpurun_out/conv_korder_ab.txt
( timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_gpu.log | cut -c1-250
timeout 300 python tools/insitu.py > gpurun_out/insitu_h.txt 2> gpurun_out/insitu_h.err
head -1 gpurun_out/insitu_h.txt
timeout 400 python bench.py --steps 3 --no-cpu-baseline > gpurun_out/bench_h.json 2> gpurun_out/bench_h.err
cut -c1-200 gpurun_out/bench_h.json; tail -2 gpurun_out/bench_h.err | cut -c1-300
bash tools/gpu_traffic.sh > gpurun_out/traffic_h.log 2>&1
tail -3 gpurun_out/traffic_h.log | cut -c1-200
