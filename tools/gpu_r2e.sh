#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call E: spatial tests (ConvNeXt fix), staggered-start sweep on the multi-item GEMMs, in-situ + bench with the new tile table
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_configs_gpu.py -m gpu -q -k spatial ) > gpurun_out/pytest_spatial.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|^E  " gpurun_out/pytest_spatial.log | cut -c1-400 | head -12
K=gligen_amd/build/kbench
cat > /tmp/st.shapes <<EOS
gemm 32768 2560 320 1 10
gemm 8192 5120 640 1 10
gemm 2048 10240 1280 1 10
gemm 32768 960 320 4 5
gemm 8192 1920 640 4 5
gemm 32768 320 1280 0 10
conv 8 64 64 320 0 320 1 0 7
EOS
for s in 0 4 8 12 16 24; do
  echo "== stagger $s"
  GL_GEMM_STAGGER=$s timeout 120 $K /tmp/st.shapes 20 | grep "^conv\|^gemm" | cut -c1-110
done > gpurun_out/stagger.txt 2>&1
cat gpurun_out/stagger.txt
timeout 300 python tools/insitu.py > gpurun_out/insitu_r2e.txt 2> gpurun_out/insitu_r2e.err
head -1 gpurun_out/insitu_r2e.txt
timeout 400 python bench.py --steps 3 --no-cpu-baseline > gpurun_out/bench_r2e.json 2> gpurun_out/bench_r2e.err
cut -c1-200 gpurun_out/bench_r2e.json; tail -2 gpurun_out/bench_r2e.err | cut -c1-300
