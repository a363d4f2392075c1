#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call G2: profile evidence -- per-shape kbench tables, in-situ per-problem table, rocprofv3 kernel stats of the bench
# command, FETCH / WRITE PMC passes folded per symbol and per problem
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
K=gligen_amd/build/kbench
timeout 300 $K tools/unet_b8.shapes 10 - check > gpurun_out/kbench_unet.txt 2>&1
grep "^TOTAL\|CHECK\|MISMATCH" gpurun_out/kbench_unet.txt | cut -c1-160
timeout 300 $K tools/vae_b4.shapes 5 - check > gpurun_out/kbench_vae.txt 2>&1
tail -3 gpurun_out/kbench_vae.txt
timeout 300 python tools/insitu.py > gpurun_out/insitu.txt 2> gpurun_out/insitu.err
head -1 gpurun_out/insitu.txt
rm -rf gpurun_out/prof
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python $R/bench.py --no-cpu-baseline ) > gpurun_out/prof.log 2>&1
tail -2 gpurun_out/prof.log | cut -c1-300
find gpurun_out/prof -name "*kernel_trace*" -delete
find gpurun_out/prof -name "*kernel_stats*"
bash tools/gpu_traffic.sh > gpurun_out/traffic.log 2>&1
tail -26 gpurun_out/traffic.log | cut -c1-220
