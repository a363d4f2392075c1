#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# Round-end evidence: GPU tests, bench line, rocprofv3 kernel stats of the bench command, per-shape kbench tables
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
K=gligen_amd/build/kbench
timeout 300 $K tools/unet_b8.shapes 10 - check > gpurun_out/kbench_unet.txt 2>&1
tail -8 gpurun_out/kbench_unet.txt
timeout 300 $K tools/vae_b4.shapes 5 - check > gpurun_out/kbench_vae.txt 2>&1
tail -3 gpurun_out/kbench_vae.txt
( timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
tail -2 gpurun_out/bench.err
rm -rf gpurun_out/prof
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python $R/bench.py --no-cpu-baseline ) > gpurun_out/prof.log 2>&1
tail -2 gpurun_out/prof.log
find gpurun_out/prof -name "*kernel_trace*" -delete
find gpurun_out/prof -name "*kernel_stats*"
bash tools/gpu_traffic.sh > gpurun_out/traffic.log 2>&1
tail -3 gpurun_out/traffic.log
