#!/bin/bash
# One GPU-box pass: per-shape kernel bench, GPU parity tests, bench line, rocprofv3 kernel stats.
set -x
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
K=gligen_amd/build/kbench
timeout 300 $K tools/unet_b8.shapes 10 > gpurun_out/kbench_unet.txt 2>&1
tail -12 gpurun_out/kbench_unet.txt
timeout 200 $K tools/vae_b4.shapes 5 > gpurun_out/kbench_vae.txt 2>&1
tail -5 gpurun_out/kbench_vae.txt
( timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
tail -3 gpurun_out/bench.err
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline ) > gpurun_out/prof.log 2>&1
tail -3 gpurun_out/prof.log
find gpurun_out/prof -name "*kernel_stats*" | head
# keep only the small summaries (traces can be huge)
find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete
du -sh gpurun_out
