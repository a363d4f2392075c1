#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 3, call U: head-layout epilogues without integer divides (host-prepared multipliers for / T and / d, sample / token hoisted
# out of the column loop) against the previous commit's library; then the full GPU suite and the driver's bench command
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3u
mkdir -p $O
B=gligen_amd/build
{
for round in 1 2; do
for arm in main old; do
  k=$B/kbench; [ $arm != main ] && k=$B/var_$arm/kbench
  echo "== $arm, round $round"
  timeout 200 $k tools/unet_b8.shapes 10 gemm check | grep "^gemm [0-9]* [0-9]* [0-9]* 4 \|^TOTAL gemm\|CHECK\|MISMATCH\|mismatch" | cut -c1-110
done
done
} > $O/qkv_nodiv_kbench.txt 2>&1
cat $O/qkv_nodiv_kbench.txt
( timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "^FAILED|^ERROR|passed|failed|pytest rc" $O/pytest_gpu.log | cut -c1-250 | tee $O/pytest_gpu_summary.txt
( unset GL_DEV_SWITCHES; timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>/dev/null ); cut -c1-200 $O/bench_driver_cmd.json
