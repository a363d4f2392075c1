#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 3, call J: GEGLU projection with the previous item's epilogue under the K loop (gemm_geglu_kernel) against gemm_wide_kernel<4>
# (GL_GEGLU_PIPE=0): op tests, per-shape check + timing, whole path
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3j
mkdir -p $O
K=gligen_amd/build/kbench
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "geglu or ln_folded or linear" ) > $O/pytest_ops.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|Error|^E  " $O/pytest_ops.log | cut -c1-300 | head -20
{
for f in 0 1 0 1; do
  echo "== GL_GEGLU_PIPE=$f"
  GL_GEGLU_PIPE=$f timeout 200 $K tools/unet_b8.shapes 10 gemm check | grep "^gemm [0-9]* [0-9]* [0-9]* 1 \|^TOTAL gemm\|CHECK\|MISMATCH\|mismatch" | cut -c1-150
done
} > $O/geglu_pipe_kbench.txt 2>&1
cat $O/geglu_pipe_kbench.txt
{
for f in 0 1 0 1; do
  echo "== GL_GEGLU_PIPE=$f"
  GL_GEGLU_PIPE=$f timeout 400 python bench.py --steps 4 --no-cpu-baseline 2>$O/bench_$f.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('bench images/s', round(d['value'],4), 'unet_step_ms', round(d['unet_step_ms'],3), 'eager_sum', r['eager_sum_ms'], 'dominant', r['kernel'], round(r['achieved'],1), 'sclk', d['gpu_clocks']['sclk_mhz']['mean']); [print('   ', k) for k in r['kernels'][:3]]"
done
} > $O/geglu_pipe_bench_ab.txt 2>&1
cat $O/geglu_pipe_bench_ab.txt; tail -2 $O/bench_1.err | cut -c1-300
