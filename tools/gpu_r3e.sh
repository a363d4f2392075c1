#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 3, calls E and Q (Q: again at the end of the round, after the epilogue changes): fresh on-device autotune of every BASELINE configuration with the 64 x 64 tile candidate (4 workgroups per CU)
# and the folded-LayerNorm problem keys -> logs for tools/make_tuned_table.py; A/B shipped table vs fresh tuning
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3q
mkdir -p $O
for c in C2 C3 C4 C5; do
  GL_GEMM_NO_TABLE=1 GL_GEMM_TUNE_LOG=1 GL_GEMM_TUNE_REPS=10 timeout 600 python bench.py --steps 1 --warmup 1 --lanes 1 --no-cpu-baseline --config $c > $O/tune_$c.json 2> $O/tune_$c.log
  echo "$c: $(grep -c 'gemm autotune' $O/tune_$c.log) problems tuned, $(grep 'gemm autotune' $O/tune_$c.log | grep -c 'cfg 4 ') won by the 64x64 tile"
done
grep "gemm autotune" $O/tune_C2.log | grep "cfg 4 " | cut -c1-200 | head -40
{
for arm in table fresh table fresh; do
  echo "== $arm"
  if [ $arm = fresh ]; then export GL_GEMM_NO_TABLE=1; else unset GL_GEMM_NO_TABLE; fi
  timeout 400 python bench.py --steps 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('bench images/s', round(d['value'],4), 'unet_step_ms', round(d['unet_step_ms'],3), 'eager_sum', r['eager_sum_ms'], 'launches', d.get('launches_per_unet_eval'), 'sclk', d['gpu_clocks']['sclk_mhz']['mean'])"
done
} > $O/table_vs_fresh.txt 2>&1
cat $O/table_vs_fresh.txt
