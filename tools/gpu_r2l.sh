#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call L: whole-path A/B of the conv K order on one box -- bench.py + in-situ with the library built from the same tree
# minus the K-order commit (var_tap: tap-major gather, weights packed to match) against the current one, alternating
export TMPDIR=/tmp
mkdir -p gpurun_out
L=gligen_amd/libgligen_amd.so
cp $L /tmp/new.so
{
for arm in tap new tap new tap new; do
  if [ $arm = tap ]; then cp gligen_amd/build/var_tap/libgligen_amd.so $L; else cp /tmp/new.so $L; fi
  echo "== $arm"
  timeout 300 python tools/insitu.py 2>/dev/null | head -1
  timeout 400 python bench.py --steps 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bench images/s', round(d['value'],4), 'unet_step_ms', round(d['unet_step_ms'],3), 'vae ms', round(d['vae_decode_ms'],2), 'sclk', d['gpu_clocks']['sclk_mhz']['mean'])"
done
cp /tmp/new.so $L
} > gpurun_out/korder_bench_ab.txt 2>&1
cat gpurun_out/korder_bench_ab.txt
