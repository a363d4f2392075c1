#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 3, call F: op test of the folded LayerNorm, full GPU suite with the re-tuned tile table, bench
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3f
mkdir -p $O
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "ln_folded or linear or geglu" ) > $O/pytest_lnop.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|Error|^E  " $O/pytest_lnop.log | cut -c1-300 | head -20
( timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "^FAILED|^ERROR|passed|failed|pytest rc" $O/pytest_gpu.log | cut -c1-300
timeout 600 python bench.py --steps 6 --warmup 2 > $O/bench.json 2> $O/bench.err
python -c "import sys,json; d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1]); r=d['roofline']; print('bench images/s', round(d['value'],4), 'unet_step_ms', round(d['unet_step_ms'],3), 'eager_sum', r['eager_sum_ms'], 'launches', d.get('launches_per_unet_eval'), 'frac', round(r['frac'],4), 'whole', round(r['whole_image']['frac'],4)); [print('   ', k) for k in r['kernels'][:14]]"
tail -2 $O/bench.err | cut -c1-300
