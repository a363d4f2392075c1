# whole batches in flight against the number of hardware queues the HIP runtime may use (GPU_MAX_HW_QUEUES, default 4), one box
export TMPDIR=/tmp
O=gpurun_out/${GL_OUT:-r6_lanes}; mkdir -p $O; : > $O/hwq_sweep.txt
run() { q=$1; l=$2; s=$3
  ( [ "$q" != def ] && export GPU_MAX_HW_QUEUES=$q; timeout 600 python bench.py --lanes $l --steps $s --warmup 2 --no-cpu-baseline --no-train-step --no-ff-ab 2> $O/q.err ) > $O/q.json
  python -c "import json; d=json.loads(open('$O/q.json').read().strip().splitlines()[-1]); print('hwq $q lanes $l steps $s value %.3f one_lane %.3f ms_per_step %.1f' % (d['value'], d['value_one_lane'], d['ms_per_step']))" | tee -a $O/hwq_sweep.txt; }
for r in 1 2; do
run def 3 12; run 8 3 12; run 8 4 12; run 8 6 12; run 2 3 12; run def 4 12
done
