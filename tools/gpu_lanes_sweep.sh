# bench.py at 1 .. 4 whole batches in flight, alternating on one box.  Usage: gpurun -- 'bash tools/gpu_lanes_sweep.sh'
export TMPDIR=/tmp
O=gpurun_out/${GL_OUT:-r6_lanes}; mkdir -p $O; : > $O/lanes_sweep.txt
for r in 1 2; do for l in 2 3 4 1; do
  ( timeout 600 python bench.py --lanes $l --steps 12 --warmup 4 --no-cpu-baseline --no-train-step --no-ff-ab 2> $O/l$l.err ) > $O/l$l.json
  python -c "import json; d=json.loads(open('$O/l$l.json').read().strip().splitlines()[-1]); print('lanes $l value %.3f one_lane %.3f unet_step_ms %.3f ms_per_step %.1f' % (d['value'], d['value_one_lane'], d['unet_step_ms'], d['ms_per_step']))" | tee -a $O/lanes_sweep.txt
done; done
