export TMPDIR=/tmp
O=gpurun_out/r6_lanes; mkdir -p $O
for r in 1 2 3; do for l in 2 3; do
  ( timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --lanes $l --no-cpu-baseline --no-train-step --no-ff-ab 2> $O/d$l.err ) > $O/d$l.json
  python -c "import json; d=json.loads(open('$O/d$l.json').read().strip().splitlines()[-1]); print('driver-steps lanes $l value %.3f one_lane %.3f unet_step_ms %.3f ms_per_step %.1f' % (d['value'], d['value_one_lane'], d['unet_step_ms'], d['ms_per_step']))" | tee -a $O/lanes_sweep.txt
done; done
