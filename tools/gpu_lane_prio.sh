export TMPDIR=/tmp GL_DEV_SWITCHES=1
O=gpurun_out/r6_lanes; mkdir -p $O; : > $O/prio.txt
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" | tee -a $O/prio.txt
for r in 1 2; do for p in "0,0,0" "-1,0,0" "-1,-1,0" "-1,0,1"; do
  ( export GL_LANE_PRIO="$p"; timeout 400 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-train-step --no-ff-ab 2> $O/p.err ) > $O/p.json
  python -c "import json; d=json.loads(open('$O/p.json').read().strip().splitlines()[-1]); print('lane priorities $p value %.3f one_lane %.3f' % (d['value'], d['value_one_lane']))" 2>&1 | tail -1 | tee -a $O/prio.txt
done; done
