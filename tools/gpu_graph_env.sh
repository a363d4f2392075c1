# HIP runtime graph knobs against the replay time of one UNet evaluation (one lane) and the three-lane bench value, one box
export TMPDIR=/tmp
O=gpurun_out/${GL_OUT:-r6_lanes}; mkdir -p $O; : > $O/graph_env.txt
run() { tag="$1"; shift
  ( export "$@"; timeout 400 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-train-step --no-ff-ab 2> $O/g.err ) > $O/g.json
  python -c "import json; d=json.loads(open('$O/g.json').read().strip().splitlines()[-1]); print('$tag value %.3f one_lane %.3f unet_step_ms %.3f' % (d['value'], d['value_one_lane'], d['unet_step_ms']))" 2>&1 | tail -1 | tee -a $O/graph_env.txt; }
run default X=1
run packet_capture_0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run packet_capture_1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run batch_1 DEBUG_HIP_GRAPH_BATCH_SIZE=1
run batch_16 DEBUG_HIP_GRAPH_BATCH_SIZE=16
run batch_512 DEBUG_HIP_GRAPH_BATCH_SIZE=512
run force_queues_1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run force_queues_4 DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run sys_scope_0 ROC_SYSTEM_SCOPE_SIGNAL=0
run dev_kernarg HIP_FORCE_DEV_KERNARG=1
run default X=1
