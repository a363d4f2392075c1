# Images/s against the images per evaluation (one lane) and against two whole batches in flight: what a caller's batch size buys.
# Usage: gpurun -- 'bash tools/gpu_batch_sweep.sh'   -> gpurun_out/$GL_OUT/batch_sweep.txt
export TMPDIR=/tmp
O=gpurun_out/${GL_OUT:-r6_b8}; mkdir -p $O
run() { name=$1; shift; ( timeout 700 python bench.py "$@" --no-cpu-baseline --no-train-step --no-ff-ab 2> $O/$name.err ) > $O/$name.json; python -c "import sys,json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name value %.3f one_lane %.3f unet_step_ms %.3f ms_per_step %.1f' % (d['value'], d['value_one_lane'], d['unet_step_ms'], d['ms_per_step']))" | tee -a $O/batch_sweep.txt; }
: > $O/batch_sweep.txt
run b4_l2 --steps 4
run b8_l1 --batch 8 --lanes 1 --steps 2
run b8_l2 --batch 8 --lanes 2 --steps 4
run b16_l1 --batch 16 --lanes 1 --steps 2
run b16_l2 --batch 16 --lanes 2 --steps 2
run b32_l1 --batch 32 --lanes 1 --steps 1
( timeout 600 python -m pytest tests/test_path_gpu.py -q -k "vae_decode" 2>&1 | tail -3 ) | tee -a $O/batch_sweep.txt
