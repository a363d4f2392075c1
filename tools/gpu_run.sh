#!/bin/bash
# One parameterised runner for every GPU call of this repo (replaces the one-shot tools/gpu_r2*.sh .. gpu_r4*.sh of rounds 2-4;
# `git log -- tools/` has them as they were run). Usage, always from the repo root through gpurun:
#     gpurun --timeout 900 -- 'bash tools/gpu_run.sh TASK [TASK ...]'        output under gpurun_out/$GL_OUT (default: run)
# Tasks (each prints a short summary; the full logs stay in the output folder):
#   smoke            __graft_entry__.smoke()
#   suite [-k EXPR]  pytest -m gpu (SUITE_K="expr" narrows it)
#   bench [ARGS]     python bench.py $BENCH_ARGS  (no developer switches: what the driver runs)
#   driver           the driver's literal command: python3 bench.py --gpus 1 --steps 20 --warmup 5
#   configs          bench lines of C3 C4 C5, the [0.3,0,0.7] gate schedule and --lanes 1
#   kbench FILE [N]  per-shape kernel table (gligen_amd/build/kbench FILE N - check)
#   insitu           per-problem event timing inside one eager UNet evaluation (tools/insitu.py)
#   prof             rocprofv3 --kernel-trace --stats of the bench command -> bench_kernel_stats.csv
#   traffic          FETCH_SIZE / WRITE_SIZE PMC passes per symbol and per problem (tools/gpu_traffic.sh)
#   mfma             MFMA-busy / LDS-conflict PMC passes per symbol (tools/gpu_mfma_util.sh)
#   train            tools/train_bench.py $TRAIN_ARGS (default --b4only: the bench line's train_step shape; --full64: every configuration)
#                    + rocprofv3 stats of one B = 4 iteration with TRAIN_PROF=1
#   ab VAR A B [N]   same-box A/B of a developer switch: bench.py alternated N times (default 2) with VAR=A / VAR=B
#   abn VAR N v1 v2 .. --   the same for several values of VAR, N rounds (end the value list with --)
#   tune NAME [ENV=V ..] --   bench.py with the tile table ignored and every GEMM problem tuned on the device under the given switches
#                    (GL_GEMM_TUNE_CORUN=1: candidates timed as two streams side by side); log -> tune_NAME.log (input of make_tuned_table.py)
#   calib            gl_mfma_calibrate table (tools/calib_mfma.py): MFMA ceiling by shape / waves per SIMD / accumulators with the in-loop clock
#   lib DIR          swap in the variant library built by tools/build_variant.sh for the tasks that follow
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${GL_OUT:-run}
mkdir -p $O
K=gligen_amd/build/kbench
dev() { export GL_DEV_SWITCHES=1; }        # the library reads its developer switches only with this set
nodev() { unset GL_DEV_SWITCHES; }
while [ $# -gt 0 ]; do
  task=$1; shift
  case $task in
    smoke) dev; ( timeout 600 python __graft_entry__.py --smoke ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-240 ;;
    suite) dev; ( timeout 1700 python -m pytest tests -m gpu -q ${SUITE_K:+-k "$SUITE_K"} ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
           grep -E "^FAILED|^ERROR|passed|failed|pytest rc" $O/pytest_gpu.log | cut -c1-250 | tee $O/pytest_gpu_summary.txt
           cp gpurun_out/parity_report.json gpurun_out/parity_report_configs.json $O/ 2>/dev/null ;;
    bench) nodev; ( timeout 900 python bench.py $BENCH_ARGS > $O/bench_default.json 2> $O/bench_default.err ); cut -c1-600 $O/bench_default.json; tail -2 $O/bench_default.err | cut -c1-200 ;;
    driver) nodev; ( timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ); cut -c1-400 $O/bench.json; tail -2 $O/bench.err | cut -c1-200 ;;
    configs) nodev
           for c in C3 C4 C5; do timeout 500 python bench.py --config $c --steps 3 --no-cpu-baseline --no-train-step > $O/bench_$c.json 2> $O/bench_$c.err; cut -c1-140 $O/bench_$c.json; done
           timeout 500 python bench.py --alpha-type 0.3,0,0.7 --steps 3 --no-cpu-baseline --no-train-step > $O/bench_alpha.json 2> $O/bench_alpha.err; cut -c1-140 $O/bench_alpha.json
           timeout 500 python bench.py --lanes 1 --steps 2 --no-cpu-baseline --no-train-step > $O/bench_l1.json 2> $O/bench_l1.err; cut -c1-140 $O/bench_l1.json ;;
    kbench) dev; f=$1; shift; n=5; case "$1" in ''|*[!0-9]*) ;; *) n=$1; shift ;; esac
           b=$(basename $f .shapes); timeout 400 $K $f $n - check > $O/kbench_$b.txt 2>&1; grep -E "^TOTAL|CHECK|MISMATCH|^FFN|^ATT" $O/kbench_$b.txt | cut -c1-200 | tail -40 ;;
    insitu) dev; timeout 400 python tools/insitu.py > $O/insitu_per_problem.txt 2> $O/insitu.err; head -1 $O/insitu_per_problem.txt ;;
    prof) nodev; rm -rf gpurun_out/prof
           ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python $R/bench.py --no-cpu-baseline --no-train-step --no-ff-ab ) > $O/prof.log 2>&1
           find gpurun_out/prof -name "*kernel_trace*" -delete
           cp $(find gpurun_out/prof -name "*kernel_stats*" | head -1) $O/bench_kernel_stats.csv; head -8 $O/bench_kernel_stats.csv | cut -c1-160 ;;
    traffic) bash tools/gpu_traffic.sh > $O/traffic.log 2>&1; cp gpurun_out/pmc_traffic.csv gpurun_out/pmc_traffic_per_problem.csv $O/; tail -12 $O/traffic.log | cut -c1-200 ;;
    mfma) bash tools/gpu_mfma_util.sh > $O/mfma.log 2>&1; cp gpurun_out/pmc_mfma.csv gpurun_out/pmc_mfma_report.txt $O/; head -12 $O/pmc_mfma_report.txt | cut -c1-150 ;;
    train) dev; ( PYTHONPATH=. timeout 900 python tools/train_bench.py ${TRAIN_ARGS:---b4only} ) > $O/train_bench.txt 2> $O/train_bench.err; cat $O/train_bench.txt | cut -c1-260
           if [ -n "$TRAIN_PROF" ]; then rm -rf gpurun_out/tprof
             ( cd /tmp && PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tprof -- python $R/tools/train_bench.py --b4only ) > $O/train_prof.log 2>&1
             find gpurun_out/tprof -name "*kernel_trace*" -delete; cp $(find gpurun_out/tprof -name "*kernel_stats*" | head -1) $O/train_kernel_stats.csv; head -16 $O/train_kernel_stats.csv | cut -c1-170; fi ;;
    ab) dev; var=$1; a=$2; b=$3; shift 3; n=2; case "$1" in ''|*[!0-9]*) ;; *) n=$1; shift ;; esac
           : > $O/ab_$var.txt
           for i in $(seq $n); do for v in $a $b; do
             ( export $var=$v; timeout 500 python bench.py --steps 6 --no-cpu-baseline --no-train-step --no-ff-ab 2>/dev/null ) | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v value %.3f one_lane %.3f unet_step_ms %.3f eager_sum %.3f' % (d['value'], d['value_one_lane'], d['unet_step_ms'], d['roofline']['eager_sum_ms']))" | tee -a $O/ab_$var.txt
           done; done ;;
    abn) dev; var=$1; n=$2; shift 2; vals=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done; [ "$1" = "--" ] && shift
           : > $O/ab_$var.txt
           for i in $(seq $n); do for v in "${vals[@]}"; do
             ( export $var=$v; timeout 500 python bench.py --steps 6 --no-cpu-baseline --no-train-step --no-ff-ab 2>/dev/null ) | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v value %.3f one_lane %.3f unet_step_ms %.3f eager_sum %.3f' % (d['value'], d['value_one_lane'], d['unet_step_ms'], d['roofline']['eager_sum_ms']))" | tee -a $O/ab_$var.txt
           done; done ;;
    tune) dev; name=$1; shift; envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; [ "$1" = "--" ] && shift
           ( export GL_GEMM_NO_TABLE=1 GL_GEMM_TUNE_LOG=1 GL_GEMM_TUNE_REPS=${TUNE_REPS:-10} "${envs[@]}"; timeout 800 python bench.py --steps 6 --no-cpu-baseline --no-train-step --no-ff-ab 2> $O/tune_$name.log ) | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tune $name value %.3f one_lane %.3f unet_step_ms %.3f eager_sum %.3f' % (d['value'], d['value_one_lane'], d['unet_step_ms'], d['roofline']['eager_sum_ms']))" | tee -a $O/tune_summary.txt
           grep -c "gemm autotune" $O/tune_$name.log ;;
    calib) dev; ( PYTHONPATH=. timeout 300 python tools/calib_mfma.py ${CALIB_MS:-40} ) > $O/calib_mfma.txt 2> $O/calib_mfma.err; cat $O/calib_mfma.txt | cut -c1-200; tail -2 $O/calib_mfma.err | cut -c1-200 ;;
    lib) d=$1; shift; cp $d/libgligen_amd.so gligen_amd/libgligen_amd.so; [ -f $d/kbench ] && cp $d/kbench $K; echo "library <- $d" ;;
    *) echo "unknown task $task"; exit 2 ;;
  esac
done
