#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches only with this set
# round 4 evidence run, part $1 (a: full GPU suite + bench lines of every BASELINE configuration, gate schedule, one batch in flight,
# the driver's literal command; b: per-shape kbench tables, in-situ per-problem table, rocprofv3 kernel stats of the bench command,
# FETCH / WRITE PMC passes per symbol and per problem, MFMA / LDS counters) -> gpurun_out/r4_final/ (copied to profiles/r4_final/)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r4_final
mkdir -p $O
K=gligen_amd/build/kbench
part=${1:-a}
if [ "$part" = a ]; then
( timeout 600 python __graft_entry__.py --smoke ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-200
( timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "^FAILED|^ERROR|passed|failed|pytest rc" $O/pytest_gpu.log | cut -c1-250 | tee $O/pytest_gpu_summary.txt
cp gpurun_out/parity_report.json gpurun_out/parity_report_configs.json $O/ 2>/dev/null
( unset GL_DEV_SWITCHES; timeout 700 python bench.py > $O/bench_default.json 2> $O/bench_default.err )     # no flags
cut -c1-400 $O/bench_default.json; tail -2 $O/bench_default.err | cut -c1-200
( unset GL_DEV_SWITCHES; timeout 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err )   # exactly the driver's command
cut -c1-300 $O/bench.json
for c in C3 C4 C5; do
  timeout 400 python bench.py --config $c --steps 2 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
  cut -c1-140 $O/bench_$c.json; tail -1 $O/bench_$c.err | cut -c1-200
done
timeout 400 python bench.py --alpha-type 0.3,0,0.7 --steps 2 --no-cpu-baseline > $O/bench_alpha.json 2> $O/bench_alpha.err
cut -c1-140 $O/bench_alpha.json
timeout 400 python bench.py --lanes 1 --steps 2 --no-cpu-baseline > $O/bench_l1.json 2> $O/bench_l1.err
cut -c1-140 $O/bench_l1.json
else
timeout 300 $K tools/unet_b8.shapes 10 - check > $O/kbench_unet.txt 2>&1
grep "^TOTAL\|CHECK\|MISMATCH" $O/kbench_unet.txt | cut -c1-160
timeout 300 $K tools/vae_b4.shapes 5 - check > $O/kbench_vae.txt 2>&1
tail -3 $O/kbench_vae.txt
timeout 120 $K tools/ffn.shapes 5 > $O/kbench_ffn.txt 2>&1
grep "^FFN" $O/kbench_ffn.txt | cut -c1-170
timeout 300 python tools/insitu.py > $O/insitu_per_problem.txt 2> $O/insitu.err
head -1 $O/insitu_per_problem.txt
rm -rf gpurun_out/prof
( cd /tmp && unset GL_DEV_SWITCHES && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python $R/bench.py --no-cpu-baseline ) > $O/prof.log 2>&1
tail -2 $O/prof.log | cut -c1-300
find gpurun_out/prof -name "*kernel_trace*" -delete
cp $(find gpurun_out/prof -name "*kernel_stats*" | head -1) $O/bench_kernel_stats.csv
head -12 $O/bench_kernel_stats.csv | cut -c1-200
bash tools/gpu_traffic.sh > $O/traffic.log 2>&1
tail -26 $O/traffic.log | cut -c1-220
cp gpurun_out/pmc_traffic.csv gpurun_out/pmc_traffic_per_problem.csv $O/
bash tools/gpu_mfma_util.sh > $O/mfma.log 2>&1
cp gpurun_out/pmc_mfma.csv gpurun_out/pmc_mfma_report.txt $O/
head -10 $O/pmc_mfma_report.txt | cut -c1-150
fi
