#!/bin/bash
# HBM-side traffic of the bench command per kernel symbol: FETCH_SIZE and WRITE_SIZE in separate --pmc passes
# (--kernel-trace only; eager launches: counter collection over hipGraph replays crashes rocprofv3 here), folded to per-launch means -> gpurun_out/pmc_traffic.csv (copied to profiles/<round>/)
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
for c in f:FETCH_SIZE w:WRITE_SIZE; do
  n=${c%%:*}; ctr=${c#*:}
  ( cd /tmp && timeout 700 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$n -- \
      python $R/bench.py --steps 1 --warmup 0 --lanes 1 --no-graph --no-cpu-baseline ) > gpurun_out/pmc_$n.log 2>&1
  tail -1 gpurun_out/pmc_$n.log | cut -c1-300
done
python tools/pmc_summarize.py gpurun_out/pmc_traffic.csv gpurun_out/pmc_f gpurun_out/pmc_w
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
head -30 gpurun_out/pmc_traffic.csv
