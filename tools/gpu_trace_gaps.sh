export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r6_trace; mkdir -p $O; rm -rf $R/gpurun_out/trc
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trc -- python $R/bench.py --lanes 1 --steps 2 --warmup 1 --no-cpu-baseline --no-train-step --no-ff-ab ) > $O/trace.log 2>&1
f=$(find $R/gpurun_out/trc -name "*kernel_trace.csv" | head -1); ls -la $f
python tools/trace_gaps.py $f > $O/trace_gaps_one_lane.txt; head -40 $O/trace_gaps_one_lane.txt
rm -rf $R/gpurun_out/trc
