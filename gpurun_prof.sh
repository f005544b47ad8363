export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1 -- python3 $R/bench.py --ppd 2048 --plt 1 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_r1_bench.log 2>&1
tail -1 $R/gpurun_out/prof_r1_bench.log | cut -c1-600
find $R/gpurun_out/prof_r1 -name "*stats*" | head
f=$(find $R/gpurun_out/prof_r1 -name "*kernel_stats.csv" | head -1); cat $f | head -20
