# refresh the committed round profiles: contract bench line + rocprofv3 kernel stats of the same command
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/r01_n1_bench.json 2> gpurun_out/r01_n1_bench.err; tail -c 400 gpurun_out/r01_n1_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_final -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/r01_n1_bench_kernel_stats.csv; head -3 $f | cut -c1-200
