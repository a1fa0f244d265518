set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
nproc; python -c "import os; print(len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max; lscpu | grep -E "Model name|Socket|NUMA node\(s\)|^CPU\(s\)"
cd $R
timeout 600 python bench.py --cpu-seconds 8 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.json
for op in scatter grad_apply; do
  timeout 300 python bench.py --op $op --memory-type distributed --no-cpu-baseline --steps 10 > gpurun_out/bench_$op.json 2>&1; tail -1 gpurun_out/bench_$op.json
done
timeout 300 python bench.py --op grad_apply --memory-type distributed --dist zipf --no-cpu-baseline --steps 10 > gpurun_out/bench_grad_zipf.json 2>&1; tail -1 gpurun_out/bench_grad_zipf.json
timeout 300 python bench.py --memory-type distributed --no-cpu-baseline --steps 10 > gpurun_out/bench_dist_w1.json 2>&1; tail -1 gpurun_out/bench_dist_w1.json
cd /tmp
for op in scatter grad_apply; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$op -- python $R/bench.py --op $op --memory-type distributed --no-cpu-baseline --steps 10 > /dev/null 2>&1
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sample -- python $R/experiments/sample_bench.py > $R/gpurun_out/sample_bench.log 2>&1
tail -5 $R/gpurun_out/sample_bench.log
find $R/gpurun_out -name "*kernel_stats.csv" | head
