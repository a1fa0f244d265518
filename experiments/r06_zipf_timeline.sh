# round 6: one ordered Zipf gradient-apply call launch by launch (kernel trace), dense route on / off, SGD and LazyAdam
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_grad_timeline_zipf_dense.txt
: > $OUT
for opt in sgd adam; do for mode in dense off; do
  d=/tmp/zt_${opt}_$mode; rm -rf $d
  if [ $mode = off ]; then export WM_DENSE_FOLD=0; else unset WM_DENSE_FOLD; fi
  rocprofv3 --kernel-trace --output-format csv -d $d -- python $R/bench.py --op grad_apply --dist zipf --optimizer $opt --no-cpu-baseline --steps 12 --warmup 4 --stability-steps 0 > /tmp/zt_line.json 2>/dev/null
  echo "==== $opt, dense route $mode: $(python3 -c "import json;print(json.loads(open('/tmp/zt_line.json').read().strip().splitlines()[-1])['ms_per_step'])") ms per step under the tracer" >> $OUT
  python3 $R/experiments/r05_timeline.py $(find $d -name "*kernel_trace.csv" | head -1) step_tile_kernel 2 >> $OUT
done; done
unset WM_DENSE_FOLD
cat $OUT
