# per-kernel times of the Zipf gradient apply under WM_LONG_EXP variants (experiment switch in optim.hip)
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/exp_long
for e in ${EXPS:-0 1 2}; do
  rm -rf /tmp/prof_$e
  WM_LONG_EXP=$e rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$e -- python $GRAFT_REPO_ROOT/bench.py --op grad_apply --dist zipf --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/exp_long/b_$e.log 2>&1
  f=$(find /tmp/prof_$e -name "*kernel_stats.csv" | head -1)
  echo "== exp $e  ($f)"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'step_' in r['Name']:
        print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3, 1), 'us')
PY
  tail -2 $GRAFT_REPO_ROOT/gpurun_out/exp_long/b_$e.log | cut -c1-300
done
