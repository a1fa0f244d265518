# refresh the gradient-apply side profiles: bench line (standalone) + rocprofv3 kernel stats of the same command
cd /tmp && export TMPDIR=/tmp
for d in uniform zipf; do
  tag=r01_grad_apply; [ $d = zipf ] && tag=r01_grad_apply_zipf
  python $GRAFT_REPO_ROOT/bench.py --op grad_apply --dist $d --no-cpu-baseline 2>/dev/null | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.json
  rm -rf /tmp/pg_$d
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg_$d -- python $GRAFT_REPO_ROOT/bench.py --op grad_apply --dist $d --steps 10 --no-cpu-baseline > /dev/null 2>&1
  cp $(find /tmp/pg_$d -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
  cut -c1-230 $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.json; head -3 $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv | cut -c1-150
done
cd $GRAFT_REPO_ROOT
for o in adam; do for d in uniform zipf; do python bench.py --op grad_apply --dist $d --optimizer $o --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('$o $d', r['ms_per_step'])"; done; done
python bench.py --op grad_apply --dist zipf --dtype f16 --dim 256 --rows 50000000 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('f16x256 zipf', r['ms_per_step'])"
python bench.py --op grad_apply --dist uniform --dtype f16 --dim 256 --rows 50000000 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('f16x256 uniform', r['ms_per_step'])"
