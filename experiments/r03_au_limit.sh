#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03/c5
mkdir -p $OUT
cd $R
for lim in default 100000000; do
  if [ $lim = default ]; then unset WM_AU_TABLE_MAX; else export WM_AU_TABLE_MAX=$lim; fi
  for seeds in 8192 65536; do
    python bench.py --op sample_gather --seeds $seeds --steps 20 --stability-steps 0 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('table limit $lim seeds $seeds:', r['ms_per_step'], 'ms', r['roofline']['frac'], r['frontier_sizes'])"
  done
done
unset WM_AU_TABLE_MAX
for lim in 0 1000000000; do WM_AU_TABLE_MAX=$lim python experiments/au_crossover.py 2>&1 | grep "int32" ; done
