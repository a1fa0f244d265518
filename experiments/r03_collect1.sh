#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03/collect1
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/gpu_tests_tail.txt
for i in 1 2 3; do python bench.py --no-cpu-baseline > $OUT/bench_plain_$i.json 2> $OUT/bench_plain_$i.err; python - $OUT/bench_plain_$i.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); print(r["ms_per_step"], r["roofline"]["frac"], r["stability"], r.get("launch_shape"))
PY
done
DIM_SWEEP_SETTINGS=default,inorder=0 timeout 600 python experiments/dim_sweep.py --ab --csv=$OUT/dim_sweep_ragged_ab.csv 100 129 200 300 602 2>&1 | tee $OUT/dim_sweep_ragged_ab.txt | cut -c1-200
