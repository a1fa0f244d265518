#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03/tree
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_exchange_optim_gpu.py tests/test_full_size_gpu.py -m gpu -x -q -k "tree or 16bit or c4" 2>&1 | tail -6
( timeout 300 python experiments/grad_env_ab.py sgd zipf 128 f32 "ordered:WM_GRAD_FOLD=ordered;tree32:WM_GRAD_FOLD=tree;tree64:WM_GRAD_FOLD=tree,WM_GRAD_FOLD_MIN=64;tree128:WM_GRAD_FOLD=tree,WM_GRAD_FOLD_MIN=128;tree256:WM_GRAD_FOLD=tree,WM_GRAD_FOLD_MIN=256;tree192:WM_GRAD_FOLD=tree,WM_GRAD_FOLD_MIN=192"
  timeout 300 python experiments/grad_env_ab.py sgd uniform 128 f32 "ordered:WM_GRAD_FOLD=ordered;tree:WM_GRAD_FOLD=tree"
  timeout 300 python experiments/grad_env_ab.py adam zipf 128 f32 "ordered:WM_GRAD_FOLD=ordered;tree32:WM_GRAD_FOLD=tree;tree128:WM_GRAD_FOLD=tree,WM_GRAD_FOLD_MIN=128"
  timeout 300 python experiments/grad_env_ab.py sgd zipf 256 f16 "ordered:WM_GRAD_FOLD=ordered;tree32:WM_GRAD_FOLD=tree;tree128:WM_GRAD_FOLD=tree,WM_GRAD_FOLD_MIN=128" ) 2>&1 | grep -v amdgpu.ids | tee $OUT/tree_threshold_ab.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_tree && WM_GRAD_FOLD=tree timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tree -- python $R/bench.py --op grad_apply --dist zipf --optimizer sgd --no-cpu-baseline --steps 20 --stability-steps 0 > $OUT/sgd_zipf_tree_under_rocprof.json 2>/dev/null
cp $(find /tmp/prof_tree -name "*kernel_stats.csv" | head -1) $OUT/sgd_zipf_tree_kernel_stats.csv
python3 - $OUT/sgd_zipf_tree_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("%-80s calls %4s avg %9.1f us" % (r["Name"][:80], r["Calls"], float(r["AverageNs"])/1e3))
PY
