#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03/tree
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_exchange_optim_gpu.py -m gpu -x -q -k "tree_fold or dedup_apply or sgd_on_16bit" 2>&1 | tail -15
for fold in ordered tree; do for d in zipf uniform; do
  WM_GRAD_FOLD=$fold python bench.py --op grad_apply --dist $d --optimizer sgd --no-cpu-baseline --steps 50 --stability-steps 0 > $OUT/sgd_${d}_${fold}.json 2>/dev/null
  python - $OUT/sgd_${d}_${fold}.json $fold $d <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); print(sys.argv[2], sys.argv[3], "sgd f32:", r["ms_per_step"], "ms", r.get("roofline",{}).get("frac"))
PY
done; done
for fold in ordered tree; do
  WM_GRAD_FOLD=$fold python bench.py --op grad_apply --dist zipf --optimizer adam --no-cpu-baseline --steps 30 --stability-steps 0 > $OUT/adam_zipf_${fold}.json 2>/dev/null
  WM_GRAD_FOLD=$fold python bench.py --op grad_apply --dist zipf --optimizer sgd --dtype f16 --dim 256 --rows 50000000 --no-cpu-baseline --steps 30 --stability-steps 0 > $OUT/sgd16_zipf_${fold}.json 2>/dev/null
  python - $OUT/adam_zipf_${fold}.json $OUT/sgd16_zipf_${fold}.json $fold <<'PY'
import json,sys
print(sys.argv[3], "adam zipf:", json.load(open(sys.argv[1]))["ms_per_step"], " f16x256 zipf:", json.load(open(sys.argv[2]))["ms_per_step"])
PY
done
for m in 16 64 128; do WM_GRAD_FOLD=tree WM_GRAD_FOLD_MIN=$m python bench.py --op grad_apply --dist zipf --optimizer sgd --no-cpu-baseline --steps 30 --stability-steps 0 2>/dev/null | python -c "import json,sys; print('tree min $m:', json.loads(sys.stdin.read())['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_tree && WM_GRAD_FOLD=tree timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tree -- python $R/bench.py --op grad_apply --dist zipf --optimizer sgd --no-cpu-baseline --steps 20 --stability-steps 0 > $OUT/sgd_zipf_tree_under_rocprof.json 2>/dev/null
cp $(find /tmp/prof_tree -name "*kernel_stats.csv" | head -1) $OUT/sgd_zipf_tree_kernel_stats.csv; head -12 $OUT/sgd_zipf_tree_kernel_stats.csv | cut -c1-150
