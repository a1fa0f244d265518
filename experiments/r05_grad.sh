#!/bin/bash
# gradient apply after the split sort: forced-split test runs, then the bench lines (+ the timeline of one step)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
if [ "$1" != "notest" ]; then
WM_DEDUP_SPLIT_MIN=1 timeout 900 python -m pytest tests/test_exchange_optim_gpu.py tests/test_golden_fixtures_gpu.py tests/test_embedding_cache_gpu.py -x -q -m gpu 2>&1 | tail -3
fi
line() {
  python3 - $1 <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "ms_per_step", r["ms_per_step"], "frac", r.get("roofline", {}).get("frac"))
except Exception as e:
    print("no bench line:", e)
PY
}
for d in uniform zipf; do for o in sgd adam; do
  timeout 300 python bench.py --op grad_apply --dist $d --optimizer $o --no-cpu-baseline > gpurun_out/r05/grad_apply_${o}_$d.json 2> gpurun_out/r05/grad_apply_${o}_$d.err
  line gpurun_out/r05/grad_apply_${o}_$d.json
done; done
WM_GRAD_FOLD=tree timeout 300 python bench.py --op grad_apply --dist zipf --no-cpu-baseline > gpurun_out/r05/grad_apply_sgd_zipf_tree.json 2>/dev/null; line gpurun_out/r05/grad_apply_sgd_zipf_tree.json
TIMELINE=split_hist_kernel bash experiments/r05_prof.sh grad_uniform python $GRAFT_REPO_ROOT/bench.py --op grad_apply --no-cpu-baseline --steps 30 --stability-steps 0 < /dev/null > /dev/null; cat gpurun_out/r05/grad_uniform_timeline.txt
TIMELINE=split_hist_kernel bash experiments/r05_prof.sh grad_zipf python $GRAFT_REPO_ROOT/bench.py --op grad_apply --dist zipf --no-cpu-baseline --steps 30 --stability-steps 0 < /dev/null > /dev/null; cat gpurun_out/r05/grad_zipf_timeline.txt
