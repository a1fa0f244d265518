#!/bin/bash
# large fuzz campaign on the final round-3 tree (new seeds; the test suite runs 2 x 250 / 120 / 40 / 150 cases of the same scripts)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r03_fuzz_campaign.txt
: > $O
run() { echo "== $*" >> $O; timeout 1500 "$@" 2>&1 | tail -12 >> $O; }
FUZZ_WIDE=1 run python experiments/fuzz_rows.py 2500 ${FUZZ_SEED_BASE:-301}
run python experiments/fuzz_rows.py 1500 302
WM_ROWS_INORDER=0 FUZZ_WIDE=1 run python experiments/fuzz_rows.py 800 303
run python experiments/fuzz_optim.py 800 304
run python experiments/fuzz_sample.py 800 306
run python experiments/fuzz_cache.py 150 307
run python experiments/fuzz_append_unique.py
cat $O
