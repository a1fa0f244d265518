#!/bin/bash
# large fuzz campaign on the final round-4 tree (straight-line row kernels, chained scans) (new seeds; the test suite runs 2 x 250 / 120 / 40 / 150 cases of the same scripts)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r04_fuzz_campaign.txt
: > $O
run() { echo "== $*" >> $O; timeout 1500 "$@" 2>&1 | tail -12 >> $O; }
FUZZ_WIDE=1 run python experiments/fuzz_rows.py ${FUZZ_N_A:-2500} ${FUZZ_SEED_BASE:-401}
run python experiments/fuzz_rows.py 1500 ${FUZZ_SEED_B:-402}
WM_ROWS_INORDER=0 FUZZ_WIDE=1 run python experiments/fuzz_rows.py 800 ${FUZZ_SEED_C:-403}
run python experiments/fuzz_optim.py ${FUZZ_N_D:-800} ${FUZZ_SEED_D:-404}
run python experiments/fuzz_sample.py ${FUZZ_N_E:-800} ${FUZZ_SEED_E:-406}
run python experiments/fuzz_cache.py ${FUZZ_N_F:-150} ${FUZZ_SEED_F:-407}
run python experiments/fuzz_append_unique.py
cat $O
