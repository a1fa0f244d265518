#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for b in "" 256 128; do
 WM_ROWS_BLOCK=$b timeout 600 python bench.py --op sample_gather --steps 200 --stability-steps 0 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C5 deferred flow with hint, WM_ROWS_BLOCK=${b:-default(64)}: ms_per_step', d['ms_per_step'])"
done; done
WM_ROWS_BLOCK=256 bash experiments/trace_c5.sh 2>&1 | tail -3
