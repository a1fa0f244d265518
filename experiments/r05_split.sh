#!/bin/bash
# split sort harness: correctness log summary, timings, phase times, kernel stats
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 200 ./experiments/split_sort_test time > gpurun_out/r05/split_sort_test.log 2>&1; echo "rc=$? ok-cases=$(grep -c ': ok' gpurun_out/r05/split_sort_test.log)"
grep -E "MISMATCH|expected|FAILED|ALL OK|timing|flagged" gpurun_out/r05/split_sort_test.log | head -30
timeout 100 ./experiments/split_sort_test time 0 2>&1 | tail -2
bash experiments/r05_prof.sh split $GRAFT_REPO_ROOT/experiments/split_sort_test time < /dev/null | grep split
