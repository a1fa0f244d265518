#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in low high normal; do
  for n in 500000 10000000; do
    WM_DEDUP_LANE_PRIO=$v TIMELINE=split_hist_kernel bash experiments/r05_prof.sh prio_${v}_$n python $GRAFT_REPO_ROOT/bench.py --op grad_apply --indices $n --no-cpu-baseline --steps 30 --stability-steps 0 < /dev/null > /dev/null
    t=$(find /tmp/prof_prio_${v}_$n -name "*kernel_trace.csv" | head -1)
    echo "== prio $v n $n: start of the tile kernel after the call's first kernel (4 steps)"
    for back in 1 4 8 12; do python3 experiments/r05_timeline.py $t split_hist_kernel $back | grep -E "step_tile" | awk '{printf "%s ", $1}'; done; echo
  done
done
