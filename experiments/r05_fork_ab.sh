#!/bin/bash
# where the tile kernel starts (us after the first kernel of the call), per variant of the side-stream set-up
cd $GRAFT_REPO_ROOT
i=0
for v in "WM_DEDUP_LANE_PRIO=n" "WM_DEDUP_LANE_PRIO=low" "WM_DEDUP_LANE_PRIO=n WM_DEDUP_FORK=3" "WM_DEDUP_LANE_PRIO=low WM_DEDUP_FORK=3" "WM_DEDUP_SERIAL=1"; do
  i=$((i+1))
  for back in 1 3 5; do :; done
  env $v TIMELINE=split_hist_kernel bash experiments/r05_prof.sh fork$i python $GRAFT_REPO_ROOT/bench.py --op grad_apply --no-cpu-baseline --steps 30 --stability-steps 0 < /dev/null > /dev/null
  t=$(find /tmp/prof_fork$i -name "*kernel_trace.csv" | head -1)
  echo "== $v"
  for back in 1 4 8 12; do python3 experiments/r05_timeline.py $t split_hist_kernel $back | grep -E "step_tile|split_scatter|split_sort_kernel" | awk '{printf "%s(%s) ", $1, $4}'; echo; done
done
