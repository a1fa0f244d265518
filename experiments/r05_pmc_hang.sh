#!/bin/bash
# which arrangement hangs under rocprofv3 --pmc (one kernel at a time)?
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
try() { name=$1; lim=$2; shift; shift
  rm -rf /tmp/h_$name
  s=$(date +%s)
  env "$@" timeout $lim rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/h_$name -- python $R/bench.py --op grad_apply --no-cpu-baseline --steps 5 --warmup 2 --stability-steps 0 > /tmp/h_$name.out 2> /tmp/h_$name.err < /dev/null
  echo "$name: rc $? after $(( $(date +%s) - s )) s; counter rows: $(cat $(find /tmp/h_$name -name '*counter_collection.csv' | head -1) 2>/dev/null | wc -l)"
  tail -3 /tmp/h_$name.err | cut -c1-300; tail -1 /tmp/h_$name.out | cut -c1-200
}
try auto1 400 X=1
try one_stream 400 WM_DEDUP_SERIAL=1 WM_STEP_SERIAL=1
