#!/bin/bash
# round 6, final tree: gather / scatter / gradient apply (uniform; Zipf ordered; Zipf tree) in six fresh processes each, defaults only
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r06_six_fresh_processes.txt
: > $O
run() {  # run <label> <bench args...>
  local label=$1; shift
  for i in 1 2 3 4 5 6; do
    timeout 600 python bench.py "$@" --no-cpu-baseline --steps 100 --stability-steps 0 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
r = d.get('roofline') or {}
print('%-22s process %s  ms_per_step %.4f  frac_of_8TBps %s' % ('$label', '$i', d['ms_per_step'], r.get('frac')))
" >> $O
  done
}
run gather
run scatter --op scatter
run grad_apply_uniform --op grad_apply
run grad_apply_zipf_ordered --op grad_apply --dist zipf
WM_GRAD_FOLD=tree run grad_apply_zipf_tree --op grad_apply --dist zipf
cat $O
