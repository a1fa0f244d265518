#!/bin/bash
cd $GRAFT_REPO_ROOT
line() { python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % r['ms_per_step'], end=' ')"; }
for rep in 1 2; do
for d in 1 0; do
  echo -n "detach=$d: u0.1M "
  WM_STEP_DETACH=$d python bench.py --op grad_apply --indices 100000 --no-cpu-baseline --stability-steps 0 --steps 100 2>/dev/null | line
  echo -n " u0.5M "
  WM_STEP_DETACH=$d python bench.py --op grad_apply --indices 500000 --no-cpu-baseline --stability-steps 0 --steps 50 2>/dev/null | line
  echo -n " u10M "
  WM_STEP_DETACH=$d python bench.py --op grad_apply --no-cpu-baseline --stability-steps 0 --steps 30 2>/dev/null | line
  echo -n " adam10M "
  WM_STEP_DETACH=$d python bench.py --op grad_apply --optimizer adam --no-cpu-baseline --stability-steps 0 --steps 20 2>/dev/null | line
  for f in ordered tree; do echo -n " zipf-$f "
    WM_GRAD_FOLD=$f WM_STEP_DETACH=$d python bench.py --op grad_apply --dist zipf --no-cpu-baseline --stability-steps 0 --steps 30 2>/dev/null | line
  done; echo
done
done
