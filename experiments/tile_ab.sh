# A/B of the gradient-apply step kernels: WM_STEP_TILE=0 (wave-per-run step_short_kernel) vs the tile kernel, whole call
cd $GRAFT_REPO_ROOT
for o in sgd adam; do for d in uniform zipf; do for t in 0 1; do
  WM_STEP_TILE=$t python bench.py --op grad_apply --dist $d --optimizer $o --no-cpu-baseline --steps 30 --stability-steps 100 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('$o $d tile=$t', r['ms_per_step'], r['stability']['min_ms'], r['stability']['median_ms'])"
done; done; done
