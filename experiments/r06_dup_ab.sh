# round 6: duplicates of a run prefetched kDup at a time in step_tile_kernel (WM_TILE_DUP = 2 shipped; variants dup4 / dup8), whole calls
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_tile_dup_ab.txt
: > $OUT
for rep in 1 2; do for v in ab dup4 dup8; do for d in zipf uniform; do
  echo -n "variant $v (ab = kDup 2): " >> $OUT
  WHOLEGRAPH_AMD_VARIANT=$v timeout 300 python $R/experiments/grad_env_ab.py sgd $d 128 f32 "ordered:;tree:WM_GRAD_FOLD=tree" 2>&1 | grep "round 2" >> $OUT
done; done; done
cat $OUT
