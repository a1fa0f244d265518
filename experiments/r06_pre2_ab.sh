# round 6: the second row of a duplicated id fetched with the batch's loads (variant pre2 = -DWM_TILE_PRE2=1) against the product build
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_tile_pre2_ab.txt
: > $OUT
for rep in 1 2 3; do for v in product pre2; do for d in zipf uniform; do
  if [ $v = product ]; then unset WHOLEGRAPH_AMD_VARIANT; else export WHOLEGRAPH_AMD_VARIANT=$v; fi
  echo -n "$v: " >> $OUT
  timeout 300 python $R/experiments/grad_env_ab.py sgd $d 128 f32 "ordered:;tree:WM_GRAD_FOLD=tree" 2>&1 | grep "round 2" >> $OUT
done; done; done
unset WHOLEGRAPH_AMD_VARIANT
cat $OUT
