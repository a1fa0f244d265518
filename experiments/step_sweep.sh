cd $GRAFT_REPO_ROOT
cp wholegraph_amd/libwholegraph.so /tmp/lib_orig.so
run() { WM_STEP_BLOCKS=$1 timeout 300 python bench.py --op grad_apply --memory-type distributed --no-cpu-baseline --steps 10 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])"; }
for v in K2 K8; do
  cp experiments/variants/libwholegraph_$v.so wholegraph_amd/libwholegraph.so
  for b in 1280 2048 8192; do echo -n "$v blocks=$b uniform: "; run $b; done
  echo -n "$v blocks=8192 zipf: "; run 8192 "--dist zipf"
done
cp /tmp/lib_orig.so wholegraph_amd/libwholegraph.so
