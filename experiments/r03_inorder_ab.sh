#!/bin/bash
# Round 3: the in-order launch shape of the row kernels against the persistent one (WM_ROWS_INORDER=0), same buffers, one process.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03/inorder
mkdir -p $OUT
cd $R
for f in gather scatter; do for i in 1 2 3; do
  echo "== process $i: WM_BENCH_AB=WM_ROWS_INORDER tools/gather_scatter_bench -t chunked -l device -e 51200000000 -g 5120000000 -d 128 -c 20 -p 6 -f $f" >> $OUT/cpp_ab.txt
  WM_BENCH_AB=WM_ROWS_INORDER timeout 300 tools/gather_scatter_bench -t chunked -l device -e 51200000000 -g 5120000000 -d 128 -c 20 -p 6 -f $f >> $OUT/cpp_ab.txt 2>&1
done; done
cat $OUT/cpp_ab.txt
DIM_SWEEP_SETTINGS=default,inorder=0,block=64 timeout 900 python experiments/dim_sweep.py --ab --csv=$OUT/dim_sweep_inorder_ab.csv 32 64 100 128 129 200 256 300 512 602 1024 2>&1 | tee $OUT/dim_sweep_inorder_ab.txt | cut -c1-200
