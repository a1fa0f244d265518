#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03/batch
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gather_scatter_gpu.py tests/test_golden_fixtures_gpu.py tests/test_fuzz_gpu.py tests/test_full_size_gpu.py tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -5
for f in gather scatter; do for i in 1 2; do
  echo "== process $i: WM_BENCH_AB=WM_ROWS_BATCH tools/gather_scatter_bench ... -p 4 -f $f" >> $OUT/cpp_batch_ab.txt
  WM_BENCH_AB=WM_ROWS_BATCH timeout 300 tools/gather_scatter_bench -t chunked -l device -e 51200000000 -g 5120000000 -d 128 -c 20 -p 4 -f $f >> $OUT/cpp_batch_ab.txt 2>&1
done; done
cat $OUT/cpp_batch_ab.txt | grep -E "candidate|process|time per"
DIM_SWEEP_SETTINGS=default,batch=0,block=64 timeout 600 python experiments/dim_sweep.py --ab --csv=$OUT/dim_sweep_batch.csv 128 256 512 1024 2>&1 | cut -c1-200 | tee $OUT/dim_sweep_batch.txt
