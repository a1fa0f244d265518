#!/bin/bash
# round 3: (1) does the placement probe track the scatter level of a table? (2) WM_MALLOC_PROBE=K in a situation known to give a slow
# table (a 51.2 GB allocation made before the table, see r03_scatter_by_vram_offset.txt)
mkdir -p gpurun_out/r03
wait_empty() { for k in $(seq 1 100); do v=$(rocm-smi --showmemuse 2>/dev/null | grep "VRAM%" | awk '{print $NF}'); [ "$v" = "0" ] && return; sleep 0.3; done; }
OUT=gpurun_out/r03/probe_vs_scatter.txt
: > $OUT
for r in 1 2 3; do wait_empty; echo "== process $r: 4 tables" >> $OUT; timeout 600 python experiments/tables_in_one_process.py 4 2>&1 | grep round >> $OUT; done
echo "== tools/gather_scatter_bench, a 51.2 GB allocation before the table: WM_MALLOC_PROBE unset / 2 / 3, scatter then gather" >> $OUT
for k in 1 2 3 1 3; do
  for op in scatter gather; do
    wait_empty
    WM_MALLOC_PROBE=$k WM_BENCH_BLOCKER_GB=51.2 timeout 300 tools/gather_scatter_bench -t chunked -l device -e 51200000000 -g 5120000000 -d 128 -c 20 -f $op -n 1 2>&1 | grep -i "time per call" | sed "s/^/   WM_MALLOC_PROBE=$k $op: /" >> $OUT
  done
done
echo "== the same without the allocation before the table" >> $OUT
for k in 1 3; do
  for op in scatter gather; do
    wait_empty
    WM_MALLOC_PROBE=$k timeout 300 tools/gather_scatter_bench -t chunked -l device -e 51200000000 -g 5120000000 -d 128 -c 20 -f $op -n 1 2>&1 | grep -i "time per call" | sed "s/^/   WM_MALLOC_PROBE=$k $op: /" >> $OUT
  done
done
cat $OUT
