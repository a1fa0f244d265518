#!/bin/bash
mkdir -p gpurun_out/r03
wait_empty() { for w in $(seq 1 100); do v=$(rocm-smi --showmemuse 2>/dev/null | grep "VRAM%" | awk '{print $NF}'); [ "$v" = "0" ] && return; sleep 0.3; done; }
OUT=gpurun_out/r03/malloc_probe2.txt
: > $OUT
wait_empty; echo "== 4 tables in one process: probes against scatter" >> $OUT; timeout 600 python experiments/tables_in_one_process.py 4 2>&1 | grep "round 0" >> $OUT
for k in 1 4 1 4 1 4; do
  wait_empty
  echo "== WM_MALLOC_PROBE=$k scatter" >> $OUT
  WM_MALLOC_PROBE_VERBOSE=1 WM_MALLOC_PROBE=$k timeout 300 tools/gather_scatter_bench -t chunked -l device -e 51200000000 -g 5120000000 -d 128 -c 20 -f scatter -n 1 2>&1 | grep -i "time per call\|malloc probe" | sed "s/^/   /" >> $OUT
done
cat $OUT
