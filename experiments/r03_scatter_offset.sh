#!/bin/bash
# round 3: does the scatter's level follow WHERE the table sits in VRAM? One process per setting, each started only after the
# previous one's memory is gone (VRAM% back to 0); blockers of a, b, ... GB are allocated before the table and kept / freed.
mkdir -p gpurun_out/r03
OUT=gpurun_out/r03/scatter_by_vram_offset.txt
: > $OUT
wait_empty() { for k in $(seq 1 100); do v=$(rocm-smi --showmemuse 2>/dev/null | grep "VRAM%" | awk '{print $NF}'); [ "$v" = "0" ] && return; sleep 0.3; done; }
run() {  # run <label> <blockers> <free>
  wait_empty
  echo "== $1 (blockers GB: ${2:-none}, freed after the table exists: ${3:-0})" >> $OUT
  for op in scatter gather; do
    wait_empty
    WM_BENCH_BLOCKER_GB=$2 WM_BENCH_BLOCKER_FREE=${3:-0} timeout 300 tools/gather_scatter_bench -t chunked -l device -e 51200000000 -g 5120000000 -d 128 -c 20 -f $op -n 1 2>&1 | grep -i "time per call" | sed "s/^/   $op: /" >> $OUT
  done
}
run "table first" "" 0
run "51 GB before" 51.2 0
run "102 GB before" 51.2,51.2 0
run "154 GB before" 51.2,51.2,51.2 0
run "205 GB before" 51.2,51.2,51.2,51.2 0
run "table first (again)" "" 0
run "102 GB before (again)" 51.2,51.2 0
run "102 GB before, freed" 51.2,51.2 1
run "154 GB before, freed" 51.2,51.2,51.2 1
run "26 GB before" 25.6 0
run "77 GB before" 25.6,51.2 0
run "5 GB before" 5.12 0
cat $OUT
