#!/bin/bash
mkdir -p gpurun_out/r03
wait_empty() { for k in $(seq 1 100); do v=$(rocm-smi --showmemuse 2>/dev/null | grep "VRAM%" | awk '{print $NF}'); [ "$v" = "0" ] && return; sleep 0.3; done; }
OUT=gpurun_out/r03/tables_in_one_process.txt
: > $OUT
wait_empty; echo "== 4 tables, no blocker" >> $OUT; timeout 600 python experiments/tables_in_one_process.py 4 2>&1 | grep round >> $OUT
wait_empty; echo "== 4 tables, no blocker (second process)" >> $OUT; timeout 600 python experiments/tables_in_one_process.py 4 2>&1 | grep round >> $OUT
wait_empty; echo "== 3 tables after a 25.6 GB blocker" >> $OUT; BLOCKERS_GB=25.6 timeout 600 python experiments/tables_in_one_process.py 3 2>&1 | grep round >> $OUT
cat $OUT
