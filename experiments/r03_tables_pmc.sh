#!/bin/bash
# round 3: hardware counters of the scatter per TABLE (4 tables alive in one process, the level differs by table): one rocprofv3
# --pmc pass per counter set, every pass classifies its own tables by the kernel durations of the same pass
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03/tables_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
wait_empty() { for k in $(seq 1 100); do v=$(rocm-smi --showmemuse 2>/dev/null | grep "VRAM%" | awk '{print $NF}'); [ "$v" = "0" ] && return; sleep 0.3; done; }
i=0
for set in "TCC_EA0_WRREQ" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" \
           "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_WRITE_GMI_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_GMI_32B_sum" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
           "TCC_EA0_WRREQ_DRAM" "TCC_EA0_WRREQ_LEVEL"; do
  i=$((i+1))
  rm -rf /tmp/tp_$i
  wait_empty
  REPS=2 ROUNDS=1 timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/tp_$i -- python $R/experiments/tables_in_one_process.py 4 > $OUT/pass_${i}_stdout.txt 2>&1
  f=$(find /tmp/tp_$i -name "*counter_collection.csv" | head -1); k=$(find /tmp/tp_$i -name "*kernel_trace.csv" | head -1)
  echo "== pass $i: $set" | tee -a $OUT/summary.txt
  grep round $OUT/pass_${i}_stdout.txt | tee -a $OUT/summary.txt
  [ -n "$f" ] && python3 $R/experiments/tables_pmc_reduce.py $f $k 7 | tee -a $OUT/summary.txt
done
