#!/bin/bash
# Round 3, item 1(b): per-class (fast / slow / contiguous output buffer) hardware counters of the 512 B-row gather.
# gpurun -- 'bash experiments/placement_pmc.sh'
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03/placement
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 $R/experiments/placement_pmc 8 6 > $OUT/plain_run1.txt 2>&1; tail -30 $OUT/plain_run1.txt
timeout 300 $R/experiments/placement_pmc 8 6 > $OUT/plain_run2.txt 2>&1; grep -E "classes|round 0|copy" $OUT/plain_run2.txt
i=0
for set in "TCC_EA0_RDREQ TCC_EA0_WRREQ" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_WRITE_GMI_32B_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ" \
           "TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ" \
           "TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum"; do
  i=$((i+1))
  rm -rf /tmp/pp_$i
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv json -d /tmp/pp_$i -- $R/experiments/placement_pmc 6 3 > $OUT/pmc_${i}_stdout.txt 2>&1
  f=$(find /tmp/pp_$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $set -> $f"
  grep -E "classes" $OUT/pmc_${i}_stdout.txt
  if [ $i = 1 -o $i = 8 -o $i = 9 ]; then j=$(find /tmp/pp_$i -name "*results.json" | head -1); [ -n "$j" ] && gzip -c $j > $OUT/pmc_${i}_results.json.gz && ls -la $OUT/pmc_${i}_results.json.gz; fi
  [ -n "$f" ] && cp $f $OUT/pmc_${i}_counter_collection.csv
  [ -n "$f" ] && python3 $R/experiments/placement_pmc_reduce.py $f > $OUT/pmc_${i}_summary.txt && cat $OUT/pmc_${i}_summary.txt | head -40
done
ls -la $OUT
