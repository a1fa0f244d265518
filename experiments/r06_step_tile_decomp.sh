# review item 5 (round 6): step_tile_kernel on four batches in ONE process (experiments/step_tile_decomp.py): whole-call times,
# the kernel's own duration from rocprofv3 --kernel-trace, then the translation / memory-side counters per variant
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_step_tile_decomposition.txt
: > $OUT
echo "== whole call, HIP events, 20 steps per variant and round, one process (plain allocations) ==" >> $OUT
python $R/experiments/step_tile_decomp.py 20 3 2>/dev/null | grep -E "^(plan|round)" >> $OUT
echo "== kernel durations (rocprofv3 --kernel-trace), one process, 10 timed calls per variant and round ==" >> $OUT
d=/tmp/dec_trace; rm -rf $d
rocprofv3 --kernel-trace --output-format csv -d $d -- python $R/experiments/step_tile_decomp.py 10 2 > /dev/null 2>&1
python3 $R/experiments/step_tile_decomp_parse.py trace $(find $d -name "*kernel_trace.csv" | head -1) 10 2 >> $OUT
echo "== counters of step_tile_kernel per launch (one group per pass, one process per pass; events-only synchronisation under --pmc) ==" >> $OUT
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE WRITE_SIZE" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"; do
  d=/tmp/decpmc_$(echo $grp | tr ' ' '_' | cut -c1-40); rm -rf $d
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python $R/experiments/step_tile_decomp.py 3 1 > /dev/null 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/experiments/step_tile_decomp_parse.py pmc "$f" 3 1 >> $OUT
done
cat $OUT
