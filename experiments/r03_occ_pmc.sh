#!/bin/bash
# write traffic of step_tile_kernel: the 7-wave build (72 registers, spills to scratch) against the natural 5-wave build
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03/occ
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for occ in 7 5; do for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/pmco_${occ}_$c
  WM_TILE_OCC=$occ timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmco_${occ}_$c -- python $R/bench.py --op grad_apply --no-cpu-baseline --steps 5 --warmup 2 --stability-steps 0 > /dev/null 2>&1
  python3 - $(find /tmp/pmco_${occ}_$c -name "*counter_collection.csv" | head -1) $occ $c <<'PY' | tee -a $OUT/step_tile_traffic_by_occupancy.txt
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "step_tile_kernel" in r["Kernel_Name"]]
v=sum(float(r["Counter_Value"]) for r in rows)/len(rows)
print("WM_TILE_OCC=%s %s: %d launches, %.1f KiB per launch = %.3f GB  (scratch %s B/lane, VGPRs %s)  %s" % (sys.argv[2], sys.argv[3], len(rows), v, v*1024/1e9, rows[0]["Scratch_Size"], rows[0]["VGPR_Count"], rows[0]["Kernel_Name"][40:100]))
PY
done; done
cd $R; timeout 300 python experiments/grad_env_ab.py sgd uniform 128 f32 "occ7:WM_TILE_OCC=7;occ5:WM_TILE_OCC=5" 2>&1 | grep -v amdgpu | tee -a $OUT/step_tile_traffic_by_occupancy.txt
