#!/bin/bash
# round 4: the restructured row kernels over the row shapes and casts of the round-3 sweeps + the single-batch kernel as a
# plain streaming copy (sequential ids): the ceiling of its launch shape
cd ${GRAFT_REPO_ROOT:-.}
for d in sequential uniform sequential uniform; do
  timeout 600 python bench.py --dist $d --no-cpu-baseline --steps 100 --stability-steps 0 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}
print('gather $d ids: ms_per_step %.4f kernel_ms %s frac %s' % (d['ms_per_step'], r.get('kernel_ms'), r.get('frac')))" >> gpurun_out/r04_sequential_ceiling.txt
done
cat gpurun_out/r04_sequential_ceiling.txt
timeout 1500 python experiments/dim_sweep.py --csv=gpurun_out/r04_dim_sweep.csv 4 8 16 25 32 33 50 64 65 100 128 129 130 200 256 258 300 512 513 602 1000 1024 1030 > gpurun_out/r04_dim_sweep.txt 2>&1
tail -50 gpurun_out/r04_dim_sweep.txt
timeout 900 python experiments/cast_sweep.py > gpurun_out/r04_cast_sweep.txt 2>&1
cat gpurun_out/r04_cast_sweep.txt
