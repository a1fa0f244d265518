#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03/c5
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_graph_ops_gpu.py tests/test_c5_flow_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -12
for chain in 1 0; do for i in 1 2; do WM_MULTILAYER_CHAIN=$chain python bench.py --op sample_gather --steps 100 --stability-steps 100 > $OUT/sample_gather_chain${chain}_$i.json 2>/dev/null; python -c "
import json; r=json.load(open('$OUT/sample_gather_chain${chain}_$i.json')); print('chain=$chain', r['ms_per_step'], r['stability']['median_ms'], r['roofline']['frac'])"; done; done
python bench.py --op sample_gather --seeds 65536 --steps 20 --stability-steps 0 > $OUT/sample_gather_64k.json 2>/dev/null; python -c "
import json; r=json.load(open('$OUT/sample_gather_64k.json')); print('64k', r['ms_per_step'], r['roofline']['frac'])"
python bench.py --op sample_gather --seeds 16384 --steps 20 --stability-steps 0 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('16k', r['ms_per_step'], r['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_c5 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python $R/bench.py --op sample_gather --steps 50 --stability-steps 0 > $OUT/sample_gather_under_rocprof.json 2>/dev/null
cp $(find /tmp/prof_c5 -name "*kernel_stats.csv" | head -1) $OUT/sample_gather_kernel_stats.csv
python3 - $OUT/sample_gather_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:20]:
    print("%-90s calls %5s avg %8.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3))
PY
