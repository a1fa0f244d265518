#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_exchange_optim_gpu.py tests/test_golden_fixtures_gpu.py -m gpu -x -q -k "bucket or golden or fixture" 2>&1 | tail -2
WM_BUCKET_DENSE=0 timeout 600 python -m pytest tests/test_exchange_optim_gpu.py -m gpu -x -q -k "bucket" 2>&1 | tail -1
for r in 1 2; do for d in 1 0; do WM_BUCKET_DENSE=$d python experiments/bucket_ab.py 2>&1 | grep bucket_ids | sed "s/^product/dense=$d/"; done; done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bk && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk -- python $GRAFT_REPO_ROOT/experiments/bucket_ab.py > /dev/null 2>&1; python3 - $(find /tmp/bk -name "*kernel_stats.csv" | head -1) <<PY
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:5]:
    print("%-100s calls %5s avg %8.1f us" % (r["Name"].replace("wm::(anonymous namespace)::","")[:100], r["Calls"], float(r["AverageNs"])/1e3))
PY
