cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pu -- python $GRAFT_REPO_ROOT/bench.py --op grad_apply --dist ${DIST:-uniform} --steps 10 --no-cpu-baseline 2>/dev/null | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/grad_${DIST:-uniform}_bench.json
cp $(find /tmp/pu -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/grad_${DIST:-uniform}_kernel_stats.csv
python3 - $GRAFT_REPO_ROOT/gpurun_out/grad_${DIST:-uniform}_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r['Name'][:100], r['Calls'], round(float(r['AverageNs'])/1e3, 1), 'us')
PY
cut -c1-240 $GRAFT_REPO_ROOT/gpurun_out/grad_${DIST:-uniform}_bench.json
