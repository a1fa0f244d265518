# kernel breakdown of a cached HOST-table gather (C1 shape with a device row cache), Zipf ids
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -- python $GRAFT_REPO_ROOT/bench.py --location cpu --rows 10000000 --dim 64 --indices 1000000 --dist ${DIST:-zipf} --cache-ratio ${RATIO:-0.1} --steps 20 --no-cpu-baseline --no-check 2>/dev/null | tail -1 | cut -c1-250
python3 - $(find /tmp/pc -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r['Name'][:90], r['Calls'], 'avg', round(float(r['AverageNs'])/1e3, 1), 'min', round(float(r['MinNs'])/1e3, 1), 'max', round(float(r['MaxNs'])/1e3, 1), 'us')
PY
