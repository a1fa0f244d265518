# timeline of ONE gradient-apply call (uniform ids): every kernel / fill / copy between two step_tile_kernel launches, with gaps
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $GRAFT_REPO_ROOT/bench.py --op grad_apply --dist ${DIST:-uniform} --optimizer ${OPT:-sgd} --steps 6 --warmup 2 --stability-steps 0 --no-cpu-baseline > /dev/null 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
rows = sorted((r for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: int(r['Start_Timestamp']))
tiles = [i for i, r in enumerate(rows) if 'step_tile_kernel' in r['Kernel_Name']]
a, b = tiles[-3], tiles[-2]
sel = rows[a:b + 1]
t0 = int(sel[0]['End_Timestamp'])
prev_end = t0
for r in sel[1:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('void ', '').replace('wm::(anonymous namespace)::', '').replace('rocprim::ROCPRIM_400200_NS::detail::', 'rp::')
    print('gap %6.1f  run %7.1f  at %8.1f us  q%s  %s' % ((s - prev_end) / 1e3, (e - s) / 1e3, (s - t0) / 1e3, r.get('Queue_Id', '?'), name[:120]))
    prev_end = max(prev_end, e)
print('call period %.1f us' % ((int(sel[-1]['End_Timestamp']) - t0) / 1e3))
PY
