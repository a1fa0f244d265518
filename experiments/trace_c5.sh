# timeline of ONE C5 step (2-hop sample -> append_unique -> feature gather): everything between two feature-gather launches
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr5
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr5 -- python $GRAFT_REPO_ROOT/bench.py --op sample_gather --steps 6 --warmup 2 --stability-steps 0 ${C5_FLOW:+--c5-flow $C5_FLOW} > /dev/null 2>&1
f=$(find /tmp/tr5 -name "*kernel_trace.csv" | head -1)
m=$(find /tmp/tr5 -name "*memory_copy_trace.csv" | head -1)
python3 - $f $m <<'PY'
import csv, sys
rows = [dict(r, kind='K') for r in csv.DictReader(open(sys.argv[1]))]
if len(sys.argv) > 2 and sys.argv[2]:
    try:
        for r in csv.DictReader(open(sys.argv[2])):
            rows.append({'Kernel_Name': 'COPY ' + r.get('Direction', '') + ' ' + r.get('Bytes', r.get('Size', '')), 'Start_Timestamp': r['Start_Timestamp'], 'End_Timestamp': r['End_Timestamp'], 'kind': 'C'})
    except Exception as e:
        print('no copy trace', e)
rows.sort(key=lambda r: int(r['Start_Timestamp']))
g = [i for i, r in enumerate(rows) if 'rows_copy16_fast_kernel<int' in r['Kernel_Name'] or 'rows_batch_kernel<int' in r['Kernel_Name']]
a, b = g[-3], g[-2]
sel = rows[a:b + 1]
t0 = int(sel[0]['End_Timestamp'])
prev_end = t0
busy = 0
for r in sel[1:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('void ', '').replace('wm::(anonymous namespace)::', '').replace('rocprim::ROCPRIM_400200_NS::detail::', 'rp::')
    print('gap %6.1f  run %7.1f  at %8.1f us  %s' % ((s - prev_end) / 1e3, (e - s) / 1e3, (s - t0) / 1e3, name[:110]))
    busy += (e - s) / 1e3
    prev_end = max(prev_end, e)
print('step period %.1f us, busy %.1f us, %d entries' % ((int(sel[-1]['End_Timestamp']) - t0) / 1e3, busy, len(sel) - 1))
PY
