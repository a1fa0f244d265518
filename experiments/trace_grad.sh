# do step_long4_kernel and step_short_kernel overlap? start/end timestamps of the last gradient-apply call
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $GRAFT_REPO_ROOT/bench.py --op grad_apply --dist zipf --optimizer ${OPT:-sgd} --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
sel = [r for r in rows if any(k in r['Kernel_Name'] for k in ('step_', 'mark_long', 'remap_self', 'compact_runs'))]
sel = sel[-5:]
t0 = min(int(r['Start_Timestamp']) for r in sel)
for r in sel:
    print('%-40s start %8.1f us  end %8.1f us  queue %s' % (r['Kernel_Name'][:40].replace('void wm::(anonymous namespace)::', ''), (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3, r.get('Queue_Id', '?')))
PY
