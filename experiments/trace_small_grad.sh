# timeline of ONE gradient-apply call of a small batch (N gradient rows, default 65536): everything between two step_tile launches
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trg
N=${N:-65536}
cat > /tmp/small_grad.py <<PY
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
emb = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [10_000_000, 128])
wgth.create_wholememory_optimizer(emb, "sgd", {})
idx = torch.randint(0, 10_000_000, ($N,), device="cuda"); g = torch.randn(($N, 128), device="cuda")
for _ in range(12):
    emb.add_gradients(idx, g); emb.need_apply = True; emb.apply_gradients(0.01)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trg -- python /tmp/small_grad.py > /dev/null 2>&1
f=$(find /tmp/trg -name "*kernel_trace.csv" | head -1)
m=$(find /tmp/trg -name "*memory_copy_trace.csv" | head -1)
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
g = [i for i, r in enumerate(rows) if 'step_tile_kernel' in r['Kernel_Name']]
a, b = g[-3], g[-2]
sel = rows[a:b + 1]
t0 = int(sel[0]['End_Timestamp'])
prev_end = t0
busy = 0
for r in sel[1:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('void ', '').replace('wm::(anonymous namespace)::', '').replace('rocprim::ROCPRIM_400200_NS::detail::', 'rp::')
    print('gap %6.1f  run %7.1f  at %8.1f us  %s' % ((s - prev_end) / 1e3, (e - s) / 1e3, (s - t0) / 1e3, name[:100]))
    busy += (e - s) / 1e3
    prev_end = max(prev_end, e)
print('step period %.1f us, busy %.1f us, %d entries' % ((int(sel[-1]['End_Timestamp']) - t0) / 1e3, busy, len(sel) - 1))
PY
