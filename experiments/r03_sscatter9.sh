#!/bin/bash
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py > gpurun_out/r03/bench_after_staged.json 2>/dev/null; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench_after_staged.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel'))
print(json.dumps(d['cpu_baseline'])[:900])
PY
WM_ROWS_STAGED_MINROW=1000000 python bench.py --steps 5 --stability-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('minrow=1M c1:', json.dumps(d['cpu_baseline'].get('c1_shape'))[:600])"
