#!/bin/bash
# gradient apply across the shapes the configs name: whole call ms and fraction of 8 TB/s (roofline.frac of the line)
cd $GRAFT_REPO_ROOT
line() { python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); ro=r.get('roofline',{}); print('%.4f ms  frac %.3f  achieved %.0f %s' % (r['ms_per_step'], ro.get('frac',0), ro.get('achieved',0), ro.get('unit','')))"; }
run() { echo -n "$*: "; python bench.py --op grad_apply --no-cpu-baseline --stability-steps 0 --steps 30 "$@" 2>/dev/null | line; }
run --dtype f32 --dim 128
run --dtype f16 --dim 256 --rows 125000000
run --dtype f16 --dim 256 --rows 125000000 --memory-type continuous
run --dtype f16 --dim 256 --rows 125000000 --dist zipf
run --dtype bf16 --dim 256 --rows 125000000
run --dtype f32 --dim 128 --rows 125000000 --memory-type distributed
run --dtype f32 --dim 128 --optimizer adam
run --dtype f32 --dim 128 --optimizer adagrad
run --dtype f32 --dim 128 --optimizer rmsprop
run --dtype f32 --dim 64
run --dtype f32 --dim 256
run --dtype f32 --dim 100
run --dtype f32 --dim 128 --indices 1000000
run --dtype f32 --dim 128 --indices 100000
run --dtype f32 --dim 128 --indices 30000000
