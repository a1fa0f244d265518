# gradient apply, whole call: optimizer x id distribution (+ the fp16 x 256 shape), REPS times each
for i in $(seq 1 ${REPS:-2}); do
for o in sgd adam; do for d in uniform zipf; do python bench.py --op grad_apply --dist $d --optimizer $o --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('$o $d', r['ms_per_step'])"; done; done
python bench.py --op grad_apply --dist zipf --dtype f16 --dim 256 --rows 50000000 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('f16x256 zipf', r['ms_per_step'])"
done
