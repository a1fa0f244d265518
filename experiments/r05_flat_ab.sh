#!/bin/bash
# round 5: step_flat_kernel against round 4's kernels (WM_STEP_FLAT=0) and forced everywhere (=1), one process per shape,
# the settings interleaved inside it; batch sized to 10 M x 512 B of gradient rows.
cd "$(dirname "$0")/.."
S="round4:WM_STEP_FLAT=0;default:;flat:WM_STEP_FLAT=1"
for spec in "sgd 300" "sgd 602" "sgd 100" "sgd 200" "sgd 1000" "sgd 513" "sgd 128" "sgd 36" "adam 300" "adam 602" "adagrad 200"; do
  set -- $spec
  AB_IDS=$((1280000000 / $2)) AB_REPS=10 timeout 300 python experiments/grad_env_ab.py $1 uniform $2 f32 "$S" 2>&1 | grep -v "^\[" | tail -3
done
