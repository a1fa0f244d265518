#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in "" ku8; do
  echo "== ${v:-product}"
  WHOLEGRAPH_AMD_VARIANT=$v python experiments/cast_sweep.py 2>&1 | grep -E "^gather" | grep -E "float16 -> float32 dim (128|256)|float32 -> float16|bfloat16" | cut -c1-100
done; done
