#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "" noalign; do
  echo "== ${v:-product}"
  WHOLEGRAPH_AMD_VARIANT=$v python experiments/span_ab.py 25 33 50 65 100 129 130 258 513 602 1030 2>&1 | grep "^gather" | cut -c1-75
done; done
