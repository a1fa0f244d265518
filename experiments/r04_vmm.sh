#!/bin/bash
cd $GRAFT_REPO_ROOT/experiments
for m in 12 13 14 8 9 6; do timeout 600 ./vmm_cycle2 $m 1000; done
