#!/bin/bash
# round 5: the C5 step on a synthetic graph WITH hubs (bench.py --op sample_gather --col-dist powerlaw --col-exponent s: neighbour of
# popularity rank k with probability ~ k^-s) — round 4's table insert (WM_AU_MERGE=0 WM_AU_DIRECT_CAS=0) against the shipped one.
cd "$(dirname "$0")/.."
line() { python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r.get('stability',{}).get('median_ms'), r['frontier_sizes'])"; }
for s in 0.6 0.8 0.9; do for rep in 1 2; do for v in "WM_AU_MERGE=0 WM_AU_DIRECT_CAS=0" "X=default"; do
  echo "exponent $s  $v  ms_per_step, median, frontiers: $(env $v timeout 400 python bench.py --op sample_gather --steps 200 --col-dist powerlaw --col-exponent $s 2>/dev/null | line)"
done; done; done
for seeds in 4096 65536; do for v in "WM_AU_MERGE=0 WM_AU_DIRECT_CAS=0" "X=default"; do
  echo "exponent 0.8  $seeds seeds  $v  ms_per_step, median, frontiers: $(env $v timeout 400 python bench.py --op sample_gather --seeds $seeds --col-dist powerlaw --col-exponent 0.8 2>/dev/null | line)"
done; done
