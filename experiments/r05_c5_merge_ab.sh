cd $GRAFT_REPO_ROOT
line() { python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r.get('stability',{}).get('median_ms'))"; }
for rep in 1 2 3 4; do for v in "WM_AU_MERGE=0 WM_AU_DIRECT_CAS=0" "WM_AU_MERGE=0 WM_AU_DIRECT_CAS=2" "X=default"; do
  echo "1024 seeds  $v  ms_per_step, median: $(env $v timeout 300 python bench.py --op sample_gather --steps 200 2>/dev/null | line)"
done; done
for v in "WM_AU_MERGE=0 WM_AU_DIRECT_CAS=0" "X=default"; do
  echo "65536 seeds  $v  ms_per_step, median: $(env $v timeout 300 python bench.py --op sample_gather --seeds 65536 2>/dev/null | line)"
done
timeout 300 python experiments/r05_au_skew.py 2>&1 | grep -v "^\[\|amdgpu.ids"
