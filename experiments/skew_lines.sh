# DISTRIBUTED gather at N = 1 with the FULL exchange route on one GPU (WM_FORCE_RCCL=1 WM_EXCHANGE_SELF=1: the rank's own
# segment goes bucket -> counts -> RCCL all-to-all-v like a peer's): uniform / zipf / zipf_clustered ids, de-duplication
# decided automatically vs forced off / on. The direct single-rank route (no exchange at all) is the first line of each group.
cd $GRAFT_REPO_ROOT
line() { python bench.py --memory-type distributed --dist $1 --no-cpu-baseline --steps 30 --stability-steps 60 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('%-15s %-28s ms/step %.4f  min %.4f  median %.4f  transport %s' % ('$1', '$2', r['ms_per_step'], r['stability']['min_ms'], r['stability']['median_ms'], r['transport']))"; }
for d in uniform zipf zipf_clustered; do
  line $d "direct (no exchange)"
  export WM_FORCE_RCCL=1 WM_EXCHANGE_SELF=1
  line $d "exchange, dedup auto"
  WM_GATHER_DEDUP=0 line $d "exchange, dedup off"
  WM_GATHER_DEDUP=1 line $d "exchange, dedup on"
  unset WM_FORCE_RCCL WM_EXCHANGE_SELF
done
