#!/bin/bash
# First contact with a multi-GPU node: everything rounds 1-5 could only run on ONE GPU, in the order in which a failure is
# cheapest to understand, stopping at the first one. Writes one JSON (first_contact.json) with DESIGN.md section 4's
# predictions beside the measurements.
#
#   bash scripts/first_contact.sh [N]            N = ranks (default: the visible devices); needs N GPUs, RCCL over xGMI
#   FIRST_CONTACT_DRY=1 bash scripts/first_contact.sh 2
#                                                 dry run: toy sizes, collectives over gloo, the ranks share whatever GPU is
#                                                 there (tests/test_bench_contract.py runs this) — checks the script, not the links
# Steps (reference loop: cpp/bench/wholememory_ops/gather_scatter_bench.cu:257-392):
#   1 tests/test_rccl_transport_gpu.py::test_rccl_multi_gpu[N]     the multi-rank scenarios of tests/_dist_worker.py over RCCL
#   2 bench.py --gpus k, k = 2, 4, ... N: C3 uniform, C3 Zipf hashed, C3 Zipf clustered (SURVEY section 8d's three variants)
#   3 the CHUNKED table at N ranks both ways: direct peer loads (hipIpc mappings) vs WM_MAPPED_VIA_EXCHANGE=1 (RCCL route)
#   4 C4: --op grad_apply --dtype f16 --dim 256, CONTINUOUS table, Zipf ids
#   5 C5: --op sample_gather
#   6 rocprofv3 kernel stats of the N-rank uniform run (rank 0)
set -u
# the host driver of these nodes supports dmabuf IPC only: without this RCCL and hipIpc fail with "hipIpcGetMemHandle: invalid argument"
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
N=${1:-$(python3 -c "import torch; print(torch.cuda.device_count())")}
DRY=${FIRST_CONTACT_DRY:-0}
OUT=${FIRST_CONTACT_OUT:-$ROOT/gpurun_out/first_contact}
mkdir -p "$OUT"
if [ "$DRY" = "1" ]; then
  SIZES="--backend gloo --rows 200003 --indices 60000 --steps 2 --warmup 1 --stability-steps 0"
  C5SIZES="--backend gloo --nodes 500003 --avg-degree 10 --seeds 256 --fanouts 8,4 --steps 2 --warmup 1 --stability-steps 0"
  TMO=600
else
  SIZES=""
  C5SIZES=""
  TMO=1800
fi
fail() { echo "first_contact: FAILED at step '$1' (see $OUT/$1.err)"; python3 scripts/first_contact_report.py "$OUT" "$N" "$DRY" "$1"; exit 1; }
run() {   # run <name> <env assignments or ''> <bench args...>: one bench line -> $OUT/<name>.json
  local name=$1; local envs=$2; shift 2
  echo "== $name: $envs python bench.py $*"
  env $envs timeout $TMO python bench.py "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" || fail "$name"
  python3 -c "import json,sys; r=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('   ms_per_step', r['ms_per_step'], 'value', r['value'], r['unit'], 'rccl_ranks', r.get('rccl_ranks'))" || fail "$name"
}

# 1 the scenarios over RCCL
if [ "$DRY" != "1" ]; then
  echo "== rccl_scenarios: test_rccl_multi_gpu[$N]"
  timeout $TMO python -m pytest "tests/test_rccl_transport_gpu.py::test_rccl_multi_gpu[$N]" -x -q -m gpu > "$OUT/rccl_scenarios.log" 2> "$OUT/rccl_scenarios.err" || fail rccl_scenarios
  tail -1 "$OUT/rccl_scenarios.log"
fi
# 2 the scaling curve, three id distributions
k=2
while [ $k -le $N ]; do
  run c3_uniform_n$k "" --gpus $k --no-cpu-baseline $SIZES
  run c3_zipf_n$k "" --gpus $k --dist zipf --no-cpu-baseline $SIZES
  run c3_zipf_clustered_n$k "" --gpus $k --dist zipf_clustered --no-cpu-baseline $SIZES
  k=$((k * 2))
done
# 3 mapped table, both routes
run chunked_direct_n$N "" --gpus $N --memory-type chunked --no-cpu-baseline $SIZES
run chunked_via_exchange_n$N "WM_MAPPED_VIA_EXCHANGE=1" --gpus $N --memory-type chunked --no-cpu-baseline $SIZES
# 4 C4
#   f16 x 256 on a CONTINUOUS table (C4 as BASELINE names it; the 16-bit fold is free of the reference's order): with the
#   sender-side combination of duplicate gradient rows (the default decision from the duplicate estimate) and with every copy
#   shipped; fp32 x 128 DISTRIBUTED in the reference's order (every copy travels, the hot id's owner folds an ordered chain) and
#   with grad_fold = tree (combined). Predictions: first_contact_report.py: predicted_zipf, DESIGN.md section 4.
run c4_grad_apply_f16_n$N "" --gpus $N --op grad_apply --dtype f16 --dim 256 --memory-type continuous --dist zipf --no-cpu-baseline $SIZES
run c4_grad_apply_f16_every_copy_n$N "WM_GRAD_COMBINE=0" --gpus $N --op grad_apply --dtype f16 --dim 256 --memory-type continuous --dist zipf --no-cpu-baseline $SIZES
run c4_grad_apply_f32_n$N "" --gpus $N --op grad_apply --dim 128 --memory-type distributed --dist zipf --no-cpu-baseline $SIZES
run c4_grad_apply_f32_tree_n$N "WM_GRAD_FOLD=tree" --gpus $N --op grad_apply --dim 128 --memory-type distributed --dist zipf --no-cpu-baseline $SIZES
# 5 C5
run c5_sample_gather_n$N "" --gpus $N --op sample_gather $C5SIZES
# 6 kernel stats of the N-rank uniform run
if [ "$DRY" != "1" ] && command -v rocprofv3 > /dev/null; then
  echo "== rocprofv3 kernel stats, $N ranks, uniform"
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fc_prof && timeout $TMO rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fc_prof -- \
      python "$ROOT/bench.py" --gpus $N --no-cpu-baseline --steps 10 --stability-steps 0 > "$OUT/c3_uniform_n${N}_under_rocprof.json" 2> /dev/null )
  f=$(find /tmp/fc_prof -name "*kernel_stats.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/c3_uniform_n${N}_kernel_stats.csv"
fi
python3 scripts/first_contact_report.py "$OUT" "$N" "$DRY" "" || exit 1
echo "first_contact: all steps passed -> $OUT/first_contact.json"
