#!/bin/bash
# Build a compile-time VARIANT of the library for an A/B on the GPU box, next to (not instead of) the product build:
#   scripts/build_variant.sh NAME "-DWM_BATCH_RPS2_BPERMUTE=0 ..."      (always with the A/B knobs of knobs.hpp compiled in: -DWM_AB_KNOBS)
# -> experiments/variants/NAME/wholegraph_amd/ = the Python package (symlinked sources) + its own libwholegraph.so and
#    libwg_torch_env.so. Run a script against it with  PYTHONPATH=experiments/variants/NAME python ...
# (experiments/variants/ is git-ignored; it travels to the GPU box with the snapshot.)
set -e
NAME=$1; FLAGS=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
V=$ROOT/experiments/variants/$NAME
mkdir -p $V/wholegraph_amd $V/obj
for f in $ROOT/wholegraph_amd/*.py $ROOT/wholegraph_amd/torch; do ln -sfn $f $V/wholegraph_amd/$(basename $f); done
ln -sfn $ROOT/oracle $V/oracle
make -C $ROOT/wholegraph_amd/csrc -j8 OUT=$V/wholegraph_amd/libwholegraph.so OBJDIR=$V/obj TORCH_ENV=$V/wholegraph_amd/libwg_torch_env.so \
  TOOL=$V/gather_scatter_bench "HIPFLAGS=--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -Wall -Wno-unused-function -ffp-contract=off -DWM_AB_KNOBS $FLAGS" AB=1 \
  $V/wholegraph_amd/libwholegraph.so 2>&1 | grep -E "error|Error" || true
make -C $ROOT/wholegraph_amd/csrc OUT=$V/wholegraph_amd/libwholegraph.so OBJDIR=$V/obj TORCH_ENV=$V/wholegraph_amd/libwg_torch_env.so \
  $V/wholegraph_amd/libwg_torch_env.so 2>&1 | grep -E "error|Error" || true
ls -la $V/wholegraph_amd/*.so
