#!/bin/bash
# round 4: step_tile_kernel with the straight-line fast path — tests, then the whole gradient-apply call under the occupancy /
# launch-shape switches, interleaved in one process
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_exchange_optim_gpu.py tests/test_full_size_gpu.py tests/test_embedding_cache_gpu.py tests/test_golden_fixtures_gpu.py -m gpu -x -q 2>&1 | tail -3
O=gpurun_out/r04_grad_ab.txt
: > $O
S="default:;occ5:WM_TILE_OCC=5;inorder:WM_TILE_INORDER=1;inorder_occ5:WM_TILE_INORDER=1,WM_TILE_OCC=5"
timeout 600 python experiments/grad_env_ab.py sgd uniform 128 f32 "$S" 2>&1 | grep -v amdgpu.ids >> $O
timeout 600 python experiments/grad_env_ab.py sgd zipf 128 f32 "$S;tree:WM_GRAD_FOLD=tree" 2>&1 | grep -v amdgpu.ids >> $O
timeout 600 python experiments/grad_env_ab.py adam uniform 128 f32 "default:;inorder:WM_TILE_INORDER=1" 2>&1 | grep -v amdgpu.ids >> $O
timeout 600 python experiments/grad_env_ab.py sgd uniform 256 f16 "default:;inorder:WM_TILE_INORDER=1" 2>&1 | grep -v amdgpu.ids >> $O
cat $O
