cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gather_scatter_gpu.py -x -q -m gpu 2>&1 | tail -8
WM_ROWS_FLAT=1 timeout 900 python -m pytest tests/test_gather_scatter_gpu.py tests/test_exchange_optim_gpu.py -x -q -m gpu 2>&1 | tail -8
timeout 600 python experiments/dim_sweep.py 50 100 127 200 256 300 301 602 2>&1 | grep -v amdgpu.ids
