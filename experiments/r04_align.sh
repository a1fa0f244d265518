#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_exchange_optim_gpu.py -m gpu -x -q -k "row_align or training_flow or save_load" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gather_scatter_gpu.py tests/test_golden_fixtures_gpu.py -m gpu -x -q 2>&1 | tail -3
python experiments/span_ab.py 132 200 500 1000 2>&1 | grep "B rows" | cut -c1-200
