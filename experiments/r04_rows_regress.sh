#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gather_scatter_gpu.py tests/test_golden_fixtures_gpu.py tests/test_full_size_gpu.py tests/test_host_sorted_gather_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python experiments/fuzz_rows.py 1500 20260929 2>&1 | tail -3
