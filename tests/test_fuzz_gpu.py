"""Randomised gather / scatter shapes against torch indexing on the GPU (experiments/fuzz_rows.py): dtype pairs with casts,
dims 1..700, padded strides, column-offset views, strided outputs, int32 / int64 ids with negatives and duplicates, the
three memory types — complements the fixed reference parameter sets of test_gather_scatter_gpu.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_random_row_shapes_match_torch(wm_lib, seed):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "experiments", "fuzz_rows.py"), "250", str(seed)],
                       capture_output=True, timeout=900)
    out = p.stdout.decode() + p.stderr.decode()
    assert p.returncode == 0 and "cases 250, failures 0" in out, out[-3000:]


@pytest.mark.gpu
def test_random_dedup_apply_cases_match_the_oracle(wm_lib):
    """experiments/fuzz_optim.py: optimizer kinds x dims x strides x run-length mixes (in-wave folds, the LDS-DMA long-run
    kernel across tile and order-chunk boundaries, its fallback), two steps each, bit-exact against the oracle."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "experiments", "fuzz_optim.py"), "120", "17"],
                       capture_output=True, timeout=900)
    out = p.stdout.decode() + p.stderr.decode()
    assert p.returncode == 0 and "cases 120, failures 0" in out, out[-3000:]


@pytest.mark.gpu
def test_random_op_sequences_keep_the_row_cache_transparent(wm_lib):
    """experiments/fuzz_cache.py: gathers / training steps / write-backs / drops on a cached HOST embedding against an
    uncached twin, outputs, tables and optimizer states equal bit for bit."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "experiments", "fuzz_cache.py"), "40", "5"],
                       capture_output=True, timeout=900)
    out = p.stdout.decode() + p.stderr.decode()
    assert p.returncode == 0 and "sequences 40, failures 0" in out, out[-3000:]


@pytest.mark.gpu
def test_random_graphs_sample_like_the_oracle(wm_lib):
    """experiments/fuzz_sample.py: random CSR graphs / fan-outs / seeds / dtypes / placements — unweighted and weighted
    one-hop sampling and append_unique bit for bit against the oracle."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "experiments", "fuzz_sample.py"), "150", "9"],
                       capture_output=True, timeout=900)
    out = p.stdout.decode() + p.stderr.decode()
    assert p.returncode == 0 and "cases 150, failures 0" in out, out[-3000:]
