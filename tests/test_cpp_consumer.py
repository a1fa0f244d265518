"""tools/gather_scatter_bench: a torch-free C++ program over include/wholememory/*.h + libwholegraph.so (the counterpart of
the reference's cpp/bench/wholememory_ops/gather_scatter_bench.cu). Building it is part of `make`; here it is run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "gather_scatter_bench")


def test_tool_is_built_and_prints_usage(wm_lib):
    assert os.path.exists(TOOL), "tools/gather_scatter_bench is built by wholegraph_amd/csrc/Makefile (`all`)"
    p = subprocess.run([TOOL, "--help"], capture_output=True, timeout=60)
    assert p.returncode == 2 and b"--embedding_table_size" in p.stderr
    # links against the product library and nothing of torch
    deps = subprocess.run(["ldd", TOOL], capture_output=True, timeout=60).stdout.decode()
    assert "libwholegraph.so" in deps and "torch" not in deps


@pytest.mark.gpu
@pytest.mark.parametrize("args", [
    ["-t", "chunked", "-l", "device", "-e", str(1 << 30), "-g", str(64 << 20), "-d", "128", "-c", "5", "-f", "gather"],
    ["-t", "continuous", "-l", "device", "-e", str(1 << 28), "-g", str(16 << 20), "-d", "32", "-c", "3", "-f", "scatter"],
    ["-t", "distributed", "-l", "device", "-e", str(1 << 28), "-g", str(16 << 20), "-d", "64", "-c", "3", "-f", "gather"],
    ["-t", "hierarchy", "-l", "device", "-e", str(1 << 28), "-g", str(16 << 20), "-d", "64", "-c", "3", "-f", "gather"],
    ["-t", "chunked", "-l", "host", "-e", str(1 << 28), "-g", str(16 << 20), "-d", "64", "-c", "3", "-f", "gather"],
])
def test_cpp_bench_runs_and_verifies(wm_lib, args):
    p = subprocess.run([TOOL] + args, capture_output=True, timeout=300)
    out = p.stdout.decode() + p.stderr.decode()
    assert p.returncode == 0 and "Bandwidth:" in out and "verified" in out, out[-2000:]
