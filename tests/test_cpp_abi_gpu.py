"""A C++ program that uses only <wholememory/*.h> + libwholegraph.so (no Python on the data path), the way the
reference's gtests / bench call the library. Built with hipcc against the in-tree library and run on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "c_abi_gather_test.cpp")
EXE = os.path.join(ROOT, "build", "c_abi_gather_test")


def build_exe():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    libdir = os.path.join(ROOT, "wholegraph_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE,
                           "-L", libdir, "-lwholegraph", "-Wl,-rpath," + libdir])


def test_cpp_caller_compiles_and_links(wm_lib):
    """CPU box: the headers are consumable from C++ and every used symbol resolves."""
    build_exe()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_caller_runs(wm_lib):
    build_exe()
    out = subprocess.run([EXE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert out.returncode == 0 and b"C ABI GATHER OK" in out.stdout, out.stdout.decode(errors="replace")[-3000:]
