"""Drop-in check at the SOURCE level: the reference's own Cython binding
(python/pylibwholegraph/pylibwholegraph/binding/wholememory_binding.pyx — every ``cdef extern from "wholememory/…h"``
block of it) is cythonized against THIS repo's ``include/`` and linked against ``libwholegraph.so``; the extension then
imports (so every symbol the binding references resolves) and its host-only entry points behave.

Runs only where the reference tree and Cython exist (the build container). Nothing of the reference is copied into
the repository: the generated C++ and the extension live under ``build/`` (git-ignored and not shipped to the GPU box).
"""
import os
import shutil
import subprocess
import sys
import sysconfig
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PYX = "/root/reference/python/pylibwholegraph/pylibwholegraph/binding/wholememory_binding.pyx"
OUT = os.path.join(ROOT, "build", "refbind")

pytestmark = pytest.mark.skipif(not os.path.exists(PYX) or shutil.which("cython") is None,
                                reason="needs the reference tree and cython (build container only)")


@pytest.fixture(scope="module")
def reference_binding(wm_lib):
    os.makedirs(OUT, exist_ok=True)
    cpp = os.path.join(OUT, "wholememory_binding.cpp")
    ext = os.path.join(OUT, "wholememory_binding" + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.check_call(["cython", "--cplus", "-3", "-I", os.path.join(ROOT, "include"), PYX, "-o", cpp],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call(["g++", "-O0", "-std=c++17", "-fPIC", "-shared", "-w", "-I", os.path.join(ROOT, "include"),
                           "-I", sysconfig.get_paths()["include"], cpp, "-o", ext,
                           "-L", os.path.join(ROOT, "wholegraph_amd"), "-lwholegraph",
                           "-Wl,-rpath," + os.path.join(ROOT, "wholegraph_amd")])
    return ext


def _run(code):
    """in a fresh interpreter: torch first (so the HIP runtime is shared), then the product library, then the binding"""
    prolog = textwrap.dedent("""
        import sys, ctypes
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import torch, numpy as np
        ctypes.CDLL(%r, mode=ctypes.RTLD_GLOBAL)
        import wholememory_binding as wmb
        import oracle
    """) % (OUT, ROOT, os.path.join(ROOT, "wholegraph_amd", "libwholegraph.so"))
    out = subprocess.run([sys.executable, "-c", prolog + textwrap.dedent(code)], capture_output=True, timeout=300)
    assert out.returncode == 0, out.stdout.decode()[-2000:] + out.stderr.decode()[-3000:]
    return out.stdout.decode()


def test_reference_cython_binding_compiles_links_and_imports(reference_binding):
    out = _run("""
        names = [n for n in dir(wmb) if not n.startswith("_")]
        for n in ("init", "create_communicator", "malloc", "create_wholememory_tensor", "create_embedding",
                  "EmbeddingGatherForward", "EmbeddingGatherGradientApply", "csr_unweighted_sample_without_replacement",
                  "csr_weighted_sample_without_replacement", "append_unique", "add_csr_self_loop",
                  "load_wholememory_handle_from_filelist", "store_wholememory_handle_to_file", "create_cache_policy"):
            assert n in names, n
        print("OK", len(names))
    """)
    assert "OK" in out


def test_reference_binding_host_entry_points(reference_binding):
    """Host-only calls through the reference binding land in this library and agree with the oracle."""
    _run("""
        # enum values and the by-value tensor description struct
        d = wmb.PyWholeMemoryTensorDescription()
        d.set_dtype(wmb.WholeMemoryDataType.DtFloat); d.set_shape((10, 4)); d.set_stride((4, 1)); d.set_storage_offset(0)
        assert d.dim() == 2 and tuple(d.shape) == (10, 4) and tuple(d.stride()) == (4, 1)
        for n, w in [(1003, 1), (1003, 3), (8, 8), (7, 8), (10 ** 9, 8)]:
            sizes, _ = oracle.equal_partition(n, w)
            assert wmb.equal_partition_plan(n, w) == int(sizes[0]), (n, w)
        # error codes travel as the reference's exceptions: no GPU here -> wholememory_init reports CUDA_ERROR
        try:
            wmb.init(0)
            raise SystemExit("init succeeded without a GPU")
        except RuntimeError as e:
            assert "CUDA" in str(e)
        # a torch CPU tensor wrapped the way the reference's wholegraph_env.wrap_torch_tensor does it, filled by the
        # library's host random helper, compared with the oracle's generator
        for dt, wdt, npdt in [(torch.int32, wmb.WholeMemoryDataType.DtInt, np.int32), (torch.int64, wmb.WholeMemoryDataType.DtInt64, np.int64)]:
            t = torch.zeros(64, dtype=dt)
            desc = wmb.PyWholeMemoryTensorDescription()
            desc.set_dtype(wdt); desc.set_storage_offset(0); desc.set_shape(tuple(t.shape)); desc.set_stride(tuple(t.stride()))
            w = wmb.WrappedLocalTensor().wrap_tensor(desc, t.data_ptr())
            wmb.host_generate_random_positive_int(12345, 7, w)
            assert np.array_equal(t.numpy(), oracle.random_positive_int(12345, 7, 64, npdt))
        print("OK")
    """)
