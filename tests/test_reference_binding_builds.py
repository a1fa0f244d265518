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


def _run(code, env=None):
    """in a fresh interpreter: torch first (so the HIP runtime is shared), then the product library, then the binding"""
    prolog = textwrap.dedent("""
        import sys, ctypes
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import torch, numpy as np
        ctypes.CDLL(%r, mode=ctypes.RTLD_GLOBAL)
        import wholememory_binding as wmb
        import oracle
    """) % (OUT, ROOT, os.path.join(ROOT, "wholegraph_amd", "libwholegraph.so"))
    out = subprocess.run([sys.executable, "-c", prolog + textwrap.dedent(code)], capture_output=True, timeout=300,
                         env=dict(os.environ, **(env or {})))
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


def test_gather_scatter_and_env_functions_through_the_reference_binding(reference_binding):
    """The whole call chain a user of the reference exercises — reference binding objects, its Python-callback env
    functions (`GlobalContextWrapper`: temporary and output allocators called back from inside the library),
    `wholememory_scatter_op` / `wholememory_gather_op` / `wholememory_env_test_cython_op` — against this library. There
    is no GPU here, so the device seam is served by the CPU test backend (oracle/test_backend.cpp, tests only); the
    orchestration, the ABI structs and the callbacks are the product's."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "test_backend"], stdout=subprocess.DEVNULL)
    _run("""
        import os
        from wholegraph_amd import binding as mine
        tb = ctypes.CDLL(os.path.join(%r, "oracle", "libwm_test_backend.so"))
        tb.wm_test_backend.restype = ctypes.c_void_p
        mine.check(mine.lib().wm_testing_install_backend(ctypes.c_void_p(tb.wm_test_backend())))
        wmb.init(0)
        comm = wmb.create_communicator(wmb.PyWholeMemoryUniqueID(), 0, 1)
        assert comm.get_rank() == 0 and comm.get_size() == 1

        DT = {torch.float32: wmb.WholeMemoryDataType.DtFloat, torch.int64: wmb.WholeMemoryDataType.DtInt64,
              torch.int32: wmb.WholeMemoryDataType.DtInt}
        TD = {int(v): k for k, v in DT.items()}

        def wrap(t):
            d = wmb.PyWholeMemoryTensorDescription()
            d.set_dtype(DT[t.dtype]); d.set_storage_offset(0); d.set_shape(tuple(t.shape)); d.set_stride(tuple(t.stride()))
            return wmb.WrappedLocalTensor().wrap_tensor(d, t.data_ptr())

        class Slot(object):          # one allocation slot, as the reference's TorchMemoryContext
            tensor = None
        calls = {"create": 0, "destroy": 0, "malloc": 0, "free": 0, "out_malloc": 0}

        def t_create(glob):
            calls["create"] += 1
            return Slot()
        def t_destroy(slot, glob):
            calls["destroy"] += 1
        def t_malloc(desc, alloc_type, slot, glob):
            calls["malloc"] += 1
            slot.tensor = torch.empty(tuple(desc.shape), dtype=TD[int(desc.dtype)])   # "device" memory is host memory here
            return slot.tensor.data_ptr()
        def t_free(slot, glob):
            calls["free"] += 1
            slot.tensor = None
        def o_malloc(desc, alloc_type, slot, glob):
            calls["out_malloc"] += 1
            slot.tensor = torch.empty(tuple(desc.shape), dtype=TD[int(desc.dtype)])
            slot.kind = alloc_type.get_type()
            return slot.tensor.data_ptr()
        ctx = wmb.GlobalContextWrapper()
        # (the reference's callbacks INCREF the global contexts unconditionally: they must be real, truthy objects)
        glob = {"owner": "test"}
        ctx.create_context(t_create, t_destroy, t_malloc, t_free, glob, o_malloc, t_free, glob)
        env = ctx.get_env_fns()

        rows, dim = 5003, 24
        table = wmb.create_wholememory_matrix(wmb.WholeMemoryDataType.DtFloat, rows, dim, -1, comm,
                                              wmb.WholeMemoryMemoryType.MtDistributed, wmb.WholeMemoryMemoryLocation.MlDevice)
        assert tuple(table.shape) == (rows, dim) and table.get_local_entry_count() == rows
        full = oracle.fill_closed_form(np.float32, 0, rows, dim)
        ids = torch.arange(rows, dtype=torch.int64)
        wmb.wholememory_scatter_op(wrap(torch.from_numpy(full)), wrap(ids), table, env, 0)
        rng = np.random.default_rng(0)
        idx = rng.integers(0, rows, 4000).astype(np.int32)
        idx[::13] = -1
        out = torch.full((4000, dim), -3.0)
        wmb.wholememory_gather_op(table, wrap(torch.from_numpy(idx)), wrap(out), env, 0)
        want = np.full((4000, dim), -3.0, np.float32)
        want[idx >= 0] = full[idx[idx >= 0]]
        assert out.numpy().tobytes() == want.tobytes()

        # the env self-test op of the reference binding: fixed output + device / pinned / host outputs via output_fns
        inp = torch.arange(9, dtype=torch.float32)
        fixed = torch.zeros((5, 9))
        slots = [Slot(), Slot(), Slot()]
        wmb.wholememory_env_test_cython_op(wrap(inp), wrap(fixed), id(slots[0]), id(slots[1]), id(slots[2]), 5, env, 0)
        want = torch.arange(5, dtype=torch.float32).unsqueeze(1) + inp.unsqueeze(0)
        assert torch.equal(fixed, want) and calls["out_malloc"] == 3
        # the op's scratch buffer came from the temporary allocator: context created, filled, freed, destroyed
        assert calls["create"] >= 1 and calls["create"] == calls["destroy"] and calls["malloc"] >= 1 and calls["free"] >= 1
        for s in slots:
            assert torch.equal(s.tensor, want)
        assert [s.kind for s in slots] == [int(wmb.WholeMemoryMemoryAllocType.MatDevice), int(wmb.WholeMemoryMemoryAllocType.MatPinned),
                                           int(wmb.WholeMemoryMemoryAllocType.MatHost)]
        wmb.destroy_wholememory_tensor(table)
        wmb.destroy_communicator(comm)
        wmb.finalize()
        print("OK")
    """ % ROOT, env={"WHOLEGRAPH_AMD_TESTING": "1"})
