"""The C-ABI library loads on a CPU-only box and exports every symbol include/wholememory/*.h declares
(no compute calls here). Also: the product refuses to initialise without a GPU and the testing seam
refuses to act outside WHOLEGRAPH_AMD_TESTING=1."""
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "wholememory", "*.h")):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        text = re.sub(r"#define[^\n]*(\\\n[^\n]*)*", "", text)
        # drop function-pointer typedefs and struct members
        text = re.sub(r"typedef[^;]*;", "", text)
        text = re.sub(r"struct\s+\w+\s*\{.*?\};", "", text, flags=re.S)
        for m in re.finditer(r"\b([A-Za-z_]\w*)\s*\(", text):
            n = m.group(1)
            if n.startswith(("wholememory_", "wm_testing_", "wholegraph_csr_", "generate_")) or n in (
                    "get_device_prop", "fork_get_device_count", "get_wholememory_tensor_count", "graph_append_unique",
                    "csr_add_self_loop"):
                names.add(n)
    return names


def test_every_declared_symbol_is_exported_and_bound(wm_lib):
    from wholegraph_amd import binding
    declared = declared_functions()
    assert len(declared) > 90
    out = subprocess.check_output(["nm", "-D", "--defined-only", binding.LIB_PATH]).decode()
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = sorted(declared - exported)
    assert not missing, "declared in include/ but not exported: %s" % missing
    unbound = sorted(declared - set(binding.PROTOTYPES))
    assert not unbound, "declared in include/ but missing from binding.PROTOTYPES: %s" % unbound
    stale = sorted(set(binding.PROTOTYPES) - declared)
    assert not stale, "bound but not declared in include/: %s" % stale
    for name in declared:
        assert getattr(wm_lib, name) is not None
    # ... and nothing else leaves the library (csrc/exports.map): no C++ internals a second copy of the sources could
    # interpose on, no stray globals
    every = {line.split()[-1] for line in out.splitlines() if line.split()[-2] in "TDBRVW"}
    mangled = sorted(n for n in every if n.startswith("_Z"))
    assert not mangled, "C++ symbols exported: %s ..." % mangled[:5]
    extra = sorted(every - declared)
    assert not extra, "exported but not declared in include/: %s" % extra


def test_struct_layouts_match_the_c_abi():
    """sizeof / offsets of the by-value structs (compiled C probe vs ctypes)."""
    import ctypes as C
    import tempfile
    from wholegraph_amd import binding as B
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include <wholememory/wholegraph_amd_ext.h>
#include <wholememory/wholegraph_op.h>
#include <wholememory/graph_op.h>
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu ", sizeof(struct wholememory_tensor_description_t),
         offsetof(struct wholememory_tensor_description_t, strides), offsetof(struct wholememory_tensor_description_t, storage_offset),
         offsetof(struct wholememory_tensor_description_t, dim), offsetof(struct wholememory_tensor_description_t, dtype),
         sizeof(struct wholememory_matrix_description_t), sizeof(struct wholememory_array_description_t));
  printf("%zu %zu %zu %zu ", sizeof(struct wholememory_gref_t), offsetof(struct wholememory_gref_t, stride),
         offsetof(struct wholememory_gref_t, same_chunk), sizeof(struct wholememory_env_func_t));
  printf("%zu %zu\n", sizeof(struct wholememory_unique_id_t), sizeof(struct wm_ext_collectives_t));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as td:
        c, exe = os.path.join(td, "p.c"), os.path.join(td, "p")
        open(c, "w").write(src)
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), c, "-o", exe])  # headers are valid C
        vals = [int(x) for x in subprocess.check_output([exe]).split()]
    T = B.TensorDescription
    exp = [C.sizeof(T), T.strides.offset, T.storage_offset.offset, T.dim.offset, T.dtype.offset,
           C.sizeof(B.MatrixDescription), C.sizeof(B.ArrayDescription), C.sizeof(B.GRef), B.GRef.stride.offset,
           B.GRef.same_chunk.offset, C.sizeof(B.EnvFunc), C.sizeof(B.UniqueId), C.sizeof(B.ExtCollectives)]
    assert vals == exp


def test_init_fails_loudly_without_a_gpu(wm_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from wholegraph_amd import binding
    assert wm_lib.wholememory_ext_backend_name() == b"hip-gfx950"
    assert wm_lib.wholememory_init(0, binding.LEVEL_FATAL) == 4  # WHOLEMEMORY_CUDA_ERROR: no device, no fallback


def test_testing_seam_is_locked(wm_lib):
    old = os.environ.pop("WHOLEGRAPH_AMD_TESTING", None)
    try:
        assert wm_lib.wm_testing_install_backend(None) == 9  # NOT_SUPPORTED
    finally:
        if old is not None:
            os.environ["WHOLEGRAPH_AMD_TESTING"] = old
