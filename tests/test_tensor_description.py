"""Descriptor helpers of the product library vs the REFERENCE ITSELF: cpp/src/wholememory/tensor_description.cpp
is the one reference TU that compiles without CUDA; oracle/Makefile builds it in place into
oracle/_ref/libref_tensor_description.so and every function is compared call-for-call on random inputs."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libref_tensor_description.so")


@pytest.fixture(scope="module")
def libs(wm_lib):
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref not built (reference tree absent and no prebuilt .so)")
    from wholegraph_amd import binding as B
    ref = C.CDLL(REF, mode=os.RTLD_LOCAL | getattr(os, "RTLD_DEEPBIND", 0))
    for name, (res, args) in B.PROTOTYPES.items():
        if hasattr(ref, name) and name.startswith("wholememory_") and ("desc" in name or "dtype" in name or "tensor" in name):
            try:
                f = getattr(ref, name)
                f.restype, f.argtypes = res, args
            except AttributeError:
                pass
    return wm_lib, ref, B


def rand_desc(B, rng):
    d = B.TensorDescription()
    for i in range(8):
        d.sizes[i] = int(rng.integers(1, 6))
        d.strides[i] = int(rng.integers(1, 6))
    d.storage_offset = int(rng.integers(0, 9))
    d.dim = int(rng.integers(0, 8))
    d.dtype = int(rng.integers(0, 10))
    if rng.random() < 0.5 and d.dim > 0:
        d.strides[d.dim - 1] = 1
    return d


def same(a, b):
    return bytes(a) == bytes(b)


def test_dtype_queries(libs):
    mine, ref, B = libs
    for dt in range(0, 10):
        assert mine.wholememory_dtype_get_element_size(dt) == ref.wholememory_dtype_get_element_size(dt)
        assert mine.wholememory_dtype_is_floating_number(dt) == ref.wholememory_dtype_is_floating_number(dt)
        assert mine.wholememory_dtype_is_integer_number(dt) == ref.wholememory_dtype_is_integer_number(dt)


def test_conversions_and_extents(libs):
    mine, ref, B = libs
    rng = np.random.default_rng(0)
    for _ in range(3000):
        d = rand_desc(B, rng)
        for fn in ("wholememory_get_memory_element_count_from_tensor", "wholememory_get_memory_size_from_tensor"):
            d1, d2 = B.TensorDescription.from_buffer_copy(d), B.TensorDescription.from_buffer_copy(d)
            assert getattr(mine, fn)(C.byref(d1)) == getattr(ref, fn)(C.byref(d2))
        a1, a2 = B.ArrayDescription(), B.ArrayDescription()
        d1, d2 = B.TensorDescription.from_buffer_copy(d), B.TensorDescription.from_buffer_copy(d)
        r1 = mine.wholememory_convert_tensor_desc_to_array(C.byref(a1), C.byref(d1))
        r2 = ref.wholememory_convert_tensor_desc_to_array(C.byref(a2), C.byref(d2))
        assert r1 == r2
        if r1:
            assert (a1.size, a1.storage_offset, a1.dtype) == (a2.size, a2.storage_offset, a2.dtype)
            assert mine.wholememory_get_memory_size_from_array(C.byref(a1)) == ref.wholememory_get_memory_size_from_array(C.byref(a2))
            m1, m2 = B.MatrixDescription(), B.MatrixDescription()
            mine.wholememory_copy_array_desc_to_matrix(C.byref(m1), C.byref(a1))
            ref.wholememory_copy_array_desc_to_matrix(C.byref(m2), C.byref(a2))
            assert (m1.sizes[0], m1.sizes[1], m1.stride, m1.storage_offset, m1.dtype) == \
                   (m2.sizes[0], m2.sizes[1], m2.stride, m2.storage_offset, m2.dtype)
            t1, t2 = B.TensorDescription(), B.TensorDescription()
            mine.wholememory_copy_array_desc_to_tensor(C.byref(t1), C.byref(a1))
            ref.wholememory_copy_array_desc_to_tensor(C.byref(t2), C.byref(a2))
            assert same(t1, t2)
        m1, m2 = B.MatrixDescription(), B.MatrixDescription()
        r1 = mine.wholememory_convert_tensor_desc_to_matrix(C.byref(m1), C.byref(d1))
        r2 = ref.wholememory_convert_tensor_desc_to_matrix(C.byref(m2), C.byref(d2))
        assert r1 == r2
        if r1:
            assert (m1.sizes[0], m1.sizes[1], m1.stride, m1.storage_offset, m1.dtype) == \
                   (m2.sizes[0], m2.sizes[1], m2.stride, m2.storage_offset, m2.dtype)
            assert mine.wholememory_get_memory_element_count_from_matrix(C.byref(m1)) == \
                   ref.wholememory_get_memory_element_count_from_matrix(C.byref(m2))
            assert mine.wholememory_get_memory_size_from_matrix(C.byref(m1)) == ref.wholememory_get_memory_size_from_matrix(C.byref(m2))
            t1, t2 = B.TensorDescription(), B.TensorDescription()
            mine.wholememory_copy_matrix_desc_to_tensor(C.byref(t1), C.byref(m1))
            ref.wholememory_copy_matrix_desc_to_tensor(C.byref(t2), C.byref(m2))
            assert same(t1, t2)


def test_squeeze_unsqueeze(libs):
    mine, ref, B = libs
    rng = np.random.default_rng(1)
    for _ in range(3000):
        d = rand_desc(B, rng)
        d.dim = int(rng.integers(1, 7))
        if rng.random() < 0.6:
            d.sizes[int(rng.integers(0, d.dim))] = 1
        dim = int(rng.integers(-1, d.dim + 2))
        for fn in ("wholememory_squeeze_tensor", "wholememory_unsqueeze_tensor"):
            d1, d2 = B.TensorDescription.from_buffer_copy(d), B.TensorDescription.from_buffer_copy(d)
            r1, r2 = getattr(mine, fn)(C.byref(d1), dim), getattr(ref, fn)(C.byref(d2), dim)
            assert r1 == r2
            if r1:
                assert d1.dim == d2.dim
                assert [d1.sizes[i] for i in range(d1.dim)] == [d2.sizes[i] for i in range(d2.dim)]
                assert [d1.strides[i] for i in range(d1.dim)] == [d2.strides[i] for i in range(d2.dim)]


def test_constructors(libs):
    mine, ref, B = libs
    a1 = mine.wholememory_create_array_desc(77, 3, B.DT_INT64)
    a2 = ref.wholememory_create_array_desc(77, 3, B.DT_INT64)
    assert (a1.size, a1.storage_offset, a1.dtype) == (a2.size, a2.storage_offset, a2.dtype)
    sz = (C.c_int64 * 2)(9, 5)
    m1 = mine.wholememory_create_matrix_desc(sz, 8, 2, B.DT_HALF)
    m2 = ref.wholememory_create_matrix_desc(sz, 8, 2, B.DT_HALF)
    assert (m1.sizes[0], m1.sizes[1], m1.stride, m1.storage_offset, m1.dtype) == \
           (m2.sizes[0], m2.sizes[1], m2.stride, m2.storage_offset, m2.dtype)
    t1, t2 = B.TensorDescription(), B.TensorDescription()
    mine.wholememory_initialize_tensor_desc(C.byref(t1))
    ref.wholememory_initialize_tensor_desc(C.byref(t2))
    assert same(t1, t2)
