// wholegraph_amd — device-side element types and the reference's conversion chain.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace wm {

using half_t = _Float16;  // IEEE binary16, v_cvt_f16_f32 / v_cvt_f32_f16 (round-to-nearest-even)

struct bf16_t {
  uint16_t bits;
};

__device__ __forceinline__ float bf16_to_float(bf16_t b) { return __uint_as_float(static_cast<uint32_t>(b.bits) << 16); }
__device__ __forceinline__ bf16_t float_to_bf16(float f)
{
  uint32_t x = __float_as_uint(f);
  bf16_t r;
  if ((x & 0x7fffffffu) > 0x7f800000u) {
    r.bits = static_cast<uint16_t>((x >> 16) | 0x40u);
  } else {
    x += 0x7fffu + ((x >> 16) & 1u);
    r.bits = static_cast<uint16_t>(x >> 16);
  }
  return r;
}

// convert_type<From,To> of reference functions/gather_scatter_func.cuh:161-208:
// half / bf16 are loaded as float and stored from float (so double -> half is double -> float ->
// half, two roundings); every other pair is a plain static_cast.
template <typename T>
struct wide_of {
  using type = T;
};
template <>
struct wide_of<half_t> {
  using type = float;
};
template <>
struct wide_of<bf16_t> {
  using type = float;
};

template <typename T>
__device__ __forceinline__ typename wide_of<T>::type load_wide(T v)
{
  return v;
}
template <>
__device__ __forceinline__ float load_wide<half_t>(half_t v)
{
  return static_cast<float>(v);
}
template <>
__device__ __forceinline__ float load_wide<bf16_t>(bf16_t v)
{
  return bf16_to_float(v);
}

template <typename T>
__device__ __forceinline__ T store_narrow(typename wide_of<T>::type v)
{
  return static_cast<T>(v);
}
template <>
__device__ __forceinline__ half_t store_narrow<half_t>(float v)
{
  return static_cast<half_t>(v);
}
template <>
__device__ __forceinline__ bf16_t store_narrow<bf16_t>(float v)
{
  return float_to_bf16(v);
}

template <typename FromT, typename ToT>
__device__ __forceinline__ ToT convert_elt(FromT v)
{
  using to_wide = typename wide_of<ToT>::type;
  return store_narrow<ToT>(static_cast<to_wide>(load_wide<FromT>(v)));
}

// Row pointers that travel between lanes (v_readlane / ds_bpermute) come back as integers, and hipcc then no longer knows
// they point to global memory: it emits FLAT loads and stores, which count on lgkmcnt as well as vmcnt — every wait for a
// ds_bpermute result (or a scalar load) then also drains the row loads in flight, one load per wave at a time. These
// helpers state the address space: table, plain, gradient and state rows are always global memory (device, pinned host
// or a peer's mapping), never LDS or scratch.
#define WM_GLOBAL_AS __attribute__((address_space(1)))
template <typename V>
__device__ __forceinline__ V ld_global(const void* p)
{
  return *(const WM_GLOBAL_AS V*)p;
}
template <typename V>
__device__ __forceinline__ V ld_global_nt(const void* p)
{
  return __builtin_nontemporal_load((const WM_GLOBAL_AS V*)p);
}
template <typename V>
__device__ __forceinline__ void st_global(void* p, V v)
{
  *(WM_GLOBAL_AS V*)p = v;
}
template <typename V>
__device__ __forceinline__ void st_global_nt(void* p, V v)
{
  __builtin_nontemporal_store(v, (WM_GLOBAL_AS V*)p);
}

}  // namespace wm
