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

}  // namespace wm
