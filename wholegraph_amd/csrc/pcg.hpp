// wholegraph_amd — the random generator behind neighbour sampling, usable from host and device code.
//
// The reference draws its random numbers from raft::random::detail::PCGenerator (raft branch-24.12, pinned in
// cpp/cmake/thirdparty/get_raft.cmake:17-18; call sites unweighted_sample_without_replacement_func.cuh:80,112,148,198
// and raft_random_gen.cu:43-62). raft is NOT vendored under /root/reference, so what follows restates the published
// algorithm — PCG-XSH-RR 64/32 (M. O'Neill, "PCG: A Family of Simple Fast Space-Efficient Statistically Good
// Algorithms for Random Number Generation", 2014; pcg32 reference implementation) with the seeding / stream
// selection / skip-ahead (F. Brown, "Random Number Generation with Arbitrary Strides", 1994) raft wraps around it:
//     state = 0; inc = (subsequence << 1) | 1; step; state += seed; step; skipahead(offset)
//     generator(rng_state, subsequence) = init(seed, base_subsequence + subsequence, offset = subsequence)
//     int32 draws are the 32-bit output with the sign bit cleared; UniformDistParams<int32>{0,1} returns the draw as is.
// PARITY UNPINNED: nothing in /root/reference (tests included) holds a fixed output of this generator — the
// reference's own tests re-run raft on the host — so bit-equality with raft cannot be demonstrated here.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define WM_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define WM_HD inline
#endif

namespace wm {

struct pcg32 {
  uint64_t state;
  uint64_t inc;

  // raft DeviceState{seed, base_subsequence} + per-thread subsequence
  WM_HD pcg32(uint64_t seed, uint64_t base_subsequence, uint64_t subsequence) { init(seed, base_subsequence + subsequence, subsequence); }

  WM_HD void init(uint64_t seed, uint64_t subsequence, uint64_t offset)
  {
    state = 0;
    inc   = (subsequence << 1u) | 1u;
    (void)next_u32();
    state += seed;
    (void)next_u32();
    skipahead(offset);
  }
  WM_HD void skipahead(uint64_t offset)
  {
    uint64_t G = 1, h = 6364136223846793005ULL, C = 0, f = inc;
    while (offset) {
      if (offset & 1) {
        G = G * h;
        C = C * h + f;
      }
      f = f * (h + 1);
      h = h * h;
      offset >>= 1;
    }
    state = state * G + C;
  }
  WM_HD uint32_t next_u32()
  {
    const uint64_t old        = state;
    state                     = old * 6364136223846793005ULL + inc;
    const uint32_t xorshifted = static_cast<uint32_t>(((old >> 18u) ^ old) >> 27u);
    const uint32_t rot        = static_cast<uint32_t>(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31u));
  }
  WM_HD uint64_t next_u64()
  {
    const uint32_t a = next_u32();
    const uint32_t b = next_u32();
    return static_cast<uint64_t>(a) | (static_cast<uint64_t>(b) << 32);
  }
  WM_HD int32_t next_i32() { return static_cast<int32_t>(next_u32() & 0x7fffffffu); }
  WM_HD int64_t next_i64() { return static_cast<int64_t>(next_u64() & 0x7fffffffffffffffULL); }
  WM_HD float next_float() { return static_cast<float>(next_u32() >> 8) / static_cast<float>(1u << 24); }
};

// log2(1 + x) for x in [-1, 0), in double, built from +, *, / only (no libm), so that host and device round every step
// identically (both are compiled with -ffp-contract=off): the weighted sampler ranks neighbours by keys that contain
// this value, and "device == oracle" must not depend on two math libraries agreeing in the last bit.
//   tiny |x|: x - x^2/2;   else U = 1 + x (exact: x carries 24 significant bits), U = f * 2^e with f in [sqrt(1/2),
//   sqrt(2)), ln U = e ln2 + 2 atanh((f-1)/(f+1)), atanh by its odd series to s^27 (|s| <= 0.172 -> < 1e-20).
WM_HD double det_log2_1p(double x)
{
  const double kInvLn2 = 1.4426950408889634074;
  const double kLn2    = 0.69314718055994530942;
  if (x > -7.450580596923828125e-9) return (x - x * x * 0.5) * kInvLn2;  // |x| < 2^-27
  const double U = 1.0 + x;
  if (!(U > 0.0)) return -__builtin_inf();
  uint64_t bits;
  __builtin_memcpy(&bits, &U, 8);
  int e = static_cast<int>((bits >> 52) & 0x7ff) - 1023;
  bits  = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
  double f;
  __builtin_memcpy(&f, &bits, 8);  // [1, 2)
  if (f > 1.4142135623730951) {
    f *= 0.5;
    e += 1;
  }
  const double s  = (f - 1.0) / (f + 1.0);
  const double s2 = s * s;
  double p        = 2.0 / 27.0;
  p               = p * s2 + 2.0 / 25.0;
  p               = p * s2 + 2.0 / 23.0;
  p               = p * s2 + 2.0 / 21.0;
  p               = p * s2 + 2.0 / 19.0;
  p               = p * s2 + 2.0 / 17.0;
  p               = p * s2 + 2.0 / 15.0;
  p               = p * s2 + 2.0 / 13.0;
  p               = p * s2 + 2.0 / 11.0;
  p               = p * s2 + 2.0 / 9.0;
  p               = p * s2 + 2.0 / 7.0;
  p               = p * s2 + 2.0 / 5.0;
  p               = p * s2 + 2.0 / 3.0;
  p               = p * s2 + 2.0;
  return (static_cast<double>(e) * kLn2 + p * s) * kInvLn2;
}

// A-Res key of one neighbour: log2(u) / weight with u uniform in (0,1) assembled from a 24-bit mantissa in [0.5, 1)
// and a geometric exponent (leading zero bits of a 64-bit stream, redrawn while it is all zeros) — the construction
// of weighted_sample_without_replacement_func.cuh:44-63. Larger key = more likely kept.
WM_HD float weighted_sample_key(pcg32& rng, float weight)
{
  const float u0 = rng.next_float();
  const float m  = static_cast<float>(-(0.5 + 0.5 * static_cast<double>(u0)));  // [-1, -0.5]
  uint64_t stream;
  int redraws = -1;
  do {
    stream = rng.next_u64();
    redraws++;
  } while (stream == 0);
  int zeros = redraws * 64;
  while ((stream >> 63) == 0) {  // count leading zeros without a builtin that differs between host and device
    stream <<= 1;
    zeros++;
  }
  double scale = 1.0;
  for (int z = zeros; z > 0;) {  // 2^-zeros, exact
    const int step = z > 30 ? 30 : z;
    scale *= 1.0 / static_cast<double>(1u << step);
    z -= step;
  }
  const float log2u = static_cast<float>(det_log2_1p(static_cast<double>(m) * scale));
  return log2u * (1.0f / weight);
}

// order-preserving map float -> uint32 (larger float = larger integer; -inf lowest)
WM_HD uint32_t orderable_float(float v)
{
  uint32_t b;
  __builtin_memcpy(&b, &v, 4);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// Launch geometry the reference derives from max_sample_count (unweighted_sample_without_replacement_func.cuh:
// 407-445): the sampler's random streams are keyed by (center node, thread), so the SAME virtual geometry must be
// used to reproduce the same samples, whatever the physical kernel looks like.
struct sample_geometry {
  int threads;  // virtual threads per center node
  int items;    // draws per virtual thread
};
WM_HD sample_geometry sample_geometry_for(int max_sample_count)
{
  const int warp_count[32] = {1, 1, 1, 2, 2, 2, 4, 4, 4, 4, 4, 4, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8};
  const int items[32]      = {1, 2, 3, 2, 3, 3, 2, 2, 3, 3, 3, 3, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4};
  const int f              = (max_sample_count - 1) / 32;
  sample_geometry g;
  g.threads = warp_count[f] * 32;
  g.items   = items[f];
  return g;
}

}  // namespace wm
