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
