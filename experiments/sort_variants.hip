// Side measurement: rocPRIM radix_sort_pairs configurations for the owner-side id sort (10 M keys of 27 significant bits,
// payload = position). hipcc --offload-arch=gfx950 -O3 experiments/sort_variants.hip -o experiments/gv_sort_variants
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <rocprim/rocprim.hpp>

#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct narrow_to_u32 {
  __host__ __device__ uint32_t operator()(const uint64_t& v) const { return static_cast<uint32_t>(v); }
};

template <class Config>
float run(const char* name, const uint64_t* d_ids, uint32_t* d_sorted, int32_t* d_order, size_t n, unsigned bits, void* temp,
          size_t temp_cap)
{
  rocprim::counting_iterator<int32_t> pos(0);
  auto keys = rocprim::make_transform_iterator(d_ids, narrow_to_u32());
  size_t need = 0;
  CK((rocprim::radix_sort_pairs<Config>(nullptr, need, keys, d_sorted, pos, d_order, n, 0, bits, nullptr)));
  if (need > temp_cap) { printf("%s: temp %zu > cap\n", name, need); return -1; }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) CK((rocprim::radix_sort_pairs<Config>(temp, need, keys, d_sorted, pos, d_order, n, 0, bits, nullptr)));
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; i++) CK((rocprim::radix_sort_pairs<Config>(temp, need, keys, d_sorted, pos, d_order, n, 0, bits, nullptr)));
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-44s bits %2u : %.1f us\n", name, bits, ms / reps * 1e3);
  return ms / reps;
}

template <unsigned BS, unsigned IPT, unsigned RB>
using osw = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 12>, rocprim::kernel_config<BS, IPT>, RB,
                                                                           rocprim::block_radix_rank_algorithm::match>>;

int main(int argc, char** argv)
{
  const size_t n   = argc > 1 ? atoll(argv[1]) : 10000000;
  const uint64_t N = argc > 2 ? atoll(argv[2]) : 100000000;
  std::vector<uint64_t> h(n);
  std::mt19937_64 g(1);
  for (auto& v : h) v = g() % N;
  uint64_t* d_ids; uint32_t* d_sorted; int32_t* d_order; void* temp;
  const size_t cap = 1ull << 30;
  CK(hipMalloc(&d_ids, n * 8)); CK(hipMalloc(&d_sorted, n * 4)); CK(hipMalloc(&d_order, n * 4)); CK(hipMalloc(&temp, cap));
  CK(hipMemcpy(d_ids, h.data(), n * 8, hipMemcpyHostToDevice));
  unsigned bits = 1;
  while ((N - 1) >> bits) bits++;
  run<rocprim::default_config>("default", d_ids, d_sorted, d_order, n, bits, temp, cap);
  run<rocprim::default_config>("default, 32 bits", d_ids, d_sorted, d_order, n, 32, temp, cap);
  run<osw<1024, 8, 9>>("onesweep match <1024,8> 9 bits/pass", d_ids, d_sorted, d_order, n, bits, temp, cap);
  run<osw<1024, 7, 9>>("onesweep match <1024,7> 9 bits/pass", d_ids, d_sorted, d_order, n, bits, temp, cap);
  run<osw<1024, 6, 9>>("onesweep match <1024,6> 9 bits/pass", d_ids, d_sorted, d_order, n, bits, temp, cap);
  run<osw<1024, 10, 9>>("onesweep match <1024,10> 9 bits/pass", d_ids, d_sorted, d_order, n, bits, temp, cap);
  run<osw<768, 8, 9>>("onesweep match <768,8> 9 bits/pass", d_ids, d_sorted, d_order, n, bits, temp, cap);
  run<osw<512, 8, 9>>("onesweep match <512,8> 9 bits/pass", d_ids, d_sorted, d_order, n, bits, temp, cap);
  run<osw<512, 10, 9>>("onesweep match <512,10> 9 bits/pass", d_ids, d_sorted, d_order, n, bits, temp, cap);
  run<osw<256, 16, 9>>("onesweep match <256,16> 9 bits/pass", d_ids, d_sorted, d_order, n, bits, temp, cap);
  run<osw<1024, 4, 9>>("onesweep match <1024,4> 9 bits/pass", d_ids, d_sorted, d_order, n, bits, temp, cap);
  run<osw<1024, 8, 10>>("onesweep match <1024,8> 10 bits/pass", d_ids, d_sorted, d_order, n, bits, temp, cap);
  run<osw<512, 8, 10>>("onesweep match <512,8> 10 bits/pass", d_ids, d_sorted, d_order, n, bits, temp, cap);
  run<osw<1024, 4, 10>>("onesweep match <1024,4> 10 bits/pass", d_ids, d_sorted, d_order, n, bits, temp, cap);
  return 0;
}
