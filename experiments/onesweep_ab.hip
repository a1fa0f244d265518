// Experiment harness, round 3: the hand-written onesweep id sort (experiments/onesweep.cuh) against the tuned
// rocPRIM call the library used until now — bit-exact comparison of (sorted keys, order) and timing, several tile shapes,
// uniform / skewed / constant ids, ragged sizes, key widths.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/onesweep_ab.hip -o experiments/onesweep_ab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <rocprim/rocprim.hpp>
#include "onesweep.cuh"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

struct key_of_id {   // the library's narrow_key_iterator in small: ids[i] - base, `span` for ids outside
  const uint64_t* ptr; uint64_t base; uint32_t span;
  __host__ __device__ uint32_t operator[](int64_t i) const { const uint64_t o = ptr[i] - base; return o < span ? (uint32_t)o : span; }
};
struct narrow_fn { uint64_t base; uint32_t span; __host__ __device__ uint32_t operator()(const uint64_t& v) const { const uint64_t o = v - base; return o < span ? (uint32_t)o : span; } };

using tuned = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 12>, rocprim::kernel_config<1024, 8>, 9, rocprim::block_radix_rank_algorithm::match>, 0>;   // (round 4: never the merge sort)

static hipEvent_t e0, e1;
static int g_debug = 0;
template <typename F> float timed(F f, int reps)
{
  for (int i = 0; i < 3; i++) f();
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; i++) f();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps * 1e3f;
}

template <int BLOCK, int IPT>
bool own(const char* tag, key_of_id keys, size_t n, unsigned bits, uint32_t* d_sorted, uint32_t* d_order, void* ws, size_t ws_cap,
         const uint32_t* ref_sorted, const uint32_t* ref_order, int reps)
{
  const size_t need = wm::osw::workspace_bytes<BLOCK, IPT>((int64_t)n, bits);
  if (need > ws_cap) { printf("  own %4d x %2d: workspace %zu > cap\n", BLOCK, IPT, need); return false; }
  CK(hipMemset(d_sorted, 0xff, n * 4)); CK(hipMemset(d_order, 0xff, n * 4));
  int rc = wm::osw::sort_pairs<BLOCK, IPT>(keys, d_sorted, d_order, (int64_t)n, bits, ws, 0);
  CK(hipDeviceSynchronize());
  std::vector<uint32_t> hs(n), ho(n);
  CK(hipMemcpy(hs.data(), d_sorted, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ho.data(), d_order, n * 4, hipMemcpyDeviceToHost));
  size_t bad = 0, first = 0;
  for (size_t i = 0; i < n; i++) if (hs[i] != ref_sorted[i] || ho[i] != ref_order[i]) { if (!bad) first = i; bad++; }
  float us = reps > 0 ? timed([&] { wm::osw::sort_pairs<BLOCK, IPT>(keys, d_sorted, d_order, (int64_t)n, bits, ws, 0, g_debug); }, reps) : 0.f;
  const wm::osw::plan p = wm::osw::make_plan((int64_t)n, bits, BLOCK * IPT);
  printf("  own %4d x %2d (%d passes x %d bits, %d tiles) rc %d: %s", BLOCK, IPT, p.passes, p.radix_bits, p.tiles, rc, bad ? "MISMATCH" : "bit-exact");
  if (bad) printf(" (%zu wrong, first at %zu: got %u/%u want %u/%u)", bad, first, hs[first], ho[first], ref_sorted[first], ref_order[first]);
  if (reps > 0) printf("  %.1f us", us);
  printf("  [%s]\n", tag);
  return bad == 0;
}

int main(int argc, char** argv)
{
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  const int only = argc > 2 ? atoi(argv[2]) : -1;   // one case only
  g_debug = argc > 3 ? atoi(argv[3]) : 0;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct cs { const char* name; size_t n; uint64_t rows; int dist; };   // dist 0 uniform, 1 skewed (cube of a uniform), 2 constant, 3 with out-of-range ids
  const cs cases[] = {{"uniform 10M of 100M rows", 10000000, 100000000ull, 0}, {"skewed 10M of 100M rows", 10000000, 100000000ull, 1},
                      {"uniform 10M of 125M rows, ids outside the range", 10000000, 125000000ull, 3}, {"constant 3M", 3000001, 100000000ull, 2},
                      {"uniform 1000 of 1000 rows", 1000, 1000ull, 0}, {"uniform 1 of 5 rows", 1, 5ull, 0}, {"uniform 65537 of 2^20 rows", 65537, 1ull << 20, 0},
                      {"uniform 5M of 2^31 rows", 5000000, 1ull << 31, 0}, {"uniform 5M of 4e9 rows", 5000000, 4000000000ull, 0},
                      {"uniform 2M of 70000 rows", 2000003, 70000ull, 0},
                      // round 4: the sizes of one mini-batch
                      {"uniform 128K of 10M rows", 131072, 10000000ull, 0}, {"uniform 256K of 10M rows", 262144, 10000000ull, 0},
                      {"uniform 512K of 10M rows", 524288, 10000000ull, 0}, {"uniform 1M of 10M rows", 1048576, 10000000ull, 0},
                      {"uniform 512K of 100M rows", 524288, 100000000ull, 0}, {"uniform 2M of 100M rows", 2097152, 100000000ull, 0},
                      {"uniform 4M of 100M rows", 4194304, 100000000ull, 0}, {"skewed 512K of 100M rows", 524288, 100000000ull, 1}};
  const size_t cap = 10000000;
  uint64_t* d_ids; uint32_t *d_sorted, *d_order, *r_sorted; int32_t* r_order; void *ws, *rtemp;
  const size_t ws_cap = 256u << 20;
  CK(hipMalloc(&d_ids, cap * 8)); CK(hipMalloc(&d_sorted, cap * 4)); CK(hipMalloc(&d_order, cap * 4)); CK(hipMalloc(&r_sorted, cap * 4));
  CK(hipMalloc(&r_order, cap * 4)); CK(hipMalloc(&ws, ws_cap)); CK(hipMalloc(&rtemp, ws_cap));
  bool all = true;
  int ci = -1;
  for (const cs& c : cases) {
    ci++;
    if (only >= 0 && ci != only) continue;
    std::vector<uint64_t> h(c.n);
    std::mt19937_64 g(7);
    const uint64_t base = c.dist == 3 ? 1000 : 0;
    for (auto& v : h) {
      const uint64_t u = g();
      if (c.dist == 0) v = u % c.rows;
      else if (c.dist == 1) { const double x = (u >> 11) * (1.0 / 9007199254740992.0); v = (uint64_t)(x * x * x * x * x * c.rows) * 2654435761ull % c.rows; }
      else if (c.dist == 2) v = 4242;
      else v = (u & 15) == 0 ? (u & 16 ? ~0ull - (u >> 40) : base + c.rows + (u >> 50)) : base + u % c.rows;
    }
    CK(hipMemcpy(d_ids, h.data(), c.n * 8, hipMemcpyHostToDevice));
    const uint32_t span = (uint32_t)c.rows;
    unsigned bits = 1;
    while (bits < 32 && (((uint64_t)span) >> bits) != 0) bits++;   // bits of span itself (the out-of-range marker is a key too)
    key_of_id keys{d_ids, base, span};
    auto rkeys = rocprim::make_transform_iterator(d_ids, narrow_fn{base, span});
    rocprim::counting_iterator<int32_t> pos(0);
    size_t need = 0;
    CK((rocprim::radix_sort_pairs<tuned>(nullptr, need, rkeys, r_sorted, pos, r_order, c.n, 0, bits, nullptr)));
    if (need > ws_cap) { printf("rocprim temp too big\n"); return 1; }
    CK((rocprim::radix_sort_pairs<tuned>(rtemp, need, rkeys, r_sorted, pos, r_order, c.n, 0, bits, nullptr)));
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> rs(c.n), ro(c.n);
    CK(hipMemcpy(rs.data(), r_sorted, c.n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ro.data(), r_order, c.n * 4, hipMemcpyDeviceToHost));
    const int r = c.n >= 100000 ? reps : 0;
    printf("%s (n %zu, %u bits)\n", c.name, c.n, bits);
    if (r > 0) printf("  rocPRIM tuned onesweep: %.1f us\n", timed([&] { (void)rocprim::radix_sort_pairs<tuned>(rtemp, need, rkeys, r_sorted, pos, r_order, c.n, 0, bits, nullptr); }, r));
    all &= own<512, 16>("", keys, c.n, bits, d_sorted, d_order, ws, ws_cap, rs.data(), ro.data(), r);
    all &= own<256, 16>("", keys, c.n, bits, d_sorted, d_order, ws, ws_cap, rs.data(), ro.data(), r);
    all &= own<1024, 8>("", keys, c.n, bits, d_sorted, d_order, ws, ws_cap, rs.data(), ro.data(), r);
    all &= own<512, 24>("", keys, c.n, bits, d_sorted, d_order, ws, ws_cap, rs.data(), ro.data(), r);
    all &= own<1024, 16>("", keys, c.n, bits, d_sorted, d_order, ws, ws_cap, rs.data(), ro.data(), r);
    fflush(stdout);
  }
  printf(all ? "ALL BIT-EXACT\n" : "MISMATCHES\n");
  return all ? 0 : 1;
}
