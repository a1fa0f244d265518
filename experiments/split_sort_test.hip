// EXPERIMENT harness (round 5): the split sort of kernels/split_sort.cuh against a CPU stable sort, and its timing.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I wholegraph_amd/csrc experiments/split_sort_test.hip -o experiments/split_sort_test
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>
#define WM_SPLIT_DEBUG 1
#include "kernels/split_sort.cuh"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

using namespace wm;

struct result { std::vector<int64_t> uniq; std::vector<int32_t> starts, order; int64_t n_unique; };

static result cpu_ref(const std::vector<int64_t>& ids, int64_t lower, int64_t span)
{
  const int64_t n = ids.size();
  std::vector<uint32_t> key(n);
  for (int64_t i = 0; i < n; i++) {
    const uint64_t off = static_cast<uint64_t>(ids[i]) - static_cast<uint64_t>(lower);
    key[i] = off < static_cast<uint64_t>(span) ? static_cast<uint32_t>(off) : static_cast<uint32_t>(span);
  }
  result r;
  r.order.resize(n);
  std::iota(r.order.begin(), r.order.end(), 0);
  std::stable_sort(r.order.begin(), r.order.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
  for (int64_t j = 0; j < n; j++) {
    const uint32_t k = key[r.order[j]];
    if (k == static_cast<uint32_t>(span)) break;
    if (j == 0 || k != key[r.order[j - 1]]) { r.uniq.push_back(static_cast<int64_t>(k) + lower); r.starts.push_back(static_cast<int32_t>(j)); }
  }
  int64_t valid = 0;
  for (int64_t i = 0; i < n; i++) valid += key[i] != static_cast<uint32_t>(span);
  r.n_unique = r.uniq.size();
  r.starts.push_back(static_cast<int32_t>(valid));
  return r;
}

static int run_case(const char* name, const std::vector<int64_t>& ids, int64_t lower, int64_t span, bool timing)
{
  const int64_t n = ids.size();
  const bool hot = getenv("SPLIT_HOT") != nullptr && getenv("SPLIT_HOT")[0] == '1';   // round 6: hot ids get buckets of their own
  split::plan p = split::make_plan(n, span, getenv("SPLIT_IPT") ? atoi(getenv("SPLIT_IPT")) : 0, getenv("SPLIT_CAPBITS") ? atoi(getenv("SPLIT_CAPBITS")) : 0, hot);
  if (!p.ok) { printf("%-40s n=%ld span=%ld: plan not ok (skipped)\n", name, (long)n, (long)span); return 0; }
  int64_t* d_ids; void* d_ws; int64_t* d_uniq; int32_t *d_starts, *d_order; int64_t* d_nu;
  CK(hipMalloc(&d_ids, 8 * n)); CK(hipMalloc(&d_ws, p.total)); CK(hipMalloc(&d_uniq, 8 * n)); CK(hipMalloc(&d_starts, 4 * (n + 1)));
  CK(hipMalloc(&d_order, 4 * n)); CK(hipMalloc(&d_nu, 8));
  CK(hipMemcpy(d_ids, ids.data(), 8 * n, hipMemcpyHostToDevice));
  CK(hipMemset(d_ws, 0xCD, p.total)); CK(hipMemset(d_order, 0xFF, 4 * n)); CK(hipMemset(d_starts, 0xFF, 4 * (n + 1)));
  hipStream_t st; CK(hipStreamCreate(&st));
  int rc = hot ? split::launch_hot<uint64_t>(p, reinterpret_cast<const uint64_t*>(d_ids), n, static_cast<uint64_t>(lower), static_cast<uint32_t>(span),
                                             d_uniq, d_starts, d_order, d_nu, d_ws, nullptr, 0, st)
               : split::launch<uint64_t>(p, reinterpret_cast<const uint64_t*>(d_ids), n, static_cast<uint64_t>(lower), static_cast<uint32_t>(span),
                                   d_uniq, d_starts, d_order, d_nu, d_ws, nullptr, 0, st);
  CK(hipStreamSynchronize(st));
  uint32_t hot_n[4] = {0, 0, 0, 0};
  if (hot) CK(hipMemcpy(hot_n, static_cast<char*>(d_ws) + p.off_hot_n, sizeof(hot_n), hipMemcpyDeviceToHost));
  if (rc != 0) { printf("%s: launch rc %d\n", name, rc); return 1; }
  uint32_t ctl[split::kCtlWords];
  CK(hipMemcpy(ctl, static_cast<char*>(d_ws) + p.off_ctl, sizeof(ctl), hipMemcpyDeviceToHost));
  int bad = 0;
  if (ctl[split::kCtlError]) { printf("%s: look-back timeout flagged\n", name); bad = 1; }
  result ref = cpu_ref(ids, lower, span);
  // does the CPU agree about the overflow? (HOT mode: which ids were peeled is the device's choice — not checked)
  if (!hot) {
    std::vector<int64_t> cnt(p.buckets + 1, 0);
    for (int64_t i = 0; i < n; i++) {
      const uint64_t off = static_cast<uint64_t>(ids[i]) - static_cast<uint64_t>(lower);
      if (off < static_cast<uint64_t>(span)) cnt[off >> p.shift]++;
    }
    bool ov = false;
    for (int b = 0; b < p.buckets; b++) ov |= cnt[b] > (1 << p.cap_bits);
    if (ov != (ctl[split::kCtlOverflow] != 0)) { printf("%s: overflow flag %u, expected %d\n", name, ctl[split::kCtlOverflow], (int)ov); bad = 1; }
  }
  if (ctl[split::kCtlOverflow]) {
    printf("%-40s n=%ld span=%ld shift=%d buckets=%d: OVERFLOW (as expected: %s)%s\n", name, (long)n, (long)span, p.shift, p.buckets, bad ? "NO" : "yes", hot ? " [hot mode]" : "");
    if (hot) printf("    hot ids %u (threshold %u samples)\n", hot_n[0], hot_n[1]);
  } else {
    int64_t nu; CK(hipMemcpy(&nu, d_nu, 8, hipMemcpyDeviceToHost));
    std::vector<int32_t> order(n), starts(n + 1); std::vector<int64_t> uniq(n);
    CK(hipMemcpy(order.data(), d_order, 4 * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(starts.data(), d_starts, 4 * (n + 1), hipMemcpyDeviceToHost));
    CK(hipMemcpy(uniq.data(), d_uniq, 8 * n, hipMemcpyDeviceToHost));
    if (nu != ref.n_unique) { printf("%s: n_unique %ld, expected %ld\n", name, (long)nu, (long)ref.n_unique); bad = 1; }
    else {
      const int64_t valid = ref.starts.back();
      int64_t e = 0;
      for (int64_t j = 0; j < valid && e < 5; j++) if (order[j] != ref.order[j]) { printf("%s: order[%ld] = %d, expected %d\n", name, (long)j, order[j], ref.order[j]); e++; }
      // the tail holds the dropped positions (any order: a multiset compare)
      {
        std::vector<int32_t> a(order.begin() + valid, order.end()), b(ref.order.begin() + valid, ref.order.end());
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
        if (a != b) { printf("%s: the tail of order[] does not hold the dropped positions\n", name); e++; }
      }
      for (int64_t j = 0; j <= nu && e < 10; j++) if (starts[j] != ref.starts[j]) { printf("%s: run_starts[%ld] = %d, expected %d\n", name, (long)j, starts[j], ref.starts[j]); e++; }
      for (int64_t j = 0; j < nu && e < 15; j++) if (uniq[j] != ref.uniq[j]) { printf("%s: unique[%ld] = %ld, expected %ld\n", name, (long)j, (long)uniq[j], (long)ref.uniq[j]); e++; }
      bad |= e != 0;
    }
    printf("%-40s n=%ld span=%ld shift=%d buckets=%d cap=%d tiles=%d ipt=%d passes=%dx%d n_unique=%ld: %s\n", name, (long)n, (long)span, p.shift,
           p.buckets, 1 << p.cap_bits, p.tiles, p.ipt, p.passes, p.digit_bits, (long)nu, bad ? "MISMATCH" : "ok");
    if (hot) printf("    hot ids %u (threshold %u samples), segments put in order afterwards %u\n", hot_n[0], hot_n[1], hot_n[2]);
  }
  if (timing && !bad) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0, st));
      const int K = 20;
      for (int k = 0; k < K; k++)
        if (hot)
          split::launch_hot<uint64_t>(p, reinterpret_cast<const uint64_t*>(d_ids), n, static_cast<uint64_t>(lower), static_cast<uint32_t>(span), d_uniq,
                                      d_starts, d_order, d_nu, d_ws, nullptr, 0, st);
        else
          split::launch<uint64_t>(p, reinterpret_cast<const uint64_t*>(d_ids), n, static_cast<uint64_t>(lower), static_cast<uint32_t>(span), d_uniq,
                                  d_starts, d_order, d_nu, d_ws, nullptr, 0, st);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("    timing: %.1f us per call (%s)\n", ms * 1000.f / K, hot ? "hot mode: 7 launches" : "4 launches");
    }
  }
  CK(hipFree(d_ids)); CK(hipFree(d_ws)); CK(hipFree(d_uniq)); CK(hipFree(d_starts)); CK(hipFree(d_order)); CK(hipFree(d_nu));
  CK(hipStreamDestroy(st));
  return bad;
}

int main(int argc, char** argv)
{
  const bool timing = argc > 1 && strcmp(argv[1], "time") == 0;
  if (argc > 1 && strcmp(argv[1], "fuzz") == 0) {   // fuzz <cases> <seed>: random sizes, spans, offsets, id distributions
    const int cases = argc > 2 ? atoi(argv[2]) : 200;
    std::mt19937_64 r(argc > 3 ? atoll(argv[3]) : 7);
    int failed = 0, overflowed = 0;
    for (int c = 0; c < cases; c++) {
      const int64_t span = 1 + static_cast<int64_t>(r() % (r() % 3 == 0 ? (1ull << 29) : (r() % 2 ? 200000000ull : 3000000ull)));
      const int64_t n    = 1 + static_cast<int64_t>(r() % (r() % 4 == 0 ? 3000000ull : 200000ull));
      const int64_t lower = (r() % 2) ? static_cast<int64_t>(r() % 1000000000ull) : 0;
      const int kind = static_cast<int>(r() % 6);
      std::vector<int64_t> v(n);
      const uint64_t hot = r() % static_cast<uint64_t>(span);
      const uint64_t width = 1 + r() % static_cast<uint64_t>(span);
      for (int64_t i = 0; i < n; i++) {
        uint64_t k;
        switch (kind) {
          case 0: k = r() % static_cast<uint64_t>(span); break;                                         // uniform
          case 1: k = (r() % 100 < 30) ? hot : r() % static_cast<uint64_t>(span); break;                // one hot id
          case 2: k = (hot + r() % std::min<uint64_t>(width, 5000)) % static_cast<uint64_t>(span); break;   // clustered
          case 3: k = static_cast<uint64_t>(i) * 7 % static_cast<uint64_t>(span); break;                // ascending
          case 4: k = (r() % std::max<uint64_t>(1, static_cast<uint64_t>(n) / 3)) * 977 % static_cast<uint64_t>(span); break;   // ~3 ids per row
          default: k = (r() % 64) * (static_cast<uint64_t>(span) / 64 + 1) % static_cast<uint64_t>(span); break;   // 64 rows
        }
        v[i] = lower + static_cast<int64_t>(k);
        if (r() % 50 == 0) v[i] = (r() & 1) ? -1 - static_cast<int64_t>(r() % 100) : lower + span + static_cast<int64_t>(r() % 100);
      }
      char nm[64];
      snprintf(nm, sizeof(nm), "fuzz %d kind %d", c, kind);
      failed += run_case(nm, v, lower, span, false);
    }
    printf(failed ? "FUZZ FAILED (%d)\n" : "FUZZ OK (%d failures)\n", failed);
    (void)overflowed;
    return failed != 0;
  }
  if (argc > 2) {   // debug mode: only the big case, kernels with parts switched off (results are wrong on purpose)
    int dbg = atoi(argv[2]);
    std::mt19937_64 r2(42);
    std::vector<int64_t> v(10000000);
    for (auto& x : v) x = static_cast<int64_t>(r2() % 100000000ull);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(wm::split::g_split_debug), &dbg, sizeof(int)));
    run_case("10M uniform / 100M rows (debug)", v, 0, 100000000, dbg != 0);
    if (dbg == 0) {   // phase times of the last (only) call
      static unsigned long long t[2][4096][12];
      CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(wm::split::g_split_times), sizeof(t)));
      const int nph[2] = {7, 11}, nwg[2] = {480, 3000};
      for (int k = 0; k < 2; k++) {
        unsigned long long t0 = ~0ull, t1 = 0;
        double ph[12] = {0};
        for (int g = 0; g < nwg[k]; g++) {
          if (t[k][g][0] < t0) t0 = t[k][g][0];
          if (t[k][g][nph[k] - 1] > t1) t1 = t[k][g][nph[k] - 1];
          for (int q = 1; q < nph[k]; q++) ph[q] += double(t[k][g][q] - t[k][g][q - 1]);
        }
        printf("kernel %d: span %.1f us; mean per-workgroup phase durations (us):", k, (t1 - t0) / 100.0);
        double tot = 0;
        for (int q = 1; q < nph[k]; q++) { printf(" %.2f", ph[q] / nwg[k] / 100.0); tot += ph[q] / nwg[k] / 100.0; }
        printf("  | total %.2f\n", tot);
      }
    }
    return 0;
  }
  std::mt19937_64 rng(42);
  int bad = 0;
  auto uniform = [&](int64_t n, int64_t lower, int64_t span, double drop_frac) {
    std::vector<int64_t> v(n);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    for (auto& x : v) {
      x = lower + static_cast<int64_t>(rng() % static_cast<uint64_t>(span));
      if (u(rng) < drop_frac) x = (rng() & 1) ? -1 - static_cast<int64_t>(rng() % 1000) : lower + span + static_cast<int64_t>(rng() % 1000);
    }
    return v;
  };
  const int64_t sizes[] = {1, 2, 63, 64, 65, 1000, 4097, 50000, 300000, 1000000};
  const bool big_only = argc > 1 && strcmp(argv[1], "big") == 0;   // only the 10 M-id cases, timed (for rocprofv3 --stats)
  const int64_t spans[] = {1, 7, 100, 65536, 1000003, 100000000, 125000000, INT64_C(1) << 27, (INT64_C(1) << 29) + 12345};
  if (!big_only) {
  for (int64_t n : sizes)
    for (int64_t sp : spans) {
      char nm[96];
      snprintf(nm, sizeof(nm), "uniform");
      bad |= run_case(nm, uniform(n, 0, sp, 0.0), 0, sp, false);
      snprintf(nm, sizeof(nm), "uniform+drops+lower");
      bad |= run_case(nm, uniform(n, 12345678, sp, 0.05), 12345678, sp, false);
    }
  // all dropped, all equal
  bad |= run_case("all dropped", std::vector<int64_t>(5000, -1), 0, 100000000, false);
  bad |= run_case("all equal (small)", std::vector<int64_t>(9000, 77), 0, 100000000, false);
  bad |= run_case("all equal (overflow)", std::vector<int64_t>(40000, 77), 0, 100000000, false);
  // sorted, reversed
  { std::vector<int64_t> v(200000); for (size_t i = 0; i < v.size(); i++) v[i] = static_cast<int64_t>(i) * 400; bad |= run_case("ascending", v, 0, 100000000, false);
    std::reverse(v.begin(), v.end()); bad |= run_case("descending", v, 0, 100000000, false); }
  // zipf (hashed): the hot id overflows its bucket
  {
    const int64_t n = 2000000, N = 100000000;
    std::vector<int64_t> v(n);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    for (auto& x : v) { const double r = u(rng); const uint64_t k = static_cast<uint64_t>(1.0 / std::pow(1.0 - r * 0.9999, 20.0)); x = static_cast<int64_t>((k * 2654435761ull) % N); }
    bad |= run_case("zipf-like hashed", v, 0, N, false);
  }
  }
  // the C4-sized batch
  bad |= run_case("10M uniform / 100M rows", uniform(10000000, 0, 100000000, 0.0), 0, 100000000, timing || big_only);
  if (big_only) {
    const int64_t n = 10000000, N = 100000000;
    std::vector<int64_t> v(n);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    const double a = 1.05, am1 = a - 1.0, b = std::pow(2.0, am1);
    for (auto& x : v) {
      for (;;) {
        const double U = 1.0 - u(rng), V = u(rng);
        const double X = std::floor(std::pow(U, -1.0 / am1));
        if (X < 1.0 || X > 9e18) continue;
        const double T = std::pow(1.0 + 1.0 / X, am1);
        if (V * X * (T - 1.0) / (b - 1.0) <= T / b) { x = static_cast<int64_t>((static_cast<uint64_t>(X) * 2654435761ull) % N); break; }
      }
    }
    if (getenv("SPLIT_FIX_ONLY")) {   // split_fix_small_kernel with only one class of segments (results wrong on purpose): kernel times from rocprofv3
      int dbg = atoi(getenv("SPLIT_FIX_ONLY"));
      CK(hipMemcpyToSymbol(HIP_SYMBOL(wm::split::g_split_debug), &dbg, sizeof(int)));
    }
    bad |= run_case("10M zipf(1.05) hashed / 100M rows", v, 0, N, true);
    printf(bad ? "FAILED\n" : "ALL OK\n");
    return bad;
  }
  bad |= run_case("10M uniform / 125M rows (shard)", uniform(10000000, 375000000, 125000000, 0.0), 375000000, 125000000, timing);
  bad |= run_case("0.5M uniform / 100M rows", uniform(500000, 0, 100000000, 0.0), 0, 100000000, timing);
  // shapes that stress one stage: ids in ascending order (a tile feeds one or two buckets: every id of a tile on the same LDS
  // counter), and 10 M ids that are 4 M distinct rows (2.5 ids per run: most buckets take the radix passes)
  { std::vector<int64_t> v(10000000); for (size_t i = 0; i < v.size(); i++) v[i] = static_cast<int64_t>(i) * 10; bad |= run_case("10M ascending / 100M rows", v, 0, 100000000, timing); }
  { std::vector<int64_t> v(10000000); for (auto& x : v) x = static_cast<int64_t>(rng() % 4000000ull) * 25; bad |= run_case("10M ids on 4M distinct rows", v, 0, 100000000, timing); }
  // round 6: the batches the hot mode is for — Zipf(1.05) ids hashed over the table (numpy's rejection sampler), one id everywhere,
  // a hot id + uniform background, ids clustered in a few thousand rows (overflows whatever is peeled)
  {
    const int64_t n = 10000000, N = 100000000;
    std::vector<int64_t> v(n);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    const double a = 1.05, am1 = a - 1.0, b = std::pow(2.0, am1);
    for (auto& x : v) {
      for (;;) {
        const double U = 1.0 - u(rng), V = u(rng);
        const double X = std::floor(std::pow(U, -1.0 / am1));
        if (X < 1.0 || X > 9e18) continue;
        const double T = std::pow(1.0 + 1.0 / X, am1);
        if (V * X * (T - 1.0) / (b - 1.0) <= T / b) { x = static_cast<int64_t>((static_cast<uint64_t>(X) * 2654435761ull) % N); break; }
      }
    }
    bad |= run_case("10M zipf(1.05) hashed / 100M rows", v, 0, N, timing);
    std::vector<int64_t> w(v.begin(), v.begin() + 500000);
    bad |= run_case("0.5M zipf(1.05) hashed / 100M rows", w, 0, N, timing);
    for (auto& x : v) x = x % 3000;
    bad |= run_case("10M zipf ids clustered in 3000 rows", v, 0, N, false);
  }
  bad |= run_case("10M x one id", std::vector<int64_t>(10000000, 4242), 0, 100000000, timing);
  { std::vector<int64_t> v = uniform(3000000, 0, 100000000, 0.02); for (size_t i = 0; i < v.size(); i += 7) v[i] = 31337; for (size_t i = 3; i < v.size(); i += 50) v[i] = 31338;
    bad |= run_case("3M: two hot neighbours + uniform + drops", v, 0, 100000000, false); }
  printf(bad ? "FAILED\n" : "ALL OK\n");
  return bad;
}
