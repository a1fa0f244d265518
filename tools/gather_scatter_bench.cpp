// wholegraph_amd — torch-free gather / scatter benchmark over the C ABI (include/wholememory/*.h + libwholegraph.so).
// Counterpart of the reference's cpp/bench/wholememory_ops/gather_scatter_bench.cu (same option letters, same
// "Bandwidth = gathered bytes / time" convention, one process per GPU) — the program a C++ user of the reference would
// port first. Processes are forked before HIP is touched; rank 0 publishes the communicator's unique id through a
// shared page; every rank fills its shard with the closed form value(row r) = float(r & 0xFFFFFF), draws its own
// uniform ids, and times `loop_count` calls between barriers. Results are verified against the closed form.
//
//   tools/gather_scatter_bench -t chunked -l device -e 51200000000 -g 5120000000 -d 128 -c 20 -f gather -n 1
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include <wholememory/env_func_ptrs.h>
#include <wholememory/tensor_description.h>
#include <wholememory/wholememory.h>
#include <wholememory/wholememory_op.h>
#include <wholememory/wholememory_tensor.h>

namespace {

struct options {
  wholememory_memory_type_t type         = WHOLEMEMORY_MT_CHUNKED;
  wholememory_memory_location_t location = WHOLEMEMORY_ML_DEVICE;
  int64_t table_bytes  = INT64_C(1) << 30;   // -e, whole table
  int64_t gather_bytes = INT64_C(64) << 20;  // -g, per rank
  int64_t dim          = 128;
  int loops            = 20;
  bool scatter         = false;
  int gpus             = 1;
  int candidates       = 1;  // -p, gather / scatter row buffers to choose from (the level follows placement)
};

struct shared_page {
  wholememory_unique_id_t id;
  std::atomic<int> id_ready;
  std::atomic<int> failed;
  double ms_per_call[64];
};

#define HIP_OK(call)                                                                          \
  do {                                                                                        \
    hipError_t e__ = (call);                                                                  \
    if (e__ != hipSuccess) {                                                                  \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
      return 1;                                                                               \
    }                                                                                         \
  } while (0)
#define WM_OK(call)                                                                             \
  do {                                                                                          \
    wholememory_error_code_t e__ = (call);                                                      \
    if (e__ != WHOLEMEMORY_SUCCESS) {                                                           \
      fprintf(stderr, "%s:%d %s -> wholememory error %d\n", __FILE__, __LINE__, #call, (int)e__); \
      return 1;                                                                                 \
    }                                                                                           \
  } while (0)

__global__ void fill_closed_form(float* rows, int64_t first_row, int64_t n_rows, int64_t dim)
{
  const int64_t total = n_rows * dim;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    rows[i] = static_cast<float>((first_row + i / dim) & 0xFFFFFF);
}

int run_rank(const options& o, int rank, shared_page* sh)
{
  int n_dev = 0;
  HIP_OK(hipGetDeviceCount(&n_dev));
  if (n_dev < 1) {
    fprintf(stderr, "no GPU visible\n");
    return 1;
  }
  HIP_OK(hipSetDevice(rank % n_dev));
  WM_OK(wholememory_init(0, LEVEL_WARN));
  if (rank == 0) {
    WM_OK(wholememory_create_unique_id(&sh->id));
    sh->id_ready.store(1);
  }
  while (sh->id_ready.load() == 0) {
    if (sh->failed.load()) return 1;
    usleep(1000);
  }
  wholememory_comm_t comm = nullptr;
  WM_OK(wholememory_create_communicator(&comm, sh->id, rank, o.gpus));

  const int64_t rows = o.table_bytes / (o.dim * 4);
  const int64_t n    = o.gather_bytes / (o.dim * 4);
  int64_t tsz[2]     = {rows, o.dim};
  auto mdesc         = wholememory_create_matrix_desc(tsz, o.dim, 0, WHOLEMEMORY_DT_FLOAT);
  wholememory_tensor_description_t tdesc;
  wholememory_copy_matrix_desc_to_tensor(&tdesc, &mdesc);
  wholememory_tensor_t table = nullptr, local = nullptr;
  // experiments (profiles/r03_scatter_by_vram_offset.txt): WM_BENCH_BLOCKER_GB=a,b,... device allocations made BEFORE the table,
  // so that the table lands further up in VRAM; WM_BENCH_BLOCKER_FREE=1 releases them again right after the table exists
  std::vector<void*> blockers;
  if (const char* be = getenv("WM_BENCH_BLOCKER_GB")) {
    std::string spec(be);
    size_t pos = 0;
    while (pos < spec.size()) {
      size_t comma    = spec.find(',', pos);
      if (comma == std::string::npos) comma = spec.size();
      const double gb = atof(spec.substr(pos, comma - pos).c_str());
      void* b         = nullptr;
      if (gb > 0 && hipMalloc(&b, static_cast<size_t>(gb * 1e9)) == hipSuccess) {
        (void)hipMemset(b, 0, static_cast<size_t>(gb * 1e9));
        blockers.push_back(b);
      }
      pos = comma + 1;
    }
    HIP_OK(hipDeviceSynchronize());
  }
  WM_OK(wholememory_create_tensor(&table, &tdesc, comm, o.type, o.location));
  if (const char* bf = getenv("WM_BENCH_BLOCKER_FREE")) {
    if (bf[0] == '1') {
      for (void* b : blockers) (void)hipFree(b);
      blockers.clear();
    }
  }
  WM_OK(wholememory_tensor_map_local_tensor(table, &local));
  std::vector<size_t> offs(o.gpus + 1);
  WM_OK(wholememory_tensor_get_entry_offsets(offs.data(), table));
  const int64_t my_first = static_cast<int64_t>(offs[rank]), my_rows = static_cast<int64_t>(offs[rank + 1] - offs[rank]);
  float* shard = static_cast<float*>(wholememory_tensor_get_data_pointer(local));
  if (my_rows > 0) {
    if (o.location == WHOLEMEMORY_ML_DEVICE) {
      hipLaunchKernelGGL(fill_closed_form, dim3(4096), dim3(256), 0, nullptr, shard, my_first, my_rows, o.dim);
      HIP_OK(hipDeviceSynchronize());
    } else {
      for (int64_t i = 0; i < my_rows * o.dim; i++) shard[i] = static_cast<float>((my_first + i / o.dim) & 0xFFFFFF);
    }
  }
  WM_OK(wholememory_communicator_barrier(comm));

  std::vector<int64_t> h_idx(n);
  std::mt19937_64 rng(42 + rank);
  std::uniform_int_distribution<int64_t> pick(0, rows - 1);
  for (auto& v : h_idx) v = pick(rng);
  int64_t* d_idx = nullptr;
  float* d_rows  = nullptr;
  HIP_OK(hipMalloc(&d_idx, sizeof(int64_t) * n));
  HIP_OK(hipMalloc(&d_rows, sizeof(float) * n * o.dim));
  HIP_OK(hipMemcpy(d_idx, h_idx.data(), sizeof(int64_t) * n, hipMemcpyHostToDevice));
  HIP_OK(hipMemset(d_rows, 0, sizeof(float) * n * o.dim));
  auto idesc     = wholememory_create_array_desc(n, 0, WHOLEMEMORY_DT_INT64);
  int64_t osz[2] = {n, o.dim};
  auto odesc     = wholememory_create_matrix_desc(osz, o.dim, 0, WHOLEMEMORY_DT_FLOAT);
  wholememory_tensor_description_t it, ot;
  wholememory_copy_array_desc_to_tensor(&it, &idesc);
  wholememory_copy_matrix_desc_to_tensor(&ot, &odesc);
  wholememory_tensor_t idx_t = nullptr, rows_t = nullptr;
  WM_OK(wholememory_make_tensor_from_pointer(&idx_t, d_idx, &it));
  WM_OK(wholememory_make_tensor_from_pointer(&rows_t, d_rows, &ot));
  wholememory_env_func_t* env = wholememory_get_cached_env_func();

  if (o.scatter) {  // scatter rows that equal the closed form: the table stays verifiable, duplicates agree
    WM_OK(wholememory_gather(table, idx_t, rows_t, env, nullptr, -1));
    HIP_OK(hipDeviceSynchronize());
  }
  auto call = [&]() {
    return o.scatter ? wholememory_scatter(rows_t, idx_t, table, env, nullptr, -1)
                     : wholememory_gather(table, idx_t, rows_t, env, nullptr, -1);
  };
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  // -p K: the row buffer is chosen among K allocations, each probed with a few calls of the op about to be timed (the level
  // the memory system serves depends on where the buffer sits relative to the table: DESIGN.md section 3.1); every probe is
  // printed, the first one is what a single allocation gives
  std::vector<float*> spare;
  if (o.candidates > 1) {
    float* best_buf = d_rows;
    float best_ms   = 0;
    for (int k = 0; k < o.candidates; k++) {
      float* buf = d_rows;
      if (k > 0) {
        if (hipMalloc(&buf, sizeof(float) * n * o.dim) != hipSuccess) break;
        HIP_OK(hipMemcpy(buf, d_rows, sizeof(float) * n * o.dim, hipMemcpyDeviceToDevice));
        spare.push_back(buf);
      }
      wholememory_tensor_t cand_t = nullptr;
      WM_OK(wholememory_make_tensor_from_pointer(&cand_t, buf, &ot));
      auto probe = [&]() {
        return o.scatter ? wholememory_scatter(cand_t, idx_t, table, env, nullptr, -1)
                         : wholememory_gather(table, idx_t, cand_t, env, nullptr, -1);
      };
      float pms = 0;
      // WM_BENCH_AB=<variable>: every candidate is also probed with <variable>=0 (e.g. WM_ROWS_INORDER: the persistent launch
      // shape of rounds 1-2 beside the in-order one, on the very same buffers of one process)
      const char* ab_var = getenv("WM_BENCH_AB");
      for (int pass = ab_var != nullptr ? 0 : 1; pass < 2; pass++) {
        if (ab_var != nullptr) {
          if (pass == 0) setenv(ab_var, "0", 1); else unsetenv(ab_var);
        }
        for (int i = 0; i < 2; i++) WM_OK(probe());
        HIP_OK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < 6; i++) WM_OK(probe());
        HIP_OK(hipEventRecord(e1, nullptr));
        HIP_OK(hipEventSynchronize(e1));
        HIP_OK(hipEventElapsedTime(&pms, e0, e1));
        pms /= 6;
        if (rank == 0 && pass == 0) printf("  candidate row buffer %d with %s=0: %.4f ms per call\n", k, ab_var, pms);
      }
      if (rank == 0) printf("  candidate row buffer %d: %.4f ms per call\n", k, pms);
      if (k == 0 || pms < best_ms) best_ms = pms, best_buf = buf;
      WM_OK(wholememory_destroy_tensor(cand_t));
    }
    if (best_buf != d_rows) {   // the timed calls use the chosen buffer
      WM_OK(wholememory_destroy_tensor(rows_t));
      WM_OK(wholememory_make_tensor_from_pointer(&rows_t, best_buf, &ot));
      for (auto& b : spare)
        if (b == best_buf) std::swap(b, d_rows);
    }
    for (auto b : spare) (void)hipFree(b);
    spare.clear();
  }
  for (int i = 0; i < 3; i++) WM_OK(call());
  HIP_OK(hipDeviceSynchronize());
  WM_OK(wholememory_communicator_barrier(comm));
  HIP_OK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < o.loops; i++) WM_OK(call());
  HIP_OK(hipEventRecord(e1, nullptr));
  HIP_OK(hipEventSynchronize(e1));
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  sh->ms_per_call[rank] = ms / o.loops;
  WM_OK(wholememory_communicator_barrier(comm));

  // verify: gather once more (after a scatter this re-reads what was written) and compare with the closed form
  HIP_OK(hipMemset(d_rows, 0xff, sizeof(float) * n * o.dim));
  WM_OK(wholememory_gather(table, idx_t, rows_t, env, nullptr, -1));
  HIP_OK(hipDeviceSynchronize());
  const int64_t probe = n < 4096 ? n : 4096;
  std::vector<float> h_rows(probe * o.dim);
  for (int part = 0; part < 2; part++) {
    const int64_t first = part == 0 ? 0 : n - probe;
    HIP_OK(hipMemcpy(h_rows.data(), d_rows + first * o.dim, sizeof(float) * probe * o.dim, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < probe; i++)
      for (int64_t c = 0; c < o.dim; c++)
        if (h_rows[i * o.dim + c] != static_cast<float>(h_idx[first + i] & 0xFFFFFF)) {
          fprintf(stderr, "rank %d: row %ld col %ld differs from the closed form\n", rank, (long)(first + i), (long)c);
          return 1;
        }
  }
  WM_OK(wholememory_communicator_barrier(comm));
  WM_OK(wholememory_destroy_tensor(idx_t));
  WM_OK(wholememory_destroy_tensor(rows_t));
  WM_OK(wholememory_destroy_tensor(local));
  WM_OK(wholememory_destroy_tensor(table));
  (void)hipFree(d_idx);
  (void)hipFree(d_rows);
  WM_OK(wholememory_destroy_communicator(comm));
  WM_OK(wholememory_finalize());
  return 0;
}

const char* kUsage =
  "usage: %s [options]\n"
  "  -t, --memory_type           continuous | chunked | distributed | hierarchy   (default chunked)\n"
  "  -l, --memory_location       device | host                                    (default device)\n"
  "  -e, --embedding_table_size  bytes of the whole table                         (default 1 GiB)\n"
  "  -g, --gather_size           bytes gathered / scattered per rank per call     (default 64 MiB)\n"
  "  -d, --embedding_dim         fp32 elements per row                            (default 128)\n"
  "  -c, --loop_count            timed calls                                      (default 20)\n"
  "  -p, --placement_candidates  row buffers to choose the fastest from           (default 1: the first allocation)\n"
  "  -f, --test_type             gather | scatter                                 (default gather)\n"
  "  -n, --num_gpu               processes = GPUs of this node                    (default 1)\n";

bool parse(int argc, char** argv, options* o)
{
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto value = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
    if (a == "-h" || a == "--help") return false;
    if (a == "-t" || a == "--memory_type") {
      std::string v = value();
      o->type = v == "continuous" ? WHOLEMEMORY_MT_CONTINUOUS : v == "chunked" ? WHOLEMEMORY_MT_CHUNKED
                : v == "distributed" ? WHOLEMEMORY_MT_DISTRIBUTED : v == "hierarchy" ? WHOLEMEMORY_MT_HIERARCHY
                                                                                     : WHOLEMEMORY_MT_NONE;
      if (o->type == WHOLEMEMORY_MT_NONE) return false;
    } else if (a == "-l" || a == "--memory_location") {
      std::string v = value();
      if (v != "device" && v != "host") return false;
      o->location = v == "device" ? WHOLEMEMORY_ML_DEVICE : WHOLEMEMORY_ML_HOST;
    } else if (a == "-e" || a == "--embedding_table_size") {
      o->table_bytes = atoll(value());
    } else if (a == "-g" || a == "--gather_size") {
      o->gather_bytes = atoll(value());
    } else if (a == "-d" || a == "--embedding_dim") {
      o->dim = atoll(value());
    } else if (a == "-c" || a == "--loop_count") {
      o->loops = atoi(value());
    } else if (a == "-p" || a == "--placement_candidates") {
      o->candidates = std::max(1, atoi(value()));
    } else if (a == "-f" || a == "--test_type") {
      std::string v = value();
      if (v != "gather" && v != "scatter") return false;
      o->scatter = v == "scatter";
    } else if (a == "-n" || a == "--num_gpu") {
      o->gpus = atoi(value());
    } else {
      return false;
    }
  }
  return o->dim > 0 && o->loops > 0 && o->gpus >= 1 && o->gpus <= 64 && o->table_bytes >= o->dim * 4 &&
         o->gather_bytes >= o->dim * 4;
}

}  // namespace

int main(int argc, char** argv)
{
  options o;
  if (!parse(argc, argv, &o)) {
    fprintf(stderr, kUsage, argv[0]);
    return 2;
  }
  void* page = mmap(nullptr, sizeof(shared_page), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (page == MAP_FAILED) return 1;
  auto* sh = new (page) shared_page();
  sh->id_ready.store(0);
  sh->failed.store(0);
  int rc = 0;
  if (o.gpus == 1) {
    rc = run_rank(o, 0, sh);
  } else {
    std::vector<pid_t> kids;
    for (int r = 0; r < o.gpus; r++) {
      pid_t pid = fork();  // before anything touches HIP in this process
      if (pid == 0) {
        int child = run_rank(o, r, sh);
        if (child != 0) sh->failed.store(1);
        _exit(child);
      }
      kids.push_back(pid);
    }
    for (pid_t pid : kids) {
      int status = 0;
      waitpid(pid, &status, 0);
      if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) rc = 1;
    }
  }
  if (rc != 0) {
    fprintf(stderr, "FAILED\n");
    return 1;
  }
  double worst = 0, best = 1e30, sum = 0;
  for (int r = 0; r < o.gpus; r++) {
    worst = sh->ms_per_call[r] > worst ? sh->ms_per_call[r] : worst;
    best  = sh->ms_per_call[r] < best ? sh->ms_per_call[r] : best;
    sum += sh->ms_per_call[r];
  }
  const double row_bytes = static_cast<double>(o.gather_bytes / (o.dim * 4)) * o.dim * 4;
  printf("%s %s: table %.2f GB, %.1f MB per rank per call, dim %ld, %d rank(s), %d calls\n",
         o.scatter ? "scatter" : "gather", o.location == WHOLEMEMORY_ML_DEVICE ? "device" : "host", o.table_bytes / 1e9,
         row_bytes / 1e6, (long)o.dim, o.gpus, o.loops);
  printf("time per call: min %.4f ms, max %.4f ms, avg %.4f ms (over ranks)\n", best, worst, sum / o.gpus);
  printf("Bandwidth: %.2f GB/s per rank, %.2f GB/s total (row bytes / slowest rank's time; verified)\n",
         row_bytes / worst / 1e6, row_bytes * o.gpus / worst / 1e6);
  return 0;
}
