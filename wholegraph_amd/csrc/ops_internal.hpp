// wholegraph_amd — internals shared by ops.cpp and embedding.cpp (host orchestration helpers).
#pragma once

#include <vector>

#include <wholememory/wholegraph_amd_ext.h>
#include <wholememory/wholememory_op.h>

#include "backend.hpp"
#include "communicator.hpp"
#include "wm_common.hpp"

namespace wm {

// One scratch allocation obtained through the caller's env functions (reference
// wholememory_ops/temp_memory_handle.hpp:23-93): create ctx -> malloc -> free -> destroy ctx.
class temp_mem {
 public:
  explicit temp_mem(wholememory_env_func_t* env);
  ~temp_mem();
  temp_mem(const temp_mem&)            = delete;
  temp_mem& operator=(const temp_mem&) = delete;
  void* alloc(int64_t elt_count, wholememory_dtype_t dtype, wholememory_memory_allocation_type_t type);
  void* device(int64_t n, wholememory_dtype_t dt) { return alloc(n, dt, WHOLEMEMORY_MA_DEVICE); }
  void* pinned(int64_t n, wholememory_dtype_t dt) { return alloc(n, dt, WHOLEMEMORY_MA_PINNED); }
  void* host(int64_t n, wholememory_dtype_t dt) { return alloc(n, dt, WHOLEMEMORY_MA_HOST); }
  void* get() const { return ptr_; }

 private:
  wholememory_env_func_t* env_;
  void* ctx_ = nullptr;
  void* ptr_ = nullptr;
};

// Result of bucket + exchange of lookup ids (reference bucket_and_exchange_ids_func,
// functions/exchange_ids_nccl_func.cu:157-226).
struct id_exchange {
  explicit id_exchange(wholememory_env_func_t* env)
    : bucketed_mem(env), raw_mem(env), recv_mem(env), aux_offsets(env), aux_counts(env), aux_ws(env)
  {
  }
  std::vector<int64_t> send_counts, recv_counts;    // per peer, as they travel (self = 0 when kept local)
  std::vector<int64_t> send_offsets, recv_offsets;  // exclusive prefix of the above, W+1
  std::vector<int64_t> bucket_offsets;              // W+1: where each owner's segment starts in bucketed_ids
  int64_t total_send  = 0;                          // ids that travel
  int64_t total_recv  = 0;                          // ids received from peers
  int64_t total_valid = 0;                          // non-negative ids of this rank (all owners)
  int64_t global_moved = 0;                         // ids that change rank, summed over ALL ranks (same on every rank)
  // one rank and not a single negative id: nothing was moved or dropped — bucketed_ids IS the caller's array and
  // raw_indices (null) stands for the identity. Only produced when the caller asked for it (allow_identity)
  bool identity = false;
  // the ids were handed over sorted and distinct (sorted_unique): bucketed_ids IS that array, owner segments are its
  // contiguous pieces and raw_indices (null) stands for the identity — rows can be received straight into a dense
  // [total_valid, dim] buffer in bucketed order, with no reorder pass
  bool presorted = false;
  // mean over the ranks of the duplicate estimate (permille of the sampled ids), when bucket_and_exchange_ids was asked
  // for it; the same number on every rank
  int64_t dup_permille = 0;
  int64_t self_count  = 0;                          // ids of this rank that it owns itself
  int64_t self_offset = 0;                          // their position in bucketed_ids / raw_indices
  void* bucketed_ids   = nullptr;                   // [n]   ids grouped by owner (index dtype)
  int64_t* raw_indices = nullptr;                   // [n]   original position of each grouped id
  void* recv_ids       = nullptr;                   // [total_recv] ids received, peer-major
  temp_mem bucketed_mem, raw_mem, recv_mem;
  // a deferred exchange (bucket_and_exchange_ids(..., defer_ids) then finish_id_exchange): row offsets on the device and the
  // bucketing workspace with the scanned block counts, kept from the first half for the second; aux_counts: its counts dummy
  temp_mem aux_offsets, aux_counts, aux_ws;
};

// ids that are already sorted (as unsigned keys) and distinct, their number still on the device: what dedup_ids leaves
struct sorted_unique {
  const int64_t* n_dev;  // device: number of ids (<= the n passed to bucket_and_exchange_ids)
  // optional (device): this rank's vote, carried in the spare slot of the counts exchange where the duplicate estimate of a
  // gather rides — id_exchange::dup_permille is then the mean of the ranks' votes, or -1 when any rank voted below zero (a
  // veto: every rank learns it from the same W numbers and takes the same way out)
  const int64_t* vote_dev = nullptr;
};
void bucket_and_exchange_ids(wholememory_comm_t comm, const void* indices, wholememory_dtype_t index_dtype, int64_t n,
                             const std::vector<size_t>& entry_offsets, wholememory_env_func_t* env, void* stream,
                             id_exchange* x, bool keep_self_local = false, bool allow_identity = false,
                             const sorted_unique* sorted = nullptr, bool estimate_duplicates = false,
                             bool defer_ids = false);
void finish_id_exchange(wholememory_comm_t comm, const void* indices, wholememory_dtype_t index_dtype, int64_t n,
                        const std::vector<size_t>& entry_offsets, wholememory_env_func_t* env, void* stream, id_exchange* x);

// all-to-all-v of fixed-size rows with explicit per-peer row offsets on both sides
void exchange_segments(wholememory_comm_t comm, const void* send, const std::vector<int64_t>& send_counts,
                       const std::vector<int64_t>& send_offsets, void* recv, const std::vector<int64_t>& recv_counts,
                       const std::vector<int64_t>& recv_offsets, size_t row_bytes, void* stream);

// all-to-all-v of fixed-size rows: counts in rows, peer-major contiguous on both sides
void exchange_rows(wholememory_comm_t comm, const void* send, const std::vector<int64_t>& send_counts, void* recv,
                   const std::vector<int64_t>& recv_counts, size_t row_bytes, void* stream);

// number of row-chunks the rows all-to-all-v is pipelined in (WM_EXCHANGE_CHUNKS overrides; 1 = no pipelining);
// global_moved = id_exchange::global_moved, so that every rank decides alike
int exchange_chunks(int world_size, int64_t global_moved);

// Chunk-major order of per-peer segments (backend.hpp: permute_chunks): chunk c of a segment of n rows is rows
// [n*c/C, n*(c+1)/C) of it — the same cut on both ends of a pair, so the sizes of an exchanged chunk always match — and the
// chunk-major order lists chunk 0 of every segment (in peer order), then chunk 1 of every segment, ... A chunk of the
// pipelined exchange is then ONE contiguous range of ids, positions and row buffer: one row kernel per chunk and side
// whatever the number of ranks (distributed gather since round 5; distributed scatter and gradient apply since round 6).
struct chunk_layout {
  chunk_layout(const std::vector<int64_t>& counts, int n_chunks) : counts_(counts), C_(n_chunks), start_(n_chunks + 1, 0)
  {
    for (int c = 0; c < C_; c++) {
      int64_t s = 0;
      for (size_t p = 0; p < counts_.size(); p++) s += count(c, static_cast<int>(p));
      start_[c + 1] = start_[c] + s;
    }
  }
  int64_t first(int c, int p) const { return counts_[p] * c / C_; }                 // first row of chunk c inside segment p
  int64_t count(int c, int p) const { return counts_[p] * (c + 1) / C_ - counts_[p] * c / C_; }
  int64_t start(int c) const { return start_[c]; }                                   // where chunk c begins, chunk-major
  int64_t size(int c) const { return start_[c + 1] - start_[c]; }
  int64_t pos(int c, int p) const                                                    // where chunk c of segment p begins
  {
    int64_t at = start_[c];
    for (int q = 0; q < p; q++) at += count(c, q);
    return at;
  }
  int64_t total() const { return start_[C_]; }

 private:
  std::vector<int64_t> counts_;
  int C_;
  std::vector<int64_t> start_;
};

// RAII bundle of backend events
class event_set {
 public:
  explicit event_set(int n);
  ~event_set();
  void* operator[](int i) const { return events_[i]; }

 private:
  std::vector<void*> events_;
};

// row offsets [W+1] of a handle whose rows are entry_bytes wide
std::vector<size_t> entry_offsets_of(wholememory_handle_t handle, size_t entry_bytes);

// flat gref through which GLOBAL row ids address this rank's shard
wholememory_gref_t local_shard_gref(wholememory_handle_t handle);

// gref a kernel should use for a tensor mapped in this process (CONTINUOUS / CHUNKED handle or a plain pointer)
wholememory_error_code_t tensor_mapped_gref(wholememory_tensor_t t, wholememory_gref_t* gref);

struct row_cache;
// gather of an embedding with a device row cache (embedding_cache.hpp)
wholememory_error_code_t gather_cached(wholememory_tensor_t table, wholememory_tensor_t indices_tensor,
                                       wholememory_tensor_t output_tensor, wholememory_env_func_t* env, void* stream,
                                       int gather_sms, row_cache* cache, bool adjust_cache);

// true when a CHUNKED / CONTINUOUS table should be served through the all-to-all-v route (WM_MAPPED_VIA_EXCHANGE=1)
bool mapped_via_exchange(wholememory_tensor_t t, wholememory_memory_type_t mt);

}  // namespace wm
