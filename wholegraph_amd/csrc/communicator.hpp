// wholegraph_amd — communicator: rank bookkeeping + the collective transport.
// Replaces the reference's wholememory_comm_ / nccl_comms pair (cpp/src/wholememory/communicator.hpp:38-232,
// nccl_comms.cpp:82-515). Transport on MI355X is RCCL over xGMI; a world_size==1 communicator needs
// no transport at all; an external provider (wholegraph_amd_ext.h) can be plugged for tests/hosts.
#pragma once

#include <cstddef>
#include <memory>
#include <mutex>
#include <vector>

#include <wholememory/wholegraph_amd_ext.h>
#include <wholememory/wholememory.h>

namespace wm {

class collective_provider {
 public:
  virtual ~collective_provider() = default;
  virtual const char* name() const = 0;
  // ranks the transport itself reports (ncclCommCount for RCCL); -1 when the provider cannot tell
  virtual int transport_ranks() const { return -1; }
  virtual void barrier() = 0;
  // host buffers: recv[r*bytes..] = rank r's send
  virtual void allgather_host(const void* send, void* recv, size_t bytes) = 0;
  // the same on DEVICE buffers, enqueued on `stream` (no host round trip); false = this provider cannot (the caller
  // then goes through allgather_host)
  virtual bool allgather_device(const void* /*send*/, void* /*recv*/, size_t /*bytes*/, void* /*stream*/) { return false; }
  // device buffers, byte counts/displacements indexed by peer rank; enqueued on `stream`
  virtual void alltoallv_device(const void* send, const size_t* send_bytes, const size_t* send_disp, void* recv,
                                const size_t* recv_bytes, const size_t* recv_disp, void* stream) = 0;
  // new provider for the sub-group {ranks with the same color}, ordered by (key, old rank); `members` receives the
  // parent ranks of the sub-group in their new order. Returns nullptr for color == WHOLEMEMORY_SPILT_NO_COLOR
  virtual std::unique_ptr<collective_provider> split(int color, int key, int my_rank, int* new_rank, int* new_size,
                                                     std::vector<int>* members) = 0;
};

}  // namespace wm

struct wholememory_comm_ {
  int world_rank = 0;
  int world_size = 1;
  int local_size = 1;  // ranks of this communicator that run on this rank's node
  // node index of every rank (nodes numbered in order of their first rank); filled by detect_nodes() at creation
  // (reference communicator.cpp:405-500,548-580 exchanges host names / boot ids to the same end)
  std::vector<int> node_of_rank{0};
  // every node holds the same number of consecutive ranks: rank r is local rank r % local_size of node r / local_size
  // (what the HIERARCHY memory type needs)
  bool regular_nodes = true;
  void detect_nodes();
  void adopt_nodes(const wholememory_comm_& parent, const std::vector<int>& members);
  int comm_id    = 0;
  wholememory_distributed_backend_t distributed_backend = WHOLEMEMORY_DB_NCCL;
  std::unique_ptr<wm::collective_provider> transport;  // null when world_size == 1
  // WM_EXCHANGE_SELF=1 (bring-up / tests): the distributed ops send this rank's OWN segment through the transport like
  // any peer's instead of serving it locally, and a single-rank communicator that has a transport (WM_FORCE_RCCL=1)
  // runs the whole bucket -> counts -> all-to-all-v route. This is how the RCCL provider is exercised on a one-GPU box.
  bool loopback = false;
  // true when a one-rank communicator may skip bucketing and exchange altogether
  bool single_rank_direct() const { return world_size == 1 && !(loopback && transport != nullptr); }
  std::mutex mu;                                       // guards handle create/destroy (reference communicator.hpp:226)
  int live_handles = 0;
  void* side_stream = nullptr;                         // lazily created: carries the chunked all-to-all-v
  void* get_side_stream();
  ~wholememory_comm_();

  void barrier();
  void allgather_host(const void* send, void* recv, size_t bytes);
  // counts exchange: recv[r] = what rank r sends to me (reference host_alltoall, nccl_comms.cpp:383-407)
  // between_ranks (optional): sum of the off-diagonal of the whole matrix — the same number on every rank, which makes it
  // a safe input for decisions all ranks must take alike
  // `extra` (optional, in: this rank's value, out: [world_size] every rank's value) rides along with the counts: one more
  // int64 per rank in the same collective (the duplicate estimate behind the de-duplication decision)
  void alltoall_host_i64(const int64_t* send, int64_t* recv, int64_t* between_ranks = nullptr, int64_t* extra = nullptr);
  // The same exchange starting from counts that are still on the DEVICE, with ONE host synchronisation in total: the
  // W x W matrix is all-gathered on `stream` by the transport, copied to `pinned_matrix` (W * W int64, pinned) and the
  // stream is drained once. send / recv receive this rank's row / column. Returns false when the transport has no
  // device all-gather (external providers): the caller then copies its counts to the host and uses alltoall_host_i64.
  // dev_counts holds W + 1 values (the counts and one extra value), the matrices W * (W + 1); `extra` receives every
  // rank's extra value ([world_size], may be nullptr)
  bool alltoall_counts_device(const int64_t* dev_counts, int64_t* dev_matrix, int64_t* pinned_matrix, void* stream,
                              int64_t* send, int64_t* recv, int64_t* between_ranks, int64_t* extra = nullptr);
  void alltoallv_device(const void* send, const size_t* send_bytes, const size_t* send_disp, void* recv,
                        const size_t* recv_bytes, const size_t* recv_disp, void* stream);
};
