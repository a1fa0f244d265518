// wholegraph_amd — raw-binary shard I/O for WholeMemory handles ("%s_part_%d_of_%d" files written by
// WholeMemoryEmbedding.save / read back by .load). Reference: cpp/src/wholememory/file_io.cpp:1860
// (load_file_to_handle) and :2059 (store_handle_to_file). Buffered readers only: every rank reads exactly
// the file bytes that land in its own shard (files are logically concatenated and may be re-sharded: any
// number of files, any sizes; entries of file_entry_size bytes are placed at memory_entry_size strides),
// with plain or round-robin placement, and stores its own shard. Reads are multi-threaded (WG_LOAD_THREADS_PER_RANK,
// WG_LOAD_BUFFER_SIZE_MB as in the reference) through pinned double buffers; WG_LOAD_USE_DIRECTIO=1 opens the files with
// O_DIRECT and reads block-aligned windows through per-thread bounce buffers (buffered reads where the file system refuses).
#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include <wholememory/wholememory.h>

#include "knobs.hpp"
#include "backend.hpp"
#include "communicator.hpp"
#include "wm_common.hpp"

namespace wm {
// every rank reports its own outcome; all return the first failure (by rank order), or success. Collective.
static wholememory_error_code_t agree_on_result(wholememory_comm_t comm, wholememory_error_code_t mine)
{
  int code = static_cast<int>(mine);
  std::vector<int> all(static_cast<size_t>(comm->world_size), 0);
  comm->allgather_host(&code, all.data(), sizeof(int));
  for (int r = 0; r < comm->world_size; r++) {
    if (all[r] != WHOLEMEMORY_SUCCESS) {
      if (mine == WHOLEMEMORY_SUCCESS) WM_ERROR("file I/O failed on rank %d (error %d)", r, all[r]);
      return static_cast<wholememory_error_code_t>(all[r]);
    }
  }
  return WHOLEMEMORY_SUCCESS;
}
}  // namespace wm

extern "C" {

wholememory_error_code_t wholememory_load_from_file(wholememory_handle_t handle,
                                                    size_t memory_offset,
                                                    size_t memory_entry_size,
                                                    size_t file_entry_size,
                                                    const char** file_names,
                                                    int file_count,
                                                    int round_robin_size)
{
  WM_API_BEGIN
  if (handle == nullptr || file_names == nullptr || file_count <= 0) return WHOLEMEMORY_INVALID_INPUT;
  wholememory_comm_t comm;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_communicator(&comm, handle));
  // Everything a single rank can fail on (a missing file, a short read, no pinned memory) happens inside `load_local`;
  // the ranks then AGREE on the outcome (one small all-gather, which is also the closing barrier of the reference,
  // file_io.cpp:2047) — a rank that failed no longer leaves the healthy ones waiting in a barrier it never reaches.
  auto load_local = [&]() -> wholememory_error_code_t {
  if (file_entry_size == 0 || file_entry_size > memory_entry_size) return WHOLEMEMORY_INVALID_INPUT;
  if (round_robin_size < 0) return WHOLEMEMORY_INVALID_INPUT;
  const auto* bk = wm::backend();
  std::vector<size_t> file_first(file_count + 1, 0);  // first logical entry of each file
  for (int i = 0; i < file_count; i++) {
    struct stat st;
    if (file_names[i] == nullptr || stat(file_names[i], &st) != 0) {
      WM_ERROR("input_file[%d] of %d cannot be opened for read.", i, file_count);
      return WHOLEMEMORY_INVALID_INPUT;
    }
    if (static_cast<size_t>(st.st_size) % file_entry_size != 0) {
      WM_ERROR("file %s size %zu is not a multiple of the entry size %zu", file_names[i], static_cast<size_t>(st.st_size),
               file_entry_size);
      return WHOLEMEMORY_INVALID_INPUT;
    }
    file_first[i + 1] = file_first[i] + static_cast<size_t>(st.st_size) / file_entry_size;
  }
  const size_t total_entries = file_first[file_count];
  if (memory_offset + file_entry_size > memory_entry_size) return WHOLEMEMORY_INVALID_INPUT;
  if (wholememory_get_data_granularity(handle) % memory_entry_size != 0) {
    WM_ERROR("memory_entry_stride=%zu does not divide the handle granularity %zu", memory_entry_size,
             wholememory_get_data_granularity(handle));
    return WHOLEMEMORY_INVALID_INPUT;  // reference file_io.cpp:1876-1882
  }
  if (total_entries > wholememory_get_total_size(handle) / memory_entry_size) {
    WM_ERROR("files hold %zu entries, the WholeMemory only %zu", total_entries,
             wholememory_get_total_size(handle) / memory_entry_size);
    return WHOLEMEMORY_INVALID_VALUE;  // reference file_io.cpp:1926-1932
  }
  void* local_ptr;
  size_t local_size, local_offset;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_local_memory(&local_ptr, &local_size, &local_offset, handle));
  if (local_offset % memory_entry_size != 0 || local_size % memory_entry_size != 0) return WHOLEMEMORY_INVALID_INPUT;
  const size_t local_first = local_offset / memory_entry_size;
  const size_t local_rows  = local_size / memory_entry_size;

  // Rows are assembled at memory stride in PINNED staging buffers and copied to the shard chunk by chunk; two buffers
  // alternate so that the copy of one chunk overlaps the reads of the next. A chunk that the files do not fully
  // overwrite (padding columns, a column offset, rows without a file entry) is pre-filled from the shard first, so
  // whatever the files do not cover is preserved. The reads of a chunk are split over WG_LOAD_THREADS_PER_RANK threads
  // (reference file_io.cpp:1954; default 8, capped by the chunk's rows) using pread on shared descriptors.
  const size_t chunk_bytes_target = [] {
    const char* e = WM_KNOB("WG_LOAD_BUFFER_SIZE_MB");  // reference file_io.cpp:1975
    const long mb = e != nullptr ? atol(e) : 32;
    return static_cast<size_t>(std::max<long>(mb, 1)) << 20;
  }();
  const size_t kChunkRows = std::max<size_t>(1, chunk_bytes_target / memory_entry_size);
  int n_threads             = 8;
  size_t min_rows_per_thread = 1024;  // below that a thread is not worth starting — unless the caller asked for threads
  if (const char* e = WM_KNOB("WG_LOAD_THREADS_PER_RANK")) {
    n_threads           = std::max(1, atoi(e));
    min_rows_per_thread = 1;
  }
  char* staging[2] = {nullptr, nullptr};
  std::vector<int> fds(file_count, -1);
  auto cleanup = [&]() {
    for (int fd : fds)
      if (fd >= 0) close(fd);
    for (char* p : staging)
      if (p != nullptr) bk->free_pinned(p);
  };
  for (auto& p : staging) {
    void* v = nullptr;
    if (bk->malloc_pinned(&v, kChunkRows * memory_entry_size) != 0) {
      cleanup();
      return WHOLEMEMORY_OUT_OF_MEMORY;
    }
    p = static_cast<char*>(v);
  }
  // WG_LOAD_USE_DIRECTIO=1 (reference file_io.cpp:1975-1979): the files are opened with O_DIRECT and read past the page
  // cache — for feature files larger than host memory. Direct reads need block-aligned offsets, lengths and buffers, and the
  // rows wanted are none of that, so every read goes through a per-thread aligned bounce buffer: the enclosing aligned
  // window is read and the wanted bytes are copied out. A file system that refuses O_DIRECT (tmpfs) falls back to
  // buffered reads with a warning.
  bool direct_io = false;
  if (const char* e = WM_KNOB("WG_LOAD_USE_DIRECTIO")) direct_io = e[0] == '1' && e[1] == 0;
  for (int f = 0; f < file_count; f++) {
    fds[f] = open(file_names[f], direct_io ? (O_RDONLY | O_DIRECT) : O_RDONLY);
    if (fds[f] < 0 && direct_io) {
      WM_WARN("O_DIRECT refused for %s (%s): buffered reads instead", file_names[f], strerror(errno));
      for (int g = 0; g < f; g++) {
        close(fds[g]);
        fds[g] = open(file_names[g], O_RDONLY);
      }
      direct_io = false;
      fds[f]    = open(file_names[f], O_RDONLY);
    }
    if (fds[f] < 0) {
      WM_ERROR("input_file[%d] %s cannot be opened for read.", f, file_names[f]);
      cleanup();
      return WHOLEMEMORY_INVALID_INPUT;
    }
  }
  constexpr size_t kBlock  = 4096;          // alignment of direct reads
  constexpr size_t kBounce = size_t(4) << 20;
  auto pread_all = [direct_io](int fd, char* dst, size_t bytes, off_t off) -> bool {
    if (!direct_io) {
      while (bytes > 0) {
        const ssize_t got = pread(fd, dst, bytes, off);
        if (got <= 0) return false;
        dst += got, off += got, bytes -= static_cast<size_t>(got);
      }
      return true;
    }
    struct bounce_buffer {
      char* p = nullptr;
      bounce_buffer() { if (posix_memalign(reinterpret_cast<void**>(&p), kBlock, kBounce) != 0) p = nullptr; }
      ~bounce_buffer() { free(p); }
    };
    thread_local bounce_buffer bounce;
    if (bounce.p == nullptr) return false;
    while (bytes > 0) {
      const off_t window   = off & ~static_cast<off_t>(kBlock - 1);
      const size_t lead    = static_cast<size_t>(off - window);
      const size_t want    = std::min<size_t>(size_t(4) << 20, (lead + bytes + kBlock - 1) / kBlock * kBlock);
      const ssize_t got    = pread(fd, bounce.p, want, window);   // short only at the end of the file
      if (got <= static_cast<ssize_t>(lead)) return false;
      const size_t usable  = std::min(bytes, static_cast<size_t>(got) - lead);
      memcpy(dst, bounce.p + lead, usable);
      dst += usable, off += static_cast<off_t>(usable), bytes -= usable;
    }
    return true;
  };
  // logical file entries [e0, e1) -> rows of `buf` starting at `row` (thread-safe: pread, no shared cursor)
  auto read_entries = [&](char* buf, size_t e0, size_t e1, size_t row) -> bool {
    int f = 0;
    while (e0 < e1) {
      while (f < file_count && e0 >= file_first[f + 1]) f++;
      if (f >= file_count) return false;
      const size_t n  = std::min(e1, file_first[f + 1]) - e0;
      const off_t off = static_cast<off_t>((e0 - file_first[f]) * file_entry_size);
      if (file_entry_size == memory_entry_size) {
        if (!pread_all(fds[f], buf + row * memory_entry_size, n * file_entry_size, off)) return false;
      } else {
        for (size_t i = 0; i < n; i++)
          if (!pread_all(fds[f], buf + (row + i) * memory_entry_size + memory_offset, file_entry_size,
                         off + static_cast<off_t>(i * file_entry_size)))
            return false;
      }
      e0 += n;
      row += n;
    }
    return true;
  };
  const size_t W = static_cast<size_t>(comm->world_size), rank = static_cast<size_t>(comm->world_rank);
  const size_t rr = static_cast<size_t>(round_robin_size);
  // file entries of local rows [l0, l1) of a chunk that starts at local row c: returns false on I/O error
  auto read_rows = [&](char* buf, size_t c, size_t l0, size_t l1) -> bool {
    if (rr == 0) {
      const size_t e0 = local_first + l0, e1 = std::min(local_first + l1, total_entries);
      return e0 >= e1 || read_entries(buf, e0, e1, l0 - c);
    }
    // round-robin placement (the inverse of map_indices_func.cu:34-43): local row l of rank r holds file entry
    // ((l / rr) * W + r) * rr + l % rr; rows past the end of the files are left as they are
    for (size_t l = l0; l < l1;) {
      const size_t run_end = std::min(l1, (l / rr + 1) * rr);
      const size_t e0      = ((l / rr) * W + rank) * rr + l % rr;
      const size_t e1      = std::min(e0 + (run_end - l), total_entries);
      if (e0 < e1 && !read_entries(buf, e0, e1, l - c)) return false;
      l = run_end;
    }
    return true;
  };
  // does the chunk [c, c + n) receive a file entry in EVERY byte of every row?
  auto fully_overwritten = [&](size_t c, size_t n) {
    if (file_entry_size != memory_entry_size || memory_offset != 0) return false;
    if (rr == 0) return local_first + c + n <= total_entries;
    for (size_t l = c; l < c + n; l = (l / rr + 1) * rr) {
      const size_t run_end = std::min(c + n, (l / rr + 1) * rr);
      if (((l / rr) * W + rank) * rr + l % rr + (run_end - l) > total_entries) return false;
    }
    return true;
  };
  bool in_flight[2] = {false, false};
  bool failed_io = false, failed_dev = false;
  int which = 0;
  for (size_t c = 0; c < local_rows && !failed_io && !failed_dev; c += kChunkRows, which ^= 1) {
    const size_t n = std::min(kChunkRows, local_rows - c);
    char* dst      = static_cast<char*>(local_ptr) + c * memory_entry_size;
    char* buf      = staging[which];
    // this buffer's previous copy (two chunks ago) must have left it; copies are issued on the null stream in order
    if (in_flight[which] || !fully_overwritten(c, n)) {
      if (bk->stream_sync(nullptr) != 0) failed_dev = true;
      in_flight[0] = in_flight[1] = false;
    }
    if (!failed_dev && !fully_overwritten(c, n)) {
      if (bk->memcpy_async(buf, dst, n * memory_entry_size, nullptr) != 0 || bk->stream_sync(nullptr) != 0) failed_dev = true;
    }
    if (failed_dev) break;
    const int T = static_cast<int>(std::min<size_t>(static_cast<size_t>(n_threads), std::max<size_t>(1, n / min_rows_per_thread)));
    if (T <= 1) {
      failed_io = !read_rows(buf, c, c, c + n);
    } else {
      std::vector<std::thread> workers;
      std::vector<char> ok(T, 1);
      for (int t = 0; t < T; t++) {
        const size_t l0 = c + n * t / T, l1 = c + n * (t + 1) / T;
        workers.emplace_back([&, t, l0, l1] { ok[t] = read_rows(buf, c, l0, l1) ? 1 : 0; });
      }
      for (auto& w : workers) w.join();
      for (char o : ok) failed_io |= (o == 0);
    }
    if (failed_io) break;
    if (bk->memcpy_async(dst, buf, n * memory_entry_size, nullptr) != 0) failed_dev = true;
    in_flight[which] = true;
  }
  if (bk->stream_sync(nullptr) != 0) failed_dev = true;
  cleanup();
  if (failed_io) return WHOLEMEMORY_SYSTEM_ERROR;
  if (failed_dev) return WHOLEMEMORY_CUDA_ERROR;
  return WHOLEMEMORY_SUCCESS;
  };  // load_local
  return wm::agree_on_result(comm, load_local());
  WM_API_END
}

wholememory_error_code_t wholememory_store_to_file(wholememory_handle_t handle,
                                                   size_t memory_offset,
                                                   size_t memory_entry_stride,
                                                   size_t file_entry_size,
                                                   const char* local_file_name)
{
  WM_API_BEGIN
  if (handle == nullptr || local_file_name == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  wholememory_comm_t comm;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_communicator(&comm, handle));
  // reference file_io.cpp:2059-2075: peers may still be scattering into this shard — wait for everybody first
  comm->barrier();
  auto store_local = [&]() -> wholememory_error_code_t {
  if (file_entry_size == 0 || memory_offset + file_entry_size > memory_entry_stride) return WHOLEMEMORY_INVALID_INPUT;
  if (wholememory_get_data_granularity(handle) % memory_entry_stride != 0) {
    WM_ERROR("memory_entry_stride=%zu does not divide the handle granularity %zu", memory_entry_stride,
             wholememory_get_data_granularity(handle));
    return WHOLEMEMORY_INVALID_INPUT;  // reference file_io.cpp:2076-2082
  }
  const auto* bk = wm::backend();
  void* local_ptr;
  size_t local_size, local_offset;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_local_memory(&local_ptr, &local_size, &local_offset, handle));
  if (local_size % memory_entry_stride != 0) return WHOLEMEMORY_INVALID_INPUT;
  const size_t entries = local_size / memory_entry_stride;
  FILE* fp             = fopen(local_file_name, "wb");
  if (fp == nullptr) return WHOLEMEMORY_SYSTEM_ERROR;
  constexpr size_t kChunkEntries = 1 << 14;
  std::vector<char> staging(kChunkEntries * memory_entry_stride);
  for (size_t c = 0; c < entries; c += kChunkEntries) {
    const size_t n = std::min(kChunkEntries, entries - c);
    if (bk->memcpy_async(staging.data(), static_cast<char*>(local_ptr) + c * memory_entry_stride, n * memory_entry_stride,
                         nullptr) != 0 ||
        bk->stream_sync(nullptr) != 0) {
      fclose(fp);
      return WHOLEMEMORY_CUDA_ERROR;
    }
    for (size_t i = 0; i < n; i++) {
      if (fwrite(staging.data() + i * memory_entry_stride + memory_offset, file_entry_size, 1, fp) != 1) {
        fclose(fp);
        return WHOLEMEMORY_SYSTEM_ERROR;
      }
    }
  }
  if (fclose(fp) != 0) return WHOLEMEMORY_SYSTEM_ERROR;
  return WHOLEMEMORY_SUCCESS;
  };  // store_local
  return wm::agree_on_result(comm, store_local());
  WM_API_END
}

}  // extern "C"
