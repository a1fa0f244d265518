// wholegraph_amd — raw-binary shard I/O for WholeMemory handles ("%s_part_%d_of_%d" files written by
// WholeMemoryEmbedding.save / read back by .load). Reference: cpp/src/wholememory/file_io.cpp:1860
// (load_file_to_handle) and :2059 (store_handle_to_file). Buffered readers only: every rank reads exactly
// the file bytes that land in its own shard (files are logically concatenated and may be re-sharded: any
// number of files, any sizes; entries of file_entry_size bytes are placed at memory_entry_size strides),
// with plain or round-robin placement, and stores its own shard. The reference's O_DIRECT and multi-threaded
// variants of the same readers (WG_LOAD_USE_DIRECTIO, WG_LOAD_THREADS_PER_RANK) are not built.
#include <cstdio>
#include <cstring>
#include <sys/stat.h>
#include <vector>

#include <wholememory/wholememory.h>

#include "backend.hpp"
#include "communicator.hpp"
#include "wm_common.hpp"

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
  wholememory_comm_t comm;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_communicator(&comm, handle));
  void* local_ptr;
  size_t local_size, local_offset;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_local_memory(&local_ptr, &local_size, &local_offset, handle));
  if (local_offset % memory_entry_size != 0 || local_size % memory_entry_size != 0) return WHOLEMEMORY_INVALID_INPUT;
  const size_t local_first = local_offset / memory_entry_size;
  const size_t local_rows  = local_size / memory_entry_size;

  // rows are assembled at memory stride in a host staging buffer (pre-filled from the shard, so padding,
  // neighbouring columns and rows without a file entry are preserved) and copied back chunk by chunk
  constexpr size_t kChunkRows = 1 << 14;
  std::vector<char> staging(kChunkRows * memory_entry_size);
  std::vector<FILE*> fps(file_count, nullptr);
  auto close_all = [&]() {
    for (auto* f : fps)
      if (f) fclose(f);
  };
  // logical file entries [e0, e1) -> staging rows starting at `row`
  auto read_entries = [&](size_t e0, size_t e1, size_t row) -> bool {
    int f = 0;
    while (e0 < e1) {
      while (f < file_count && e0 >= file_first[f + 1]) f++;
      if (f >= file_count) return false;
      if (fps[f] == nullptr && (fps[f] = fopen(file_names[f], "rb")) == nullptr) return false;
      const size_t n = std::min(e1, file_first[f + 1]) - e0;
      if (fseeko(fps[f], static_cast<off_t>((e0 - file_first[f]) * file_entry_size), SEEK_SET) != 0) return false;
      if (file_entry_size == memory_entry_size) {
        if (fread(staging.data() + row * memory_entry_size, file_entry_size, n, fps[f]) != n) return false;
      } else {
        for (size_t i = 0; i < n; i++)
          if (fread(staging.data() + (row + i) * memory_entry_size + memory_offset, file_entry_size, 1, fps[f]) != 1) return false;
      }
      e0 += n;
      row += n;
    }
    return true;
  };
  const size_t W = static_cast<size_t>(comm->world_size), rank = static_cast<size_t>(comm->world_rank);
  const size_t rr = static_cast<size_t>(round_robin_size);
  for (size_t c = 0; c < local_rows; c += kChunkRows) {
    const size_t n = std::min(kChunkRows, local_rows - c);
    char* dst      = static_cast<char*>(local_ptr) + c * memory_entry_size;
    if (bk->memcpy_async(staging.data(), dst, n * memory_entry_size, nullptr) != 0 || bk->stream_sync(nullptr) != 0) {
      close_all();
      return WHOLEMEMORY_CUDA_ERROR;
    }
    bool touched = false;
    if (rr == 0) {
      const size_t e0 = local_first + c, e1 = std::min(local_first + c + n, total_entries);
      if (e0 < e1) {
        touched = true;
        if (!read_entries(e0, e1, 0)) {
          close_all();
          return WHOLEMEMORY_SYSTEM_ERROR;
        }
      }
    } else {
      // round-robin placement (the inverse of map_indices_func.cu:34-43): local row l of rank r holds file entry
      // ((l / rr) * W + r) * rr + l % rr; rows past the end of the files are left as they are
      for (size_t l = c; l < c + n;) {
        const size_t run_end = std::min(c + n, (l / rr + 1) * rr);
        const size_t e0      = ((l / rr) * W + rank) * rr + l % rr;
        const size_t e1      = std::min(e0 + (run_end - l), total_entries);
        if (e0 < e1) {
          touched = true;
          if (!read_entries(e0, e1, l - c)) {
            close_all();
            return WHOLEMEMORY_SYSTEM_ERROR;
          }
        }
        l = run_end;
      }
    }
    if (!touched) continue;
    if (bk->memcpy_async(dst, staging.data(), n * memory_entry_size, nullptr) != 0 || bk->stream_sync(nullptr) != 0) {
      close_all();
      return WHOLEMEMORY_CUDA_ERROR;
    }
  }
  close_all();
  comm->barrier();
  return WHOLEMEMORY_SUCCESS;
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
  if (file_entry_size == 0 || memory_offset + file_entry_size > memory_entry_stride) return WHOLEMEMORY_INVALID_INPUT;
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
  fclose(fp);
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

}  // extern "C"
