// wholegraph_amd — raw-binary shard I/O for WholeMemory handles ("%s_part_%d_of_%d" files written by
// WholeMemoryEmbedding.save / read back by .load). Reference: cpp/src/wholememory/file_io.cpp:1860
// (load_file_to_handle) and :2059 (store_handle_to_file). This build implements the plain buffered
// path: every rank reads exactly the file bytes that land in its own shard (files are logically
// concatenated, entries of file_entry_size bytes are placed at memory_entry_size strides), and
// stores its own shard; O_DIRECT / multi-thread / round-robin readers of the reference are not built.
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
  if (round_robin_size != 0) {
    WM_ERROR("wholememory_load_from_file: round-robin placement is not implemented in this build");
    return WHOLEMEMORY_NOT_IMPLEMENTED;
  }
  const auto* bk = wm::backend();
  std::vector<size_t> file_entries(file_count);
  size_t total_entries = 0;
  for (int i = 0; i < file_count; i++) {
    struct stat st;
    if (stat(file_names[i], &st) != 0) {
      WM_ERROR("cannot stat %s", file_names[i]);
      return WHOLEMEMORY_INVALID_INPUT;
    }
    if (static_cast<size_t>(st.st_size) % file_entry_size != 0) {
      WM_ERROR("file %s size %zu is not a multiple of the entry size %zu", file_names[i], static_cast<size_t>(st.st_size),
               file_entry_size);
      return WHOLEMEMORY_INVALID_INPUT;
    }
    file_entries[i] = static_cast<size_t>(st.st_size) / file_entry_size;
    total_entries += file_entries[i];
  }
  void* local_ptr;
  size_t local_size, local_offset;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_local_memory(&local_ptr, &local_size, &local_offset, handle));
  if (memory_offset >= memory_entry_size || memory_offset + file_entry_size > memory_entry_size) return WHOLEMEMORY_INVALID_INPUT;
  if (local_offset % memory_entry_size != 0 || local_size % memory_entry_size != 0) return WHOLEMEMORY_INVALID_INPUT;
  const size_t first = local_offset / memory_entry_size;
  const size_t last  = std::min(first + local_size / memory_entry_size, total_entries);
  constexpr size_t kChunkEntries = 1 << 14;
  std::vector<char> staging(kChunkEntries * memory_entry_size);
  size_t file_first = 0;
  for (int f = 0; f < file_count && first < last; f++) {
    const size_t file_last = file_first + file_entries[f];
    const size_t s = std::max(first, file_first), e = std::min(last, file_last);
    if (s < e) {
      FILE* fp = fopen(file_names[f], "rb");
      if (fp == nullptr) return WHOLEMEMORY_SYSTEM_ERROR;
      fseeko(fp, static_cast<off_t>((s - file_first) * file_entry_size), SEEK_SET);
      for (size_t c = s; c < e; c += kChunkEntries) {
        const size_t n = std::min(kChunkEntries, e - c);
        char* dst      = static_cast<char*>(local_ptr) + (c - first) * memory_entry_size;
        if (file_entry_size == memory_entry_size) {
          if (fread(staging.data(), file_entry_size, n, fp) != n) {
            fclose(fp);
            return WHOLEMEMORY_SYSTEM_ERROR;
          }
          if (bk->memcpy_async(dst, staging.data(), n * memory_entry_size, nullptr) != 0 || bk->stream_sync(nullptr) != 0) {
            fclose(fp);
            return WHOLEMEMORY_CUDA_ERROR;
          }
        } else {
          // narrower file rows: read-modify-write so padding / neighbouring columns are preserved
          if (bk->memcpy_async(staging.data(), dst, n * memory_entry_size, nullptr) != 0 || bk->stream_sync(nullptr) != 0) {
            fclose(fp);
            return WHOLEMEMORY_CUDA_ERROR;
          }
          for (size_t i = 0; i < n; i++) {
            if (fread(staging.data() + i * memory_entry_size + memory_offset, file_entry_size, 1, fp) != 1) {
              fclose(fp);
              return WHOLEMEMORY_SYSTEM_ERROR;
            }
          }
          if (bk->memcpy_async(dst, staging.data(), n * memory_entry_size, nullptr) != 0 || bk->stream_sync(nullptr) != 0) {
            fclose(fp);
            return WHOLEMEMORY_CUDA_ERROR;
          }
        }
      }
      fclose(fp);
    }
    file_first = file_last;
  }
  wholememory_comm_t comm;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_communicator(&comm, handle));
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
