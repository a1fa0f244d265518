/*
 * wholegraph_amd — MI355X-native WholeMemory gather/scatter path.
 * C ABI: the "global reference" a kernel dereferences to reach any rank's rows.
 * Replaces reference cpp/include/wholememory/global_reference.h:32-55 (same layout, same names).
 */
#ifndef WHOLEMEMORY_GLOBAL_REFERENCE_H_
#define WHOLEMEMORY_GLOBAL_REFERENCE_H_

#include <stdbool.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * CONTINUOUS: pointer = base of one flat VA range, stride == 0.
 * CHUNKED   : pointer = device array of world_size per-rank base pointers (hipIpc-mapped peers),
 *             stride  = bytes per rank when every rank but the last holds the same amount
 *             (same_chunk == true → owner = byte_offset / stride), otherwise owner is found in
 *             rank_memory_offsets[world_size + 1] (bytes, device-resident).
 */
struct wholememory_gref_t {
  void* pointer;
  size_t* rank_memory_offsets;
  int world_size;
  size_t stride;
  bool same_chunk;
};
#ifndef __cplusplus
typedef struct wholememory_gref_t wholememory_gref_t;
#endif

/* reference global_reference.h:44-49 */
struct wholememory_gref_t wholememory_create_continuous_global_reference(void* ptr);

#ifdef __cplusplus
}
#endif
#endif
