// wholegraph_amd — shared host-side plumbing: logging, error→code mapping, small integer helpers.
// Mirrors the reference's conventions (cpp/src/logger.hpp:70-87, cpp/src/error.hpp:31-147,
// cpp/src/cuda_macros.hpp:39-164) with HIP underneath: exceptions never cross the C ABI; every
// extern "C" entry wraps its body in WM_API_BEGIN / WM_API_END.
#pragma once

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include <wholememory/wholememory.h>

namespace wm {

LogLevel& log_level_ref();
void log_message(LogLevel lvl, const char* file, int line, const char* fmt, ...)
  __attribute__((format(printf, 4, 5)));

struct logic_error : std::logic_error {
  using std::logic_error::logic_error;
};
struct hip_error : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct comm_error : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct invalid_input : std::invalid_argument {
  using std::invalid_argument::invalid_argument;
};

std::string format_string(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

template <typename T>
inline T div_up(T a, T b)
{
  return (a + b - 1) / b;
}
template <typename T>
inline T round_up(T a, T b)
{
  return div_up(a, b) * b;
}

// WM_DEBUG_SYNC=1 makes ops synchronise + check after every stage (reference cuda_macros.cpp:30).
bool debug_sync_enabled();

// Completion semantics of the ops whose reference versions synchronise the stream before they return (neighbour sampling,
// append_unique: unweighted_sample_without_replacement_func.cuh:474, append_unique_func.cuh:351). Off (the default): the
// same here — outputs are complete and scratch is idle when the call returns, whatever env functions the caller
// supplied. On: such ops return with their last kernels queued; legal only when every allocator behind p_env_fns is
// stream-ordered on the op's stream and every consumer of the outputs is ordered on it too — the torch layer declares
// that (wholememory_ext_set_async_completion(1)); WM_ASYNC_OPS=0/1 overrides either way.
void set_async_completion(bool on);
bool async_completion_enabled();
// sorted-ids gather of HOST tables (ops.cpp: wholememory_gather): smallest batch that takes it (0: route off,
// WM_HOST_SORTED_GATHER=0 / WM_HOST_SORTED_MIN) and the lowest id bit the sort looks at (WM_HOST_SORTED_LOW_BIT)
int64_t host_sorted_gather_min();
int host_sorted_gather_low_bit();

}  // namespace wm

#define WM_LOG(lvl, ...)                                                              \
  do {                                                                                \
    if ((lvl) <= ::wm::log_level_ref()) ::wm::log_message((lvl), __FILE__, __LINE__, __VA_ARGS__); \
  } while (0)
#define WM_ERROR(...) WM_LOG(LEVEL_ERROR, __VA_ARGS__)
#define WM_WARN(...) WM_LOG(LEVEL_WARN, __VA_ARGS__)
#define WM_INFO(...) WM_LOG(LEVEL_INFO, __VA_ARGS__)
#define WM_DEBUG(...) WM_LOG(LEVEL_DEBUG, __VA_ARGS__)

// throw wm::logic_error when cond is false
#define WM_CHECK(cond, ...)                                                                     \
  do {                                                                                          \
    if (!(cond)) {                                                                              \
      throw ::wm::logic_error(::wm::format_string("%s:%d check `%s` failed: ", __FILE__, __LINE__, #cond) + \
                              ::wm::format_string(__VA_ARGS__));                                \
    }                                                                                           \
  } while (0)

// reference *_NOTHROW checks abort the process (error.hpp:86-95)
#define WM_CHECK_ABORT(cond, ...)                                                     \
  do {                                                                                \
    if (!(cond)) {                                                                    \
      ::wm::log_message(LEVEL_FATAL, __FILE__, __LINE__, "check `%s` failed", #cond); \
      ::wm::log_message(LEVEL_FATAL, __FILE__, __LINE__, __VA_ARGS__);                \
      abort();                                                                        \
    }                                                                                 \
  } while (0)

#define WM_API_BEGIN try {
#define WM_API_END                                               \
  }                                                              \
  catch (const ::wm::invalid_input& e)                           \
  {                                                              \
    WM_ERROR("invalid input: %s", e.what());                     \
    return WHOLEMEMORY_INVALID_INPUT;                            \
  }                                                              \
  catch (const ::wm::hip_error& e)                               \
  {                                                              \
    WM_ERROR("HIP error: %s", e.what());                         \
    return WHOLEMEMORY_CUDA_ERROR;                               \
  }                                                              \
  catch (const ::wm::comm_error& e)                              \
  {                                                              \
    WM_ERROR("communication error: %s", e.what());               \
    return WHOLEMEMORY_COMMUNICATION_ERROR;                      \
  }                                                              \
  catch (const ::wm::logic_error& e)                             \
  {                                                              \
    WM_ERROR("logic error: %s", e.what());                       \
    return WHOLEMEMORY_LOGIC_ERROR;                              \
  }                                                              \
  catch (const std::bad_alloc&)                                  \
  {                                                              \
    WM_ERROR("out of memory");                                   \
    return WHOLEMEMORY_OUT_OF_MEMORY;                            \
  }                                                              \
  catch (const std::exception& e)                                \
  {                                                              \
    WM_ERROR("unknown error: %s", e.what());                     \
    return WHOLEMEMORY_UNKNOW_ERROR;                             \
  }                                                              \
  catch (...)                                                    \
  {                                                              \
    return WHOLEMEMORY_UNKNOW_ERROR;                             \
  }
