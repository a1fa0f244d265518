// wholegraph_amd — library lifetime, device properties, backend selection.
// Reference: cpp/src/wholememory/initialize.cpp:38-77, system_info.cpp, env_func_ptrs.cpp (get_device_prop).
#include <hip/hip_runtime_api.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include <wholememory/env_func_ptrs.h>
#include <wholememory/wholegraph_amd_ext.h>
#include <wholememory/wholememory.h>

#include "knobs.hpp"
#include "backend.hpp"
#include "wm_common.hpp"

namespace wm {
namespace {
const wm_device_backend* g_backend = nullptr;
std::mutex g_mu;
bool g_inited = false;
std::vector<hipDeviceProp_t> g_props;
}  // namespace

const wm_device_backend* backend()
{
  const wm_device_backend* b = g_backend;
  return b != nullptr ? b : hip_backend();
}
}  // namespace wm

extern "C" {

wholememory_error_code_t wholememory_init(unsigned int flags, LogLevel log_level)
{
  WM_API_BEGIN
  std::lock_guard<std::mutex> g(wm::g_mu);
  if (flags != 0) return WHOLEMEMORY_INVALID_INPUT;  // reserved, reference initialize.cpp:41-44
  wm::log_level_ref() = log_level;
  if (wm::g_inited) return WHOLEMEMORY_SUCCESS;
  const auto* bk = wm::backend();
  int n          = bk->device_count();
  if (bk == wm::hip_backend()) {
    // The product has no CPU path: without a visible GPU this library cannot do anything.
    if (n <= 0) {
      WM_ERROR("wholememory_init: no HIP device visible (hipGetDeviceCount == 0); this library has no CPU fallback");
      return WHOLEMEMORY_CUDA_ERROR;
    }
    wm::g_props.resize(n);
    for (int i = 0; i < n; i++) {
      if (hipGetDeviceProperties(&wm::g_props[i], i) != hipSuccess) return WHOLEMEMORY_CUDA_ERROR;
    }
  }
  wm::g_inited = true;
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

wholememory_error_code_t wholememory_finalize()
{
  std::lock_guard<std::mutex> g(wm::g_mu);
  wm::g_inited = false;
  wm::g_props.clear();
  wholememory_drop_cached_env_func_cache();
  return WHOLEMEMORY_SUCCESS;
}

void* get_device_prop(int dev_id)
{
  if (wm::backend() != wm::hip_backend()) return nullptr;
  if (dev_id < 0 && hipGetDevice(&dev_id) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> g(wm::g_mu);
  if (dev_id >= static_cast<int>(wm::g_props.size())) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || dev_id >= n) return nullptr;
    wm::g_props.resize(n);
    for (int i = 0; i < n; i++) (void)hipGetDeviceProperties(&wm::g_props[i], i);
  }
  return &wm::g_props[dev_id];
}

// reference parallel_utils.cpp:46-63 (ForkGetDeviceCount): count devices in a child so the parent
// holds no HIP context before it forks workers
int fork_get_device_count()
{
  int fds[2];
  if (pipe(fds) != 0) return -1;
  pid_t pid = fork();
  if (pid < 0) return -1;
  if (pid == 0) {
    close(fds[0]);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = -1;
    ssize_t w = write(fds[1], &n, sizeof(n));
    (void)w;
    close(fds[1]);
    _exit(0);
  }
  close(fds[1]);
  int n     = -1;
  ssize_t r = read(fds[0], &n, sizeof(n));
  close(fds[0]);
  int status = 0;
  waitpid(pid, &status, 0);
  return r == static_cast<ssize_t>(sizeof(n)) ? n : -1;
}

const char* wholememory_ext_backend_name() { return wm::backend()->name; }

wholememory_error_code_t wm_testing_install_backend(const void* backend)
{
  const char* e = WM_KNOB("WHOLEGRAPH_AMD_TESTING");
  if (e == nullptr || strcmp(e, "1") != 0) {
    WM_ERROR("wm_testing_install_backend refused: WHOLEGRAPH_AMD_TESTING=1 is not set (test-only seam)");
    return WHOLEMEMORY_NOT_SUPPORTED;
  }
  std::lock_guard<std::mutex> g(wm::g_mu);
  wm::g_backend = static_cast<const wm_device_backend*>(backend);
  wm::g_inited  = false;
  WM_WARN("device backend replaced by '%s' — TEST MODE, not a product configuration", wm::backend()->name);
  return WHOLEMEMORY_SUCCESS;
}

}  // extern "C"
