#include "wm_common.hpp"

#include <cstring>
#include <ctime>
#include <unistd.h>

#include <wholememory/wholegraph_amd_ext.h>

namespace wm {

LogLevel& log_level_ref()
{
  static LogLevel lvl = LEVEL_INFO;
  return lvl;
}

static const char* level_name(LogLevel l)
{
  static const char* names[] = {"FATAL", "ERROR", "WARN", "INFO", "DEBUG", "TRACE"};
  return (l >= LEVEL_FATAL && l <= LEVEL_TRACE) ? names[l] : "?";
}

void log_message(LogLevel lvl, const char* file, int line, const char* fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  const char* base = strrchr(file, '/');
  fprintf(stderr, "[wholegraph_amd %s pid=%d] %s (%s:%d)\n", level_name(lvl), (int)getpid(), buf,
          base ? base + 1 : file, line);
}

std::string format_string(const char* fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return std::string(buf);
}

bool debug_sync_enabled()
{
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("WM_DEBUG_SYNC");
    v             = (e != nullptr && e[0] != '\0' && e[0] != '0') ? 1 : 0;
  }
  return v == 1;
}

namespace {
int g_async_completion = 0;   // what the host framework declared (wholememory_ext_set_async_completion); off = reference semantics
}
void set_async_completion(bool on) { g_async_completion = on ? 1 : 0; }
bool async_completion_enabled()
{
  static int forced = -2;
  if (forced == -2) {
    const char* e = getenv("WM_ASYNC_OPS");
    forced        = (e == nullptr || e[0] == '\0') ? -1 : (e[0] != '0' ? 1 : 0);
  }
  return forced >= 0 ? forced == 1 : g_async_completion == 1;
}

// (read at every call: two getenv per HOST gather are noise beside a PCIe-bound kernel, and tests switch them per case)
int64_t host_sorted_gather_min()
{
  const char* off = getenv("WM_HOST_SORTED_GATHER");
  if (off != nullptr && off[0] == '0') return 0;
  const char* e = getenv("WM_HOST_SORTED_MIN");
  return e != nullptr && atoll(e) > 0 ? static_cast<int64_t>(atoll(e)) : static_cast<int64_t>(1) << 19;
}

int host_sorted_gather_low_bit()
{
  const char* e = getenv("WM_HOST_SORTED_LOW_BIT");
  return e != nullptr && atoi(e) > 0 ? atoi(e) : 0;
}

}  // namespace wm

extern "C" wholememory_error_code_t wholememory_ext_set_async_completion(int on)
{
  wm::set_async_completion(on != 0);
  return WHOLEMEMORY_SUCCESS;
}
