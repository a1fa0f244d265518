#include "wm_common.hpp"

#include <cstring>
#include <ctime>
#include <unistd.h>

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

}  // namespace wm
