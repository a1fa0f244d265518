#include "knobs.hpp"
#include "wm_common.hpp"

#include <atomic>
#include <cstring>
#include <ctime>
#include <mutex>
#include <unordered_map>
#include <unistd.h>

#include "backend.hpp"

#include <wholememory/wholegraph_amd_ext.h>

namespace wm {

std::atomic<unsigned> g_knob_generation{0};
std::mutex g_knob_mutex;

namespace {
std::mutex g_gref_mu;
std::unordered_map<const void*, gref_host_tables> g_gref_tables;
}  // namespace

void register_gref_tables(const void* dev_rank_ptrs, int world_size, void* const* rank_ptrs, const size_t* rank_offsets)
{
  if (dev_rank_ptrs == nullptr || world_size < 1 || world_size > kOwnersByValue) return;
  gref_host_tables t{};
  t.world_size = world_size;
  for (int r = 0; r < world_size; r++) t.rank_ptrs[r] = rank_ptrs[r];
  for (int r = 0; r <= world_size; r++) t.rank_offsets[r] = rank_offsets[r];
  std::lock_guard<std::mutex> lk(g_gref_mu);
  g_gref_tables[dev_rank_ptrs] = t;
}
void unregister_gref_tables(const void* dev_rank_ptrs)
{
  std::lock_guard<std::mutex> lk(g_gref_mu);
  g_gref_tables.erase(dev_rank_ptrs);
}
bool lookup_gref_tables(const void* dev_rank_ptrs, gref_host_tables* out)
{
  std::lock_guard<std::mutex> lk(g_gref_mu);
  auto it = g_gref_tables.find(dev_rank_ptrs);
  if (it == g_gref_tables.end()) return false;
  *out = it->second;
  return true;
}

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
  const char* e = WM_KNOB("WM_DEBUG_SYNC");
  return e != nullptr && e[0] != '\0' && e[0] != '0';
}

namespace {
std::atomic<int> g_async_completion{0};   // what the host framework declared (wholememory_ext_set_async_completion); off = reference semantics
}
void set_async_completion(bool on) { g_async_completion.store(on ? 1 : 0, std::memory_order_relaxed); }
bool async_completion_enabled()
{
  const char* e    = WM_KNOB("WM_ASYNC_OPS");
  const int forced = (e == nullptr || e[0] == '\0') ? -1 : (e[0] != '0' ? 1 : 0);
  return forced >= 0 ? forced == 1 : g_async_completion.load(std::memory_order_relaxed) == 1;
}

int64_t host_sorted_gather_min()
{
  const char* off = WM_KNOB("WM_HOST_SORTED_GATHER");
  if (off != nullptr && off[0] == '0') return 0;
  const char* e = WM_KNOB("WM_HOST_SORTED_MIN");
  return e != nullptr && atoll(e) > 0 ? static_cast<int64_t>(atoll(e)) : static_cast<int64_t>(1) << 19;
}

int host_sorted_gather_low_bit()
{
  const char* e = WM_AB_KNOB("WM_HOST_SORTED_LOW_BIT");
  return e != nullptr && atoi(e) > 0 ? atoi(e) : 0;
}

}  // namespace wm

extern "C" wholememory_error_code_t wholememory_ext_reload_knobs()
{
  wm::reload_knobs();
  return WHOLEMEMORY_SUCCESS;
}

extern "C" wholememory_error_code_t wholememory_ext_set_async_completion(int on)
{
  wm::set_async_completion(on != 0);
  return WHOLEMEMORY_SUCCESS;
}
