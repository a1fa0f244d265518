// wholegraph_amd — environment knobs, read ONCE.
//
// Every WM_* / WG_* switch of the library goes through WM_KNOB("NAME"): the environment is consulted the first time the
// call site runs and the answer is kept for the life of the process — no getenv on the path of an op (a dozen of them per
// row-kernel launch were µs-scale host work on every call of a latency-bound chain, and a knob could change behaviour
// mid-process). Experiments and tests that flip a switch between calls say so explicitly:
// wholememory_ext_reload_knobs() (include/wholememory/wholegraph_amd_ext.h) makes every site read its variable again at
// its next use. A reload never frees a value: every (re)load parks its string in a list that lives as long as the process, so a
// pointer returned to an op on another thread stays valid (advisor, round 4) — that thread simply keeps the old answer until
// its next look. (A few bytes per reload and variable; reloads are a test / experiment device.)
#pragma once

#include <atomic>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <string>

namespace wm {

extern std::atomic<unsigned> g_knob_generation;   // wm_common.cpp
extern std::mutex g_knob_mutex;                    // serialises the (re)loads; the cached read takes no lock

class env_knob {
 public:
  explicit env_knob(const char* name) : name_(name), gen_(~0u) {}
  // value of the variable as of the last (re)load, nullptr when unset
  const char* str() const
  {
    const unsigned g = g_knob_generation.load(std::memory_order_acquire);
    if (gen_.load(std::memory_order_acquire) != g) {   // first use, or a reload was asked for: threads may race to get here
      std::lock_guard<std::mutex> lk(g_knob_mutex);
      if (gen_.load(std::memory_order_relaxed) != g) {
        const char* e = std::getenv(name_);
        if (e != nullptr) {
          history_.emplace_back(e);   // (deque: growing it moves no element)
          value_.store(history_.back().c_str(), std::memory_order_release);
        } else {
          value_.store(nullptr, std::memory_order_release);
        }
        gen_.store(g, std::memory_order_release);
      }
    }
    return value_.load(std::memory_order_acquire);
  }

 private:
  const char* name_;
  mutable std::atomic<unsigned> gen_;
  mutable std::atomic<const char*> value_{nullptr};
  mutable std::deque<std::string> history_;   // every value ever loaded (guarded by g_knob_mutex)
};

inline void reload_knobs() { g_knob_generation.fetch_add(1, std::memory_order_acq_rel); }

}  // namespace wm

// one static per call site (the lambda gives each expansion its own)
#define WM_KNOB(NAME)                      \
  ([]() -> const char* {                   \
    static const ::wm::env_knob k(NAME);   \
    return k.str();                        \
  }())

// A/B switches of kernel variants (WM_ROWS_*, WM_TILE_*, WM_STEP_TILE*, WM_SAMPLE_*, ... — what rounds 2-5 used to compare code
// paths inside one process) are NOT part of the product (round-5 review: ~45 variables, two thirds of them branches in hot
// launchers): in the shipped build such a site is the constant nullptr and its branch folds away. A variant build
// (scripts/build_variant.sh NAME "..." with AB=1, or make AB=1) compiles them back in for experiments/*.
#ifdef WM_AB_KNOBS
#define WM_AB_KNOB(NAME) WM_KNOB(NAME)
#else
#define WM_AB_KNOB(NAME) (static_cast<const char*>(nullptr))
#endif
