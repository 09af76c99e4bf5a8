// Development switches of libsionna_amd.so.
//
// The product path never reads the process environment.  The SAMD_* variables are copied ONCE into this registry when
// the library is loaded (api.cpp, a single pass over `environ`); afterwards a value changes only through the C entry
// samd_debug_set_option().  Code that BUILDS a handle (tables, schedules) reads the registry while it builds; code on a
// LAUNCH path reads either the snapshot its handle took at creation (samd_ldpc5g::opt) or, where there is no handle,
// a CachedOpt (one relaxed atomic load per call).  A handle therefore behaves the same for its whole life, two
// handles created under different options can decode concurrently from two host threads, and `getenv` cannot race
// with a `setenv` elsewhere in the process ("re-entrant per handle + stream").
#pragma once
#include <atomic>
#include <string>

namespace samd {

bool opt_set(const char* key);                  // present (with any value)
long opt_int(const char* key, long dflt);       // atol(value) or dflt
std::string opt_str(const char* key);           // "" when absent
int opt_generation();                           // bumped by every samd_debug_set_option

// For launch paths without a handle: re-reads the registry only when its generation moved.
struct CachedOpt {
  const char* key;
  std::atomic<int> gen{-1};
  std::atomic<int> present{0};
  std::atomic<long> value{0};
  explicit CachedOpt(const char* k) : key(k) {}
  void refresh() {
    const int g = opt_generation();
    if (gen.load(std::memory_order_acquire) == g) return;
    present.store(opt_set(key) ? 1 : 0, std::memory_order_relaxed);
    value.store(opt_int(key, 0), std::memory_order_relaxed);
    gen.store(g, std::memory_order_release);
  }
  bool is_set() { refresh(); return present.load(std::memory_order_relaxed) != 0; }
  long get(long dflt) { refresh(); return present.load(std::memory_order_relaxed) ? value.load(std::memory_order_relaxed) : dflt; }
};

}  // namespace samd
