// Library-wide pieces of the C-ABI: error string, version, device count, the development-option registry.
#include "common.h"
#include "options.h"

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

extern char** environ;

namespace samd {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

namespace {
struct Registry {
  std::mutex mu;
  std::map<std::string, std::string> kv;
  std::atomic<int> generation{0};
  Registry() {                                   // the ONE read of the process environment (library load)
    for (char** e = environ; e && *e; ++e) {
      // (where the code-object cache of the generated kernels lives, ldpc5g_jit.cpp: kept under internal keys)
      if (strncmp(*e, "XDG_CACHE_HOME=", 15) == 0) kv.emplace("SAMD__XDG_CACHE_HOME", std::string(*e + 15));
      if (strncmp(*e, "HOME=", 5) == 0) kv.emplace("SAMD__HOME", std::string(*e + 5));
      if (strncmp(*e, "SAMD_", 5) != 0) continue;
      const char* eq = strchr(*e, '=');
      if (eq) kv.emplace(std::string(*e, eq - *e), std::string(eq + 1));
    }
  }
};
Registry& registry() {
  static Registry r;
  return r;
}
}  // namespace

bool opt_set(const char* key) {
  Registry& r = registry();
  std::lock_guard<std::mutex> lk(r.mu);
  return r.kv.count(key) != 0;
}
long opt_int(const char* key, long dflt) {
  Registry& r = registry();
  std::lock_guard<std::mutex> lk(r.mu);
  auto it = r.kv.find(key);
  return it == r.kv.end() ? dflt : atol(it->second.c_str());
}
std::string opt_str(const char* key) {
  Registry& r = registry();
  std::lock_guard<std::mutex> lk(r.mu);
  auto it = r.kv.find(key);
  return it == r.kv.end() ? std::string() : it->second;
}
bool host_only() { return opt_set("SAMD_HOST_ONLY"); }
int opt_generation() { return registry().generation.load(std::memory_order_acquire); }
}  // namespace samd

extern "C" const char* samd_last_error(void) { return samd::g_last_error.c_str(); }
extern "C" int samd_version(void) { return 102; }
extern "C" int samd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return SAMD_ERR_HIP;
  return n;
}

extern "C" int samd_debug_set_option(const char* key, const char* value) {
  if (!key || strncmp(key, "SAMD_", 5) != 0 || strncmp(key, "SAMD__", 6) == 0) {
    samd::set_error("samd_debug_set_option: key must start with SAMD_");
    return SAMD_ERR_INVALID;
  }
  samd::Registry& r = samd::registry();
  {
    std::lock_guard<std::mutex> lk(r.mu);
    if (value) r.kv[key] = value;
    else r.kv.erase(key);
  }
  r.generation.fetch_add(1, std::memory_order_acq_rel);
  return SAMD_OK;
}
extern "C" int samd_debug_options_generation(void) { return samd::opt_generation(); }
// value of a development switch: its length (copied, NUL-terminated, when cap suffices) or -1 when it is not set
extern "C" long samd_debug_get_option(const char* key, char* buf, size_t cap) {
  if (!key || strncmp(key, "SAMD_", 5) != 0 || strncmp(key, "SAMD__", 6) == 0) return -1;
  if (!samd::opt_set(key)) return -1;
  const std::string v = samd::opt_str(key);
  if (buf && cap > v.size()) memcpy(buf, v.c_str(), v.size() + 1);
  return (long)v.size();
}
