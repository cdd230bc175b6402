// Thread-local error message of the C ABI (edet_last_error) and the debug launch log.
#include <cxxabi.h>
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>

#include "../../include/edet_hip.h"

static thread_local char g_err[512] = "";

void edet_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* edet_last_error(void) { return g_err; }
extern "C" int edet_version(void) { return 2; }

// ---- debug launch log: host function pointer of every kernel launched while the log is on -> count
int g_edet_launch_log_on = 0;
static std::mutex g_log_mu;
static std::map<const void*, long> g_log_counts;

void edet_log_launch(const void* host_fn) {
  std::lock_guard<std::mutex> lock(g_log_mu);
  ++g_log_counts[host_fn];
}

extern "C" int edet_debug_launch_log(int enable) {
  std::lock_guard<std::mutex> lock(g_log_mu);
  if (enable) g_log_counts.clear();
  g_edet_launch_log_on = enable ? 1 : 0;
  return 0;
}

extern "C" int edet_debug_launch_names(char* buf, size_t capacity, size_t* needed) {
  std::string text;
  {
    std::lock_guard<std::mutex> lock(g_log_mu);
    std::map<std::string, long> by_name;
    for (const auto& kv : g_log_counts) {
      const char* mangled = hipKernelNameRefByPtr(kv.first, nullptr);
      std::string name;
      if (mangled) {
        int status = 0;
        char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
        name = (status == 0 && dem) ? dem : mangled;
        free(dem);
      } else {
        char tmp[32];
        snprintf(tmp, sizeof(tmp), "kernel@%p", kv.first);
        name = tmp;
      }
      by_name[name] += kv.second;
    }
    for (const auto& kv : by_name) text += std::to_string(kv.second) + "\t" + kv.first + "\n";
  }
  if (needed) *needed = text.size() + 1;
  if (buf && capacity) {
    const size_t n = text.size() < capacity - 1 ? text.size() : capacity - 1;
    memcpy(buf, text.data(), n);
    buf[n] = 0;
  }
  return 0;
}

// ---- resident workgroups of a kernel on the current device (occupancy x compute units), cached -------------------------
// Launch planning uses it for the kernels whose workgroups each loop over a share of the work: a grid slightly larger than
// what the chip holds at once runs a nearly empty last round (r03e: 513 workgroups on 512 slots cost 28 %).
struct OccKey {
  const void* fn;
  int threads;
  size_t lds;
  bool operator<(const OccKey& o) const {
    if (fn != o.fn) return fn < o.fn;
    if (threads != o.threads) return threads < o.threads;
    return lds < o.lds;
  }
};
static std::mutex g_occ_mu;
static std::map<OccKey, int> g_occ;

int edet_resident_wgs(const void* fn, int threads, size_t lds) {
  std::lock_guard<std::mutex> lock(g_occ_mu);
  const OccKey key{fn, threads, lds};
  auto it = g_occ.find(key);
  if (it != g_occ.end()) return it->second;
  int per_cu = 0, dev = 0, cus = 0;
  int total = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) == hipSuccess && per_cu > 0 &&
      hipGetDevice(&dev) == hipSuccess &&
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
    total = per_cu * cus;
  (void)hipGetLastError();
  g_occ[key] = total;      // 0 = unknown (callers keep their fixed targets)
  return total;
}
