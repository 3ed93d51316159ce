// Error reporting and device queries of libsrlhip.so.
#include <stdarg.h>

#include <atomic>
#include <mutex>

#include "srl_common.h"

static thread_local char g_err[512] = "";

void srl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static std::atomic<long long> g_dispatch[SRL_DISP_FAMILIES];

void srl_count_dispatch(int family) {
  if (family >= 0 && family < SRL_DISP_FAMILIES) g_dispatch[family].fetch_add(1, std::memory_order_relaxed);
}

extern "C" int srl_dispatch_counts(int64_t* out, int n, int reset) {
  SRL_CHECK_ARG(out != nullptr || n == 0, "null output");
  for (int i = 0; i < n; ++i) out[i] = i < SRL_DISP_FAMILIES ? (int64_t)g_dispatch[i].load(std::memory_order_relaxed) : 0;
  if (reset)
    for (int i = 0; i < SRL_DISP_FAMILIES; ++i) g_dispatch[i].store(0, std::memory_order_relaxed);
  return 0;
}

// instantiation counters: a short table under a mutex (host-side, a few launches per microsecond at most)
static std::mutex g_tile_mu;
static uint64_t g_tile_key[256];
static long long g_tile_cnt[256];
static int g_tile_n = 0;

void srl_count_tile(int family, int p0, int p1, int p2, int flags) {
  const uint64_t key = ((uint64_t)(family & 0xff) << 56) | ((uint64_t)(p0 & 0xffff) << 40) | ((uint64_t)(p1 & 0xffff) << 24) |
                       ((uint64_t)(p2 & 0xffff) << 8) | (uint64_t)(flags & 0xff);
  std::lock_guard<std::mutex> lk(g_tile_mu);
  for (int i = 0; i < g_tile_n; ++i)
    if (g_tile_key[i] == key) { ++g_tile_cnt[i]; return; }
  if (g_tile_n < 256) { g_tile_key[g_tile_n] = key; g_tile_cnt[g_tile_n++] = 1; }
}

extern "C" int srl_dispatch_tiles(uint64_t* keys, int64_t* counts, int cap, int reset) {
  SRL_CHECK_ARG((keys && counts) || cap == 0, "null output");
  std::lock_guard<std::mutex> lk(g_tile_mu);
  const int n = g_tile_n;
  for (int i = 0; i < n && i < cap; ++i) { keys[i] = g_tile_key[i]; counts[i] = g_tile_cnt[i]; }
  if (reset) g_tile_n = 0;
  return n;
}

extern "C" int srl_abi_version(void) { return SRL_HIP_ABI_VERSION; }

extern "C" const char* srl_last_error(void) { return g_err; }

extern "C" int srl_device_info(int* num_cus, int* lds_bytes_per_cu, char* name, int name_len) {
  int dev = 0;
  SRL_HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  SRL_HIP_TRY(hipGetDeviceProperties(&prop, dev));
  if (num_cus) *num_cus = prop.multiProcessorCount;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (name && name_len > 0) {
    strncpy(name, prop.gcnArchName, (size_t)name_len - 1);
    name[name_len - 1] = 0;
  }
  return 0;
}
