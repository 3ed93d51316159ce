// Error reporting and device queries of libsrlhip.so.
#include <stdarg.h>

#include "srl_common.h"

static thread_local char g_err[512] = "";

void srl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
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
