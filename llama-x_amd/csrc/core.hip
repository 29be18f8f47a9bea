// Library-wide plumbing: version, thread-local error string, device query.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void llx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int llx_version(void) { return 105; }  // 0.1.5 (round 3)

extern "C" const char* llx_last_error_string(void) { return g_err; }

// Fills name (<=len bytes) with the gcn arch string of `device`; returns CU count or <0.
extern "C" int llx_device_info(int device, char* name, int len) {
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, device);
  if (e != hipSuccess) {
    llx_set_error("llx_device_info: %s", hipGetErrorString(e));
    return LLX_ERR_LAUNCH;
  }
  if (name && len > 0) {
    strncpy(name, p.gcnArchName, (size_t)len - 1);
    name[len - 1] = 0;
  }
  return p.multiProcessorCount;
}
