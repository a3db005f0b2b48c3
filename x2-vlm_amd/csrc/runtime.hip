// Error reporting shared by every C-ABI entry point (include/x2vlm_hip.h): entry points return 0 or a
// negative code and leave a message retrievable with x2_last_error().  Kernels never allocate; the
// caller (PyTorch's caching allocator on the Python side) owns all memory; every launch is
// asynchronous on the stream handed in.
#include "x2_common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void x2_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int x2_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return X2_OK;
  x2_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
  return X2_ERR_LAUNCH;
}

extern "C" const char* x2_last_error(void) { return g_err; }
extern "C" int x2_abi_version(void) { return 14; }

// device-side sanity: number of compute units of the current device (0 when no HIP device is usable)
extern "C" int x2_device_cus(void) {
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
  return p.multiProcessorCount;
}
