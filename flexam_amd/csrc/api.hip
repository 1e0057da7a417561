// flexam_amd/csrc/api.hip -- version / error plumbing of libflexam_hip.so.
// Convention (include/flexam_hip.h): every entry point returns 0 or a negative FLEXAM_E_* code
// and leaves a human-readable message for flexam_last_error(); nothing allocates, frees,
// synchronises or keeps a pointer past the call; all work is ordered on the caller's stream.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"
#include "flexam_hip.h"

static thread_local char g_err[512] = "";

int flexam_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int flexam_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return flexam_fail(FLEXAM_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return FLEXAM_OK;
}

int flexam_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  return dev < FLEXAM_MAX_DEVICES ? dev : FLEXAM_MAX_DEVICES - 1;
}

static int g_cu_budget[FLEXAM_MAX_DEVICES] = {};         // flexam_set_cu_budget: 0 = all of the device's CUs

static int hardware_cus() {
  static int cus[FLEXAM_MAX_DEVICES] = {};               // filled on first use per device (benign race: same value)
  const int dev = flexam_current_device();
  if (cus[dev] == 0) {
    int n = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8) n = prop.multiProcessorCount / 8 * 8;
    cus[dev] = n;
  }
  return cus[dev];
}

int flexam_num_cus() {
  const int n = hardware_cus(), b = g_cu_budget[flexam_current_device()];
  return b > 0 && b < n ? b : n;
}

// The CUs the library plans its persistent grids for (one workgroup per CU, all of its LDS and registers).  A kernel that owns EVERY CU
// leaves nothing for a collective's kernels to run on beside it: a workgroup of RCCL (or the emulation's delay wave) then waits for the
// launch to end, or -- when it got there first -- one of our 256 workgroups waits for IT.  Under an exchange that is meant to travel
// beside compute a few CUs are therefore left out of the plan (8 = one per XCD: -3 % of the matrix rate).
extern "C" int flexam_set_cu_budget(int cus) {
  const int n = hardware_cus();
  FX_REQUIRE(cus == 0 || (cus >= 8 && cus <= n && cus % 8 == 0), FLEXAM_E_ARG, "set_cu_budget: %d (0 = all, or a multiple of 8 in 8 .. %d)", cus, n);
  g_cu_budget[flexam_current_device()] = cus;
  return FLEXAM_OK;
}

extern "C" int flexam_device_cus(void) { return flexam_num_cus(); }

extern "C" const char* flexam_last_error(void) { return g_err; }
extern "C" int flexam_version(void) { return FLEXAM_HIP_VERSION; }
extern "C" const char* flexam_arch(void) { return "gfx950"; }

extern "C" int flexam_device_check(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return flexam_fail(FLEXAM_E_ARCH, "no HIP device");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return flexam_fail(FLEXAM_E_ARCH, "cannot query device %d", dev);
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return flexam_fail(FLEXAM_E_ARCH, "device %d is %s; libflexam_hip.so is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
  return FLEXAM_OK;
}
