// Library-level entry points of include/mr3d.h (version, status strings, device probe).
#include "mr3d_common.h"
#include <string.h>

extern "C" const char *mr3d_version(void) { return "mr3d-hip 0.1 (gfx950)"; }

extern "C" const char *mr3d_status_string(int status) {
  switch (status) {
    case MR3D_OK: return "ok";
    case MR3D_ERR_INVALID_ARG: return "invalid argument";
    case MR3D_ERR_WORKSPACE: return "workspace missing, misaligned or too small";
    case MR3D_ERR_LAUNCH: return "kernel launch failed";
    case MR3D_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown status";
  }
}

extern "C" int mr3d_device_ok(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

// measurement knobs (include/mr3d.h): plain ints read by the dispatch code, 0 = shipped default
int g_mr3d_tuning[MR3D_TUNE_NUM_KEYS] = {};
extern "C" void mr3d_tuning_set(int key, int value) { if (key >= 0 && key < MR3D_TUNE_NUM_KEYS) g_mr3d_tuning[key] = value; }
extern "C" int mr3d_tuning_get(int key) { return key >= 0 && key < MR3D_TUNE_NUM_KEYS ? g_mr3d_tuning[key] : 0; }
