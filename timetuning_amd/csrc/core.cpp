// Error reporting and device query for libtimetuning_hip.so.
#include "common.hpp"

namespace tt {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace tt

extern "C" const char* tt_last_error(void) { return tt::g_err; }
extern "C" int tt_abi_version(void) { return 4; }   // 4: tt_vit_params.patch_wp; 3: the coarse entry points (tt_vit_forward, ...) and their parameter structs

extern "C" int tt_device_info(char* name, int cap) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    tt::set_error("device_info: no HIP device");
    return TT_ELAUNCH;
  }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
    tt::set_error("device_info: hipGetDeviceProperties failed");
    return TT_ELAUNCH;
  }
  if (name && cap > 0) snprintf(name, (size_t)cap, "%s", p.gcnArchName);
  return p.multiProcessorCount;
}
