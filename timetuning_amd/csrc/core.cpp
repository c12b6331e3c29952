// Error reporting and device query for libtimetuning_hip.so.
#include "common.hpp"
#include <atomic>

namespace tt {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// CU count of the calling thread's current device, looked up once per device id (a count cached for whichever device was
// current at the first call is wrong for the others of a multi-GPU process)
int device_cu_count() {
  constexpr int kMaxDev = 64;
  static std::atomic<int> cache[kMaxDev];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 256;
  int n = cache[dev].load(std::memory_order_relaxed);
  if (n > 0) return n;
  hipDeviceProp_t p;
  n = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  cache[dev].store(n, std::memory_order_relaxed);
  return n;
}
}  // namespace tt

extern "C" const char* tt_last_error(void) { return tt::g_err; }
extern "C" int tt_abi_version(void) { return 5; }   // 5: the fp16-pair entry points (tt_*_pairs*), tt_vit_params.planes == 2; 4: tt_vit_params.patch_wp; 3: the coarse entry points (tt_vit_forward, ...) and their parameter structs

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
