// Error reporting and device query for libtimetuning_hip.so.
#include "common.hpp"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

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

// ---- the K-split workspace of the persistent GEMMs (common.hpp): [counters: CUs x 8 ints, padded to 256 bytes][partials: CUs x 8 waves x
// 128 x 64 floats - the larger of the two kernels' wave tiles (gemm_planes8: 128 x 64, gemm_pairs8: 64 x 64)]
size_t ksplit_ws_counter_bytes() { return (((size_t)device_cu_count() * 8 * sizeof(int)) + 255) / 256 * 256; }
size_t ksplit_ws_bytes() { return ksplit_ws_counter_bytes() + (size_t)device_cu_count() * 8 * 128 * 64 * sizeof(float); }
bool ksplit_ws_carve(void* ws, size_t bytes, KsplitWs* out) {
  if (!ws || bytes < ksplit_ws_bytes() || !aligned16(ws)) return false;
  out->counters = static_cast<int*>(ws);
  out->partials = reinterpret_cast<float*>(static_cast<unsigned char*>(ws) + ksplit_ws_counter_bytes());
  return true;
}

// ---- tuning knobs: read ONCE from the environment into atomics - a launch path never calls getenv (not safe against a
// concurrent setenv from another Python thread, and a knob must not flip under a production run) - with an explicit setter for the A/B
// tools and tests that compare settings inside one process.
static const char* const kKnobNames[KNOB_COUNT] = {"TT_PLANES_VARIANT", "TT_P8_ORDER", "TT_P8_NO_HALF", "TT_P8_CLOCK_PRINT",
                                                   "TT_Q8_ORDER",       "TT_PAIRS_NO8", "TT_PAIRS8_NO_KEPT", "TT_Q8_KSPLIT", "TT_ATTN_PAIRS_FLASH", "TT_TN_WGS", "TT_TN_XCD", "TT_Q8_STREAM", "TT_Q8_MIN_TILES", "TT_SK_PERSIST", "TT_ATTN_PAIRS_PERSIST", "TT_PAIRS_NBUF", "TT_Q4", "TT_Q4_SMALL", "TT_SPLIT_ROWS"};
static const int kKnobDefaults[KNOB_COUNT] = {0, 3, 0, 0, 3, 0, 0, 1, 0, 0, 1, 1, 128, 0, 1, 0, 0, 0, 1};
static std::atomic<int> g_knobs[KNOB_COUNT];
static std::once_flag g_knobs_once;
static void knobs_init() {
  for (int i = 0; i < KNOB_COUNT; ++i) {
    const char* e = getenv(kKnobNames[i]);
    g_knobs[i].store(e ? (*e ? atoi(e) : 1) : kKnobDefaults[i], std::memory_order_relaxed);
  }
}
int tuning_knob(int which) {
  std::call_once(g_knobs_once, knobs_init);
  return g_knobs[which].load(std::memory_order_relaxed);
}
}  // namespace tt

extern "C" int tt_set_tuning_knob(const char* name, int value) {
  std::call_once(tt::g_knobs_once, tt::knobs_init);
  for (int i = 0; i < tt::KNOB_COUNT; ++i)
    if (name && strcmp(name, tt::kKnobNames[i]) == 0) {
      tt::g_knobs[i].store(value, std::memory_order_relaxed);
      return TT_OK;
    }
  tt::set_error("set_tuning_knob: unknown knob '%s'", name ? name : "(null)");
  return TT_EINVAL;
}

extern "C" const char* tt_last_error(void) { return tt::g_err; }
extern "C" int tt_abi_version(void) { return 8; }   // 8: `precision` is an argument (tt_linear_fwd, tt_label_propagate[_maps], tt_mlp_head_forward, tt_scores_sinkhorn, tt_vit_params.precision) - tt_set_gemm_precision is gone; 7: caller-owned K-split workspace (tt_linear_ksplit_workspace_*; workspace arguments of tt_linear_fwd_pairs / _planes / tt_linear_bwd_data_pairs / _planes), the pair producers' range flag; 6: tt_linear_bwd_weight_pairs_tn*, tt_split_pairs_dual_multi, tt_sinkhorn_local_*, pair attention at any N; 5: the fp16-pair entry points (tt_*_pairs*), tt_vit_params.planes == 2; 4: tt_vit_params.patch_wp; 3: the coarse entry points (tt_vit_forward, ...) and their parameter structs

extern "C" size_t tt_amax_slot_bytes(void) { return (size_t)tt::kAmaxWays * tt::kAmaxStride * sizeof(float); }
extern "C" size_t tt_linear_ksplit_workspace_bytes(void) { return tt::ksplit_ws_bytes(); }
extern "C" int tt_linear_ksplit_workspace_init(void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  tt::KsplitWs k;
  TT_REQUIRE(tt::ksplit_ws_carve(workspace, workspace_bytes, &k), "linear_ksplit_workspace_init: null, misaligned or smaller than tt_linear_ksplit_workspace_bytes()");
  if (hipMemsetAsync(k.counters, 0, tt::ksplit_ws_counter_bytes(), tt::as_stream(stream)) != hipSuccess) {
    tt::set_error("linear_ksplit_workspace_init: hipMemsetAsync failed");
    return TT_ELAUNCH;
  }
  return TT_OK;
}

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
