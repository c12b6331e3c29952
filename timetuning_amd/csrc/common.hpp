// Shared host/device helpers for libtimetuning_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>

#include "../../include/timetuning_hip.h"

namespace tt {

constexpr int kWave = 64;  // CDNA wavefront

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(tt_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define TT_REQUIRE(cond, ...)         \
  do {                                \
    if (!(cond)) {                    \
      ::tt::set_error(__VA_ARGS__);   \
      return TT_EINVAL;               \
    }                                 \
  } while (0)

#define TT_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      ::tt::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return TT_ELAUNCH;                                                   \
    }                                                                      \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int device_cu_count();   // core.cpp: multiprocessors of the CURRENT device (looked up once per device id)
// core.cpp: tuning knobs, read once from the environment (TT_<NAME>) and settable through tt_set_tuning_knob (A/B tools, tests)
enum { KNOB_PLANES_VARIANT = 0, KNOB_P8_ORDER, KNOB_P8_NO_HALF, KNOB_P8_CLOCK_PRINT, KNOB_Q8_ORDER, KNOB_PAIRS_NO8, KNOB_PAIRS8_NO_KEPT, KNOB_Q8_KSPLIT, KNOB_ATTN_PAIRS_FLASH, KNOB_TN_WGS, KNOB_TN_XCD, KNOB_Q8_STREAM, KNOB_Q8_MIN_TILES, KNOB_SK_PERSIST, KNOB_ATTN_PAIRS_PERSIST, KNOB_PAIRS_NBUF, KNOB_Q4, KNOB_Q4_SMALL, KNOB_SPLIT_ROWS, KNOB_COUNT };
int tuning_knob(int which);

// The K-split workspace of the persistent GEMMs (gemm_pairs8.hip / gemm_planes8.hip: fp32 partials of the left-over tiles + one arrival
// counter per (tile, wave)).  CALLER-owned since ABI 7 (tt_linear_ksplit_workspace_bytes / tt_linear_ksplit_workspace_init): the counter block
// sits at the start of the buffer and must be zero when a launch begins - the init call zeroes it once, every launch leaves it zero (the
// wave that arrives last resets its counter).  A launch given no (or too small a) workspace does not split along K.
struct KsplitWs { float* partials; int* counters; };
size_t ksplit_ws_bytes();                                          // core.cpp: for the CURRENT device
size_t ksplit_ws_counter_bytes();
bool ksplit_ws_carve(void* ws, size_t bytes, KsplitWs* out);       // false: absent / too small / misaligned

// ---- device helpers -------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// max |.| of what a kernel wrote, published into a caller-owned device float (zeroed by the caller; non-negative floats order as unsigned
// integers: one relaxed atomic max per wave, order-independent = deterministic).  The PRODUCERS of a gradient leave it for the pair split
// of that gradient (tt_split_pairs_dual_parts amax_in: the power-of-two scale without a max pass of its own); NaNs are ignored as fmaxf
// ignores them - the split's range flag reports them.
// The "float" is kAmaxWays floats, kAmaxStride apart (different memory channels), and a workgroup uses way blockIdx.x % kAmaxWays: thousands
// of waves on ONE address serialise (measured: +0.2 ms on a C2 step with an atomic per wave on one address, +0.05 ms with a read in front
// of it - against the 0.11 ms of max passes this replaces).  A wave READS its way first and skips the atomic unless it would raise it.
constexpr int kAmaxWays = 16, kAmaxStride = 64;   // floats: a slot is kAmaxWays * kAmaxStride floats (4 KB), zeroed by the caller
__device__ __forceinline__ void amax_publish(float* slot, float m) {
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) {
    float* way = slot + (blockIdx.x % kAmaxWays) * kAmaxStride;
    const float cur = __hip_atomic_load(way, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (m > cur) __hip_atomic_fetch_max(reinterpret_cast<unsigned*>(way), __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// ... one access per WORKGROUP of up to 16 waves (every thread must call it; one barrier): for kernels whose waves finish together
__device__ __forceinline__ void amax_publish_block(float* slot, float m) {
  __shared__ float wg_max[16];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) wg_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (int)(blockDim.x + 63) >> 6;
    for (int w = 1; w < nw; ++w) m = fmaxf(m, wg_max[w]);
    float* way = slot + (blockIdx.x % kAmaxWays) * kAmaxStride;
    const float cur = __hip_atomic_load(way, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (m > cur) __hip_atomic_fetch_max(reinterpret_cast<unsigned*>(way), __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// exp for softmax probabilities: v_exp_f32 on x * log2(e) (a handful of instructions; ocml's expf is ~10x that and the
// softmax phase of the attention kernels is pure VALU time during which the matrix pipe idles).  Arguments are <= 0 and
// > -100 in practice; relative error ~1e-6, far inside the 1e-3 contract (the per-op tests hold 2e-5).
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }

// exact-erf GELU (nn.GELU default) and its derivative
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// GELU / GELU' of the bf16-plane epilogues (gemm_planes.hip): erf by Abramowitz-Stegun 7.1.26 (one v_rcp, one v_exp, five
// FMAs; |error| <= 1.5e-7 absolute on erf, i.e. <= 1e-7 |x| on GELU - inside the 2e-5 bound the fp32-accurate plane mode is
// tested at).  The f32-MFMA kernels keep ocml's erff (gelu_f / gelu_grad_f): measured, the cheaper erf buys them nothing.
__device__ __forceinline__ float erf_fast_f(float x, float* exp_neg_half_x2_out = nullptr) {   // erf(x / sqrt 2), |error| <= 1.5e-7
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));   // v_rcp_f32, 1 ulp (__frcp_rn is an 11-instruction IEEE division)
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float e = __expf(-z * z);   // = exp(-x^2 / 2): also the Gaussian of gelu'
  if (exp_neg_half_x2_out) *exp_neg_half_x2_out = e;
  return copysignf(1.0f - poly * e, x);
}
__device__ __forceinline__ float gelu_fast_f(float x) { return 0.5f * x * (1.0f + erf_fast_f(x)); }
// Two at a time, for the VALU-bound GELU epilogue of the pair GEMMs (gemm_pairs8.hip; ~100 issue cycles per element as the scalar form
// compiles): the same erf, written so that everything but the two transcendentals and the two |x| products is a packed fp32 instruction -
// gelu(x) = x / 2 + |x| erf(|x| / sqrt 2) / 2 with erf = 1 - poly(t) exp(-x^2 / 2): no copysign, no 1 + erf.
typedef float tt_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ tt_f32x2 gelu_fast_f2(tt_f32x2 x) {
  tt_f32x2 t, e, m;
  t[0] = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(x[0]), 1.0f));
  t[1] = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(x[1]), 1.0f));
  const tt_f32x2 x2 = x * x * -0.72134752044448170368f;   // -x^2 / 2 in log2 units
  e[0] = __builtin_amdgcn_exp2f(x2[0]);
  e[1] = __builtin_amdgcn_exp2f(x2[1]);
  tt_f32x2 poly = t * 1.061405429f + -1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t + -0.284496736f;
  poly = poly * t + 0.254829592f;
  const tt_f32x2 erf_abs = 1.0f - poly * t * e;
  m[0] = fabsf(x[0]) * erf_abs[0];
  m[1] = fabsf(x[1]) * erf_abs[1];
  return m * 0.5f + x * 0.5f;
}
// gelu'(x) = Phi(x) + x phi(x) with the same erf and its exp(-x^2 / 2) reused for phi
__device__ __forceinline__ float gelu_grad_fast_f(float x) {
  float g;
  const float cdf = 0.5f * (1.0f + erf_fast_f(x, &g));
  return cdf + x * 0.39894228040143267794f * g;
}

// GELU of the bf16-only path (gemm_planes8.hip, P = 1: the result is rounded to bf16, 2^-9 relative): x * sigmoid(2 u) with
// u = sqrt(2/pi) (x + 0.044715 x^3) - the tanh form of GELU, |error| <= 5e-4 absolute, one v_exp and one v_rcp instead of the
// erf ladder (a 128 x 64 wave tile pays 128 of these per lane in the epilogue: 36 instead of 90 issue cycles each).
__device__ __forceinline__ float gelu_bf16_f(float x) {
  const float x2 = x * x;
  const float u2 = x * fmaf(x2, -0.1029432f, -2.3022082f);           // -2 u log2(e)
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u2));
}

// ---- fp16 PAIRS: the operand format of the fp32-accurate split mode "f16x3" (gemm_pairs8.hip has the full description).
// x -> (hi, lo) with hi = fp16(x), lo = fp16((x - hi) * 2^11): x - hi is exact in fp32 (hi is x rounded to 11 significant bits), the
// scaling by a power of two is exact, so the only error is lo's own rounding, <= 2^-12 of |x - hi| <= 2^-23 |x|.  Subnormal hi / lo are
// kept by the conversion and by the MFMA (tools/probes/mfma_f16_probe.hip); |x| > 65504 gives hi = inf (loud, like fp16 autocast).
// Memory: groups of 32 consecutive elements as [hi x 32][lo x 32].
constexpr float kPairScale = 2048.0f, kPairInvScale = 0.00048828125f;
__device__ __forceinline__ void split_pair(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)((v - (float)hi) * kPairScale);
}
// hi of a value beyond fp16's range (|x| > 65504) or of a non-finite one is an infinity / a NaN: the one way the "f16x3" mode can differ
// from fp32 arithmetic.  Every kernel that PRODUCES pairs ORs this into the caller's range flag (include/timetuning_hip.h, "range flag").
__device__ __forceinline__ bool pair_hi_bad(_Float16 hi) { return (__builtin_bit_cast(unsigned short, hi) & 0x7fffu) >= 0x7c00u; }
__device__ __forceinline__ void range_flag_raise(int* flag, bool bad) {
  if (flag && bad) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float join_pair(_Float16 hi, _Float16 lo) { return fmaf((float)lo, kPairInvScale, (float)hi); }
// index of element i of a row-major array whose length is a multiple of 32, in the pair layout: hi at pair_index(i), lo 32 further
__host__ __device__ __forceinline__ long long pair_index(long long i) { return ((i >> 5) << 6) + (i & 31); }

// epilogue kinds of gemm_pairs8_kernel
// F32 / F32_RES: fp32 y (+ residual); PAIR / PAIR_GELU: y in pairs (after GELU); F32_GELUGRAD: fp32 y * gelu'(pre[m][n]) (a data-gradient
// product); BOTH: fp32 y AND the same value in pairs; BOTH_GELU: the fp32 PRE-activation and GELU of it in pairs (a forward that keeps
// what its backward needs)
enum { Q8_F32 = 0, Q8_F32_RES = 1, Q8_PAIR = 2, Q8_PAIR_GELU = 3, Q8_F32_GELUGRAD = 4, Q8_BOTH = 5, Q8_BOTH_GELU = 6 };

// XCD-aware bijective remap of a linear workgroup id (guide T1): the dispatcher deals consecutive
// ids round-robin over the 8 XCDs; this hands each XCD a contiguous run of logical tiles so that
// neighbouring tiles (which share an operand panel) hit the same 4 MiB L2.
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, slot = id >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

// 1-D grid decode for kernels where `inner` consecutive workgroups share an operand panel (the q-tiles of one
// (frame, head) share K and V): the sharing workgroups get block ids congruent mod 8 - the same XCD under the
// observed round-robin placement - in consecutive dispatch slots, so the panel is fetched into ONE L2 once instead of
// into up to `inner` of them (measured on attention_fwd: 349 MB -> see profiles/ of HBM-side reads per launch).
// Placement only affects speed.  Launch with xcd_group_grid(n_outer, inner) blocks; returns false for padding blocks.
__device__ __forceinline__ bool xcd_group_decode(int b, int inner, int n_outer, int& outer, int& in_idx) {
  const int x = b & 7, s = b >> 3;
  in_idx = s % inner;
  outer = (s / inner) * 8 + x;
  return outer < n_outer;
}
inline int xcd_group_grid(int n_outer, int inner) { return ((n_outer + 7) / 8) * 8 * inner; }


// out[n] = sum over chunks of partial[chunk][poff + n] for the 64 columns of block `blk`: the 4 waves split the chunks (fixed
// assignment and order: deterministic).  Shared by colsum_stage2 (rowops.hip) and the weight-gradient fold (gemm_f32.hip) so
// that a bias gradient has the same bits whichever launch folds it.  red: 4 x 64 floats of LDS.
__device__ __forceinline__ void colsum_fold_block(const float* __restrict__ partial, float* __restrict__ out, int chunks, int N, int pstride,
                                                  int poff, int blk, float (*red)[64]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = blk * 64 + lane;
  float s = 0.f;
  if (n < N) {
#pragma unroll 8
    for (int c = w; c < chunks; c += 4) s += partial[(long long)c * pstride + poff + n];
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && n < N) out[n] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

}  // namespace tt
