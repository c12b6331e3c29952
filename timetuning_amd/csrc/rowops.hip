// Row-wise and element-wise kernels (HBM-bound): LayerNorm fwd/bwd, L2 row normalisation fwd/bwd,
// deterministic column sums, cls-token rows, AdamW over a tensor table, EMA lerp, queue FIFO update.
// One 64-lane wave owns one row; a row (D <= 1024) is held in registers between the statistics pass
// and the output pass so that each element is read once and written once.
#include "common.hpp"

namespace tt {

constexpr int kMaxPerLane = 16;  // D <= 1024

// ------------------------------------------------------------------------------------------------
// LayerNorm (dino_vision_transformer.py:139,143,196)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int rows, int D, float eps, int skip_group) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  // skip_group = N: the input is [F][N][D] tokens and the output drops token 0 (cls) of every frame
  const long long in_row = skip_group ? (long long)(row / (skip_group - 1)) * skip_group + 1 + row % (skip_group - 1) : row;
  const float* xr = x + in_row * D;
  float v[kMaxPerLane];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    v[i] = (c < D) ? xr[c] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    const float d = (c < D) ? v[i] - mean : 0.f;
    q += d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  float* yr = y + (long long)row * D;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    if (c < D) yr[c] = (v[i] - mean) * rstd * gamma[c] + beta[c];
  }
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
}

// D = 128 NV (ViT-S 384, ViT-B 768, the 1024 / 512 / 256-wide head layers): 8-byte loads and stores, a compile-time trip count
// and R rows per wave in flight.  Same arithmetic; the per-lane partial sums group the elements differently.
template <int NV, int R>
__global__ __launch_bounds__(256) void layernorm_fwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ y,
                                                                float* __restrict__ mean_out, float* __restrict__ rstd_out, int rows,
                                                                float eps, int skip_group) {
  constexpr int D = 128 * NV;
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= rows) return;
  float2 v[R][NV];
  float s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    s[r] = 0.f;
    const int row = row0 + r < rows ? row0 + r : row0;
    const long long in_row = skip_group ? (long long)(row / (skip_group - 1)) * skip_group + 1 + row % (skip_group - 1) : row;
    const float2* xr = reinterpret_cast<const float2*>(x + in_row * D);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[r][i] = xr[lane + 64 * i];
      s[r] += v[r][i].x + v[r][i].y;
    }
  }
  float2 gm[NV], bt[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    gm[i] = reinterpret_cast<const float2*>(gamma)[lane + 64 * i];
    bt[i] = reinterpret_cast<const float2*>(beta)[lane + 64 * i];
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (row0 + r >= rows) break;
    const float mean = wave_sum(s[r]) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float dx = v[r][i].x - mean, dy = v[r][i].y - mean;
      q += dx * dx + dy * dy;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    float2* yr = reinterpret_cast<float2*>(y + (long long)(row0 + r) * D);
#pragma unroll
    for (int i = 0; i < NV; ++i)
      yr[lane + 64 * i] = make_float2((v[r][i].x - mean) * rstd * gm[i].x + bt[i].x, (v[r][i].y - mean) * rstd * gm[i].y + bt[i].y);
    if (lane == 0) {
      if (mean_out) mean_out[row0 + r] = mean;
      if (rstd_out) rstd_out[row0 + r] = rstd;
    }
  }
}

// The same with the result written as 1..3 bf16 planes (y = p0 + p1 + p2): the operand layout of gemm_planes.hip, so the
// consuming nn.Linear finds its A operand pre-split and nothing is converted on its path.
__global__ __launch_bounds__(256) void layernorm_fwd_planes_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, __bf16* __restrict__ y, long long stride,
                                                                   int planes, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                   int rows, int D, float eps, int skip_group) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long long in_row = skip_group ? (long long)(row / (skip_group - 1)) * skip_group + 1 + row % (skip_group - 1) : row;
  const float* xr = x + in_row * D;
  float v[kMaxPerLane];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    v[i] = (c < D) ? xr[c] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    const float d = (c < D) ? v[i] - mean : 0.f;
    q += d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  __bf16* yr = y + (long long)row * D;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    if (c < D) {
      const float o = (v[i] - mean) * rstd * gamma[c] + beta[c];
      const __bf16 p0 = (__bf16)o;
      yr[c] = p0;
      if (planes > 1) {
        const float r1 = o - (float)p0;
        const __bf16 p1 = (__bf16)r1;
        yr[stride + c] = p1;
        if (planes > 2) yr[2 * stride + c] = (__bf16)(r1 - (float)p1);
      }
    }
  }
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
}

// The plane-writing LayerNorm for D = 128 NV: 8-byte loads, 4-byte (two bf16) stores per plane, two rows per wave in flight.
template <int NV, int R>
__global__ __launch_bounds__(256) void layernorm_fwd_planes_vec_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                       const float* __restrict__ beta, __bf16* __restrict__ y, long long stride,
                                                                       int planes, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                       int rows, float eps, int skip_group) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  constexpr int D = 128 * NV;
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= rows) return;
  float2 v[R][NV];
  float s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    s[r] = 0.f;
    const int row = row0 + r < rows ? row0 + r : row0;
    const long long in_row = skip_group ? (long long)(row / (skip_group - 1)) * skip_group + 1 + row % (skip_group - 1) : row;
    const float2* xr = reinterpret_cast<const float2*>(x + in_row * D);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[r][i] = xr[lane + 64 * i];
      s[r] += v[r][i].x + v[r][i].y;
    }
  }
  float2 gm[NV], bt[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    gm[i] = reinterpret_cast<const float2*>(gamma)[lane + 64 * i];
    bt[i] = reinterpret_cast<const float2*>(beta)[lane + 64 * i];
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (row0 + r >= rows) break;
    const float mean = wave_sum(s[r]) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float dx = v[r][i].x - mean, dy = v[r][i].y - mean;
      q += dx * dx + dy * dy;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    bf16x2* yr = reinterpret_cast<bf16x2*>(y + (long long)(row0 + r) * D);
    const long long st2 = stride / 2;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float o0 = (v[r][i].x - mean) * rstd * gm[i].x + bt[i].x, o1 = (v[r][i].y - mean) * rstd * gm[i].y + bt[i].y;
      bf16x2 p0 = {(__bf16)o0, (__bf16)o1};
      yr[lane + 64 * i] = p0;
      if (planes > 1) {
        const float r0 = o0 - (float)p0[0], r1 = o1 - (float)p0[1];
        bf16x2 p1 = {(__bf16)r0, (__bf16)r1};
        yr[st2 + lane + 64 * i] = p1;
        if (planes > 2) {
          bf16x2 p2 = {(__bf16)(r0 - (float)p1[0]), (__bf16)(r1 - (float)p1[1])};
          yr[2 * st2 + lane + 64 * i] = p2;
        }
      }
    }
    if (lane == 0) {
      if (mean_out) mean_out[row0 + r] = mean;
      if (rstd_out) rstd_out[row0 + r] = rstd;
    }
  }
}

// LayerNorm whose result is written as fp16 PAIRS (common.hpp split_pair; the operand format of gemm_pairs8.hip): [rows][2 D] fp16,
// groups of 32 columns as [hi x 32][lo x 32].  D = 128 NV: 8-byte loads, two columns per lane -> 4-byte hi and lo stores.
template <int NV, int R>
__global__ __launch_bounds__(256) void layernorm_fwd_pairs_vec_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                      const float* __restrict__ beta, _Float16* __restrict__ y,
                                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out, int rows,
                                                                      float eps, int skip_group, int* range_flag) {
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  bool bad = false;
  constexpr int D = 128 * NV;
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= rows) return;
  float2 v[R][NV];
  float s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    s[r] = 0.f;
    const int row = row0 + r < rows ? row0 + r : row0;
    const long long in_row = skip_group ? (long long)(row / (skip_group - 1)) * skip_group + 1 + row % (skip_group - 1) : row;
    const float2* xr = reinterpret_cast<const float2*>(x + in_row * D);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[r][i] = xr[lane + 64 * i];
      s[r] += v[r][i].x + v[r][i].y;
    }
  }
  float2 gm[NV], bt[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    gm[i] = reinterpret_cast<const float2*>(gamma)[lane + 64 * i];
    bt[i] = reinterpret_cast<const float2*>(beta)[lane + 64 * i];
  }
  // columns 2 lane + 128 i, + 1: group (lane >> 4) + 4 i of the row, position 2 (lane & 15) inside it
  const int pos = ((lane >> 4) << 6) + 2 * (lane & 15);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (row0 + r >= rows) break;
    const float mean = wave_sum(s[r]) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float dx = v[r][i].x - mean, dy = v[r][i].y - mean;
      q += dx * dx + dy * dy;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    _Float16* yr = y + (long long)(row0 + r) * (2 * D) + pos;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float o0 = (v[r][i].x - mean) * rstd * gm[i].x + bt[i].x, o1 = (v[r][i].y - mean) * rstd * gm[i].y + bt[i].y;
      _Float16 h0, l0, h1, l1;
      split_pair(o0, h0, l0);
      split_pair(o1, h1, l1);
      bad = bad || pair_hi_bad(h0) || pair_hi_bad(h1);
      *reinterpret_cast<f16x2*>(yr + 256 * i) = (f16x2){h0, h1};
      *reinterpret_cast<f16x2*>(yr + 256 * i + 32) = (f16x2){l0, l1};
    }
    if (lane == 0) {
      if (mean_out) mean_out[row0 + r] = mean;
      if (rstd_out) rstd_out[row0 + r] = rstd;
    }
  }
  range_flag_raise(range_flag, bad);
}

// any D % 32 == 0 (<= 1024): one column per lane and pass
__global__ __launch_bounds__(256) void layernorm_fwd_pairs_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, _Float16* __restrict__ y,
                                                                  float* __restrict__ mean_out, float* __restrict__ rstd_out, int rows, int D,
                                                                  float eps, int skip_group, int* range_flag) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long long in_row = skip_group ? (long long)(row / (skip_group - 1)) * skip_group + 1 + row % (skip_group - 1) : row;
  const float* xr = x + in_row * D;
  bool bad = false;
  float v[kMaxPerLane];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    v[i] = (c < D) ? xr[c] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    const float d = (c < D) ? v[i] - mean : 0.f;
    q += d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  _Float16* yr = y + (long long)row * (2 * D);
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    if (c < D) {
      _Float16 hi, lo;
      split_pair((v[i] - mean) * rstd * gamma[c] + beta[c], hi, lo);
      bad |= pair_hi_bad(hi);
      yr[pair_index(c)] = hi;
      yr[pair_index(c) + 32] = lo;
    }
  }
  range_flag_raise(range_flag, bad);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
}

// Backward.  Each workgroup owns a contiguous run of rows; its 4 waves walk them, keep per-column partial
// sums of dgamma/dbeta in registers and combine them through LDS into partial[wg][2][D]; a column-sum
// pass over the partials finishes (deterministic: no atomics).
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dx,
                                                            float* __restrict__ partial, int rows, int D, int rows_per_wg,
                                                            int add_to_dx, int skip_group, float* __restrict__ amax_out) {
  __shared__ float red[4][2][64 * kMaxPerLane];
  float am = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.x * rows_per_wg;
  const int r1 = min(rows, r0 + rows_per_wg);
  float dg[kMaxPerLane], db[kMaxPerLane], gm[kMaxPerLane];
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    dg[i] = 0.f;
    db[i] = 0.f;
    const int c = lane + 64 * i;
    gm[i] = (c < D) ? gamma[c] : 0.f;
  }
  for (int row = r0 + wave; row < r1; row += 4) {
    const float mu = mean[row], rs = rstd[row];
    const long long in_row = skip_group ? (long long)(row / (skip_group - 1)) * skip_group + 1 + row % (skip_group - 1) : row;
    const float* xr = x + in_row * D;
    const float* dyr = dy + (long long)row * D;
    float xh[kMaxPerLane], gy[kMaxPerLane];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i) {
      const int c = lane + 64 * i;
      const float xv = (c < D) ? xr[c] : 0.f;
      const float dv = (c < D) ? dyr[c] : 0.f;
      xh[i] = (c < D) ? (xv - mu) * rs : 0.f;
      gy[i] = dv * gm[i];
      dg[i] += dv * xh[i];
      db[i] += dv;
      s1 += gy[i] * xh[i];
      s2 += gy[i];
    }
    const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
    float* dxr = dx + in_row * D;
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i) {
      const int c = lane + 64 * i;
      if (c < D) {
        const float o = rs * (gy[i] - c2 - xh[i] * c1);
        const float w = add_to_dx ? dxr[c] + o : o;
        dxr[c] = w;
        am = fmaxf(am, fabsf(w));
      }
    }
  }
  if (amax_out) amax_publish_block(amax_out, am);   // (uniform: every thread of the workgroup is here)
  if (!partial) return;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    red[wave][0][lane + 64 * i] = dg[i];
    red[wave][1][lane + 64 * i] = db[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a += red[w][0][c];
      b += red[w][1][c];
    }
    partial[((long long)blockIdx.x * 2 + 0) * D + c] = a;
    partial[((long long)blockIdx.x * 2 + 1) * D + c] = b;
  }
}

// The same for D = 128 NV with 8-byte accesses, a compile-time trip count and two rows per wave in flight.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, float* __restrict__ dx,
                                                                float* __restrict__ partial, int rows, int rows_per_wg, int add_to_dx,
                                                                int skip_group, float* __restrict__ amax_out) {
  constexpr int D = 128 * NV;
  float am = 0.f;
  __shared__ float2 red[4][2][64 * NV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.x * rows_per_wg;
  const int r1 = min(rows, r0 + rows_per_wg);
  float2 dg[NV], db[NV], gm[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    dg[i] = make_float2(0.f, 0.f);
    db[i] = make_float2(0.f, 0.f);
    gm[i] = reinterpret_cast<const float2*>(gamma)[lane + 64 * i];
  }
  for (int rowa = r0 + wave; rowa < r1; rowa += 8) {
    float2 xh[2][NV], gy[2][NV];
    float s1[2], s2[2], rs[2];
    long long in_row[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = rowa + 4 * r < r1 ? rowa + 4 * r : rowa;
      const float mu = mean[row];
      rs[r] = rstd[row];
      in_row[r] = skip_group ? (long long)(row / (skip_group - 1)) * skip_group + 1 + row % (skip_group - 1) : row;
      const float2* xr = reinterpret_cast<const float2*>(x + in_row[r] * D);
      const float2* dyr = reinterpret_cast<const float2*>(dy + (long long)row * D);
      s1[r] = 0.f;
      s2[r] = 0.f;
      const bool live = r == 0 || rowa + 4 < r1;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float2 xv = xr[lane + 64 * i], dv = dyr[lane + 64 * i];
        xh[r][i] = make_float2((xv.x - mu) * rs[r], (xv.y - mu) * rs[r]);
        gy[r][i] = make_float2(dv.x * gm[i].x, dv.y * gm[i].y);
        if (live) {
          dg[i].x += dv.x * xh[r][i].x; dg[i].y += dv.y * xh[r][i].y;
          db[i].x += dv.x; db[i].y += dv.y;
        }
        s1[r] += gy[r][i].x * xh[r][i].x + gy[r][i].y * xh[r][i].y;
        s2[r] += gy[r][i].x + gy[r][i].y;
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (r == 1 && rowa + 4 >= r1) break;
      const float c1 = wave_sum(s1[r]) / (float)D, c2 = wave_sum(s2[r]) / (float)D;
      float2* dxr = reinterpret_cast<float2*>(dx + in_row[r] * D);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        float2 o = make_float2(rs[r] * (gy[r][i].x - c2 - xh[r][i].x * c1), rs[r] * (gy[r][i].y - c2 - xh[r][i].y * c1));
        if (add_to_dx) {
          const float2 old = dxr[lane + 64 * i];
          o.x += old.x; o.y += old.y;
        }
        dxr[lane + 64 * i] = o;
        am = fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y)));
      }
    }
  }
  if (amax_out) amax_publish_block(amax_out, am);   // (uniform: every thread of the workgroup is here)
  if (!partial) return;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    red[wave][0][lane + 64 * i] = dg[i];
    red[wave][1][lane + 64 * i] = db[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D / 2; c += 256) {
    float2 a = make_float2(0.f, 0.f), b = make_float2(0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a.x += red[w][0][c].x; a.y += red[w][0][c].y;
      b.x += red[w][1][c].x; b.y += red[w][1][c].y;
    }
    reinterpret_cast<float2*>(partial + ((long long)blockIdx.x * 2 + 0) * D)[c] = a;
    reinterpret_cast<float2*>(partial + ((long long)blockIdx.x * 2 + 1) * D)[c] = b;
  }
}

// ------------------------------------------------------------------------------------------------
// Column sums: out[n] = sum_m a[m][n] (row stride lda).  Stage 1 writes partial[chunk][n]; stage 2 folds.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ a, float* __restrict__ partial, int M, int N,
                                                     int lda, int rows_per_chunk) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
  float s = 0.f;
  if (n < N)
    for (int m = m0 + rg; m < m1; m += 4) s += a[(long long)m * lda + n];
  red[rg][lane] = s;
  __syncthreads();
  if (rg == 0 && n < N) partial[(long long)blockIdx.y * N + n] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}
__global__ __launch_bounds__(256) void colsum_stage2(const float* __restrict__ partial, float* __restrict__ out, int chunks,
                                                     int N, int pstride, int poff) {
  // 64 columns per workgroup (common.hpp: colsum_fold_block)
  __shared__ float red[4][64];
  colsum_fold_block(partial, out, chunks, N, pstride, poff, blockIdx.x, red);
}

// dgamma and dbeta of the LayerNorm backward in ONE launch: partial[wg][2][D] -> dgamma[D] (y = 0), dbeta[D] (y = 1).  SIXTEEN waves
// split the chunks of a 64-column block (round 6: the launch sits on the data-gradient chain - 788 chunks at C2 were 197 dependent
// batches of loads per thread with four waves, 12.5 us; fixed assignment and order: deterministic).
__global__ __launch_bounds__(1024) void colsum_stage2_pair(const float* __restrict__ partial, float* __restrict__ out0, float* __restrict__ out1,
                                                           int chunks, int N) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  const int poff = blockIdx.y ? N : 0;
  float s = 0.f;
  if (n < N) {   // (whole batches of 16 loads in flight, then singly: same order)
    int c = w;
    for (; c + 16 * 15 < chunks; c += 16 * 16) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = partial[(long long)(c + 16 * j) * 2 * N + poff + n];
#pragma unroll
      for (int j = 0; j < 16; ++j) s += v[j];
    }
    for (; c < chunks; c += 16) s += partial[(long long)c * 2 * N + poff + n];
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i += 4) t += (red[i][lane] + red[i + 1][lane]) + (red[i + 2][lane] + red[i + 3][lane]);
    (blockIdx.y ? out1 : out0)[n] = t;
  }
}

// fold of partial[chunks][N] into out[N] for other translation units (the weight-gradient GEMM's fused bias gradient)
int launch_colsum_fold(const float* partial, float* out, int chunks, int N, hipStream_t s) {
  hipLaunchKernelGGL(colsum_stage2, dim3((N + 63) / 64), dim3(256), 0, s, partial, out, chunks, N, N, 0);
  TT_CHECK_LAUNCH("colsum_fold");
  return TT_OK;
}

static int colsum_chunks(int M) {
  int chunks = (M + 255) / 256;
  return chunks > 128 ? 128 : (chunks < 1 ? 1 : chunks);
}

// ------------------------------------------------------------------------------------------------
// F.normalize(x, dim=-1)  (time_tuning.py:136, :124-128; mask_propagation.py:418-419)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* x, int ldx, float* xn, float* __restrict__ inv_norm,
                                                         int rows, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long long)row * ldx;
  float v[kMaxPerLane];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    v[i] = (c < D) ? xr[c] : 0.f;
    s += v[i] * v[i];
  }
  const float nrm = sqrtf(wave_sum(s));
  const float inv = 1.0f / fmaxf(nrm, 1e-12f);
  float* yr = xn + (long long)row * D;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    if (c < D) yr[c] = v[i] * inv;
  }
  if (inv_norm && lane == 0) inv_norm[row] = inv;
}

__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dxn, const float* __restrict__ xn,
                                                         const float* __restrict__ inv_norm, float* __restrict__ dx,
                                                         int rows, int D, float* __restrict__ amax_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float am = 0.f;
  float a[kMaxPerLane], b[kMaxPerLane];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    a[i] = (c < D) ? dxn[(long long)row * D + c] : 0.f;
    b[i] = (c < D) ? xn[(long long)row * D + c] : 0.f;
    s += a[i] * b[i];
  }
  const float dot = wave_sum(s), inv = inv_norm[row];
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    if (c < D) {
      const float w = (a[i] - b[i] * dot) * inv;
      dx[(long long)row * D + c] = w;
      am = fmaxf(am, fabsf(w));
    }
  }
  if (amax_out) amax_publish(amax_out, am);
}

// cls rows of the token tensor: tokens[f][0][:] = cls + pos[0] (dino_vision_transformer.py:241-245)
__global__ void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ tokens, int F,
                                int D, long long frame_stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F * D) return;
  const int f = i / D, c = i - f * D;
  tokens[(long long)f * frame_stride + c] = cls[c] + pos[c];
}

__global__ void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] += src[i];
}

// ------------------------------------------------------------------------------------------------
// AdamW (torch.optim.AdamW single-tensor semantics, time_tuning.py:413-429) over a table of tensors.
// ------------------------------------------------------------------------------------------------
struct AdamTable {
  tt_adamw_tensor t[TT_MAX_TENSORS];
};
__global__ __launch_bounds__(256) void adamw_kernel(AdamTable tab, float beta1, float beta2, float eps, float bc1,
                                                    float bc2_sqrt) {
  const tt_adamw_tensor t = tab.t[blockIdx.y];
  const float decay = 1.0f - t.lr * t.weight_decay;
  const float step_size = t.lr / bc1;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < t.n; i += (long long)gridDim.x * 256) {
    const float g = t.g[i];
    float p = t.p[i] * decay;
    float m = t.m[i];
    m = m + (g - m) * (1.0f - beta1);                 // exp_avg.lerp_(grad, 1 - beta1)
    const float v = t.v[i] * beta2 + (1.0f - beta2) * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p -= step_size * (m / denom);
    t.p[i] = p;
    t.m[i] = m;
    t.v[i] = v;
  }
}

// g[i] *= *scale for every tensor of the table (the incoming gradient of the loss, a device scalar, applied to the gradients
// the fused step already holds: what autograd's chain rule asks of _FusedLoss.backward)
__global__ __launch_bounds__(256) void scale_tensors_kernel(AdamTable tab, const float* __restrict__ scale) {
  const tt_adamw_tensor t = tab.t[blockIdx.y];
  const float sc = *scale;
  float* gp = const_cast<float*>(t.g);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < t.n; i += (long long)gridDim.x * 256) gp[i] *= sc;
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ t, const float* __restrict__ s, long long n, float m,
                                                  float one_minus_m) {
  long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  const long long stride = (long long)gridDim.x * 256 * 4;
  for (; i + 3 < n; i += stride) {
    float4 a = *reinterpret_cast<float4*>(t + i);
    const float4 b = *reinterpret_cast<const float4*>(s + i);
    a.x = a.x * one_minus_m + b.x * m;
    a.y = a.y * one_minus_m + b.y * m;
    a.z = a.z * one_minus_m + b.z * m;
    a.w = a.w * one_minus_m + b.w * m;
    *reinterpret_cast<float4*>(t + i) = a;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long j = (n & ~3LL) + threadIdx.x;
    t[j] = t[j] * one_minus_m + s[j] * m;
  }
}

// queue FIFO (time_tuning.py:258-261): new[r] = r < m ? feats[idx[r]] : old[r - m]
__global__ void queue_push_kernel(const float* __restrict__ old_q, float* __restrict__ new_q, const float* __restrict__ feats,
                                  const int64_t* __restrict__ idx, int Q, int D, int m) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)Q * D) return;
  const int r = (int)(i / D), c = (int)(i - (long long)r * D);
  new_q[i] = (r < m) ? feats[idx[r] * (long long)D + c] : old_q[(long long)(r - m) * D + c];
}

}  // namespace tt

using namespace tt;

extern "C" int tt_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                int rows, int D, float eps, int skip_group, tt_stream_t stream) {
  TT_REQUIRE(skip_group == 0 || (skip_group >= 2 && rows % (skip_group - 1) == 0), "layernorm_fwd: rows must be a multiple of skip_group - 1");
  TT_REQUIRE(x && gamma && beta && y, "layernorm_fwd: null pointer");
  TT_REQUIRE(rows > 0 && D > 0 && D <= 64 * kMaxPerLane, "layernorm_fwd: need 0 < D <= %d (got %d)", 64 * kMaxPerLane, D);
  const bool al8 = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gamma) |
                     reinterpret_cast<uintptr_t>(beta)) & 7u) == 0;
  if (al8 && D % 128 == 0 && D <= 1024) {
    constexpr int R = 2;   // rows in flight per wave: 14.8 / 13.4 / 13.6 us at 1 / 2 / 4 on 25216 x 384 (15.8 for the scalar kernel)
    const dim3 grid((rows + 4 * R - 1) / (4 * R)), block(256);
    hipStream_t s = as_stream(stream);
    switch (D / 128) {
      case 1: hipLaunchKernelGGL((layernorm_fwd_vec_kernel<1, R>), grid, block, 0, s, x, gamma, beta, y, mean, rstd, rows, eps, skip_group); break;
      case 2: hipLaunchKernelGGL((layernorm_fwd_vec_kernel<2, R>), grid, block, 0, s, x, gamma, beta, y, mean, rstd, rows, eps, skip_group); break;
      case 3: hipLaunchKernelGGL((layernorm_fwd_vec_kernel<3, R>), grid, block, 0, s, x, gamma, beta, y, mean, rstd, rows, eps, skip_group); break;
      case 4: hipLaunchKernelGGL((layernorm_fwd_vec_kernel<4, R>), grid, block, 0, s, x, gamma, beta, y, mean, rstd, rows, eps, skip_group); break;
      case 6: hipLaunchKernelGGL((layernorm_fwd_vec_kernel<6, R>), grid, block, 0, s, x, gamma, beta, y, mean, rstd, rows, eps, skip_group); break;
      case 8: hipLaunchKernelGGL((layernorm_fwd_vec_kernel<8, R>), grid, block, 0, s, x, gamma, beta, y, mean, rstd, rows, eps, skip_group); break;
      default: goto general;
    }
    TT_CHECK_LAUNCH("layernorm_fwd");
    return TT_OK;
  }
general:
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), x, gamma, beta, y, mean, rstd,
                     rows, D, eps, skip_group);
  TT_CHECK_LAUNCH("layernorm_fwd");
  return TT_OK;
}

extern "C" int tt_layernorm_fwd_planes(const float* x, const float* gamma, const float* beta, void* y_planes, long long plane_stride,
                                       int planes, float* mean, float* rstd, int rows, int D, float eps, int skip_group,
                                       tt_stream_t stream) {
  TT_REQUIRE(skip_group == 0 || (skip_group >= 2 && rows % (skip_group - 1) == 0), "layernorm_fwd_planes: rows must be a multiple of skip_group - 1");
  TT_REQUIRE(x && gamma && beta && y_planes && planes >= 1 && planes <= 3, "layernorm_fwd_planes: null pointer / planes not in 1..3");
  TT_REQUIRE(rows > 0 && D > 0 && D <= 64 * kMaxPerLane, "layernorm_fwd_planes: need 0 < D <= %d (got %d)", 64 * kMaxPerLane, D);
  const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 7u) == 0 &&
                  (reinterpret_cast<uintptr_t>(y_planes) & 3u) == 0 && plane_stride % 2 == 0;
  if (al && (D == 384 || D == 768 || D == 128 || D == 256 || D == 512 || D == 1024)) {
    constexpr int R = 2;
    const dim3 grid((rows + 4 * R - 1) / (4 * R)), block(256);
    hipStream_t s = as_stream(stream);
    __bf16* yp = static_cast<__bf16*>(y_planes);
    switch (D / 128) {
      case 1: hipLaunchKernelGGL((layernorm_fwd_planes_vec_kernel<1, R>), grid, block, 0, s, x, gamma, beta, yp, plane_stride, planes, mean, rstd, rows, eps, skip_group); break;
      case 2: hipLaunchKernelGGL((layernorm_fwd_planes_vec_kernel<2, R>), grid, block, 0, s, x, gamma, beta, yp, plane_stride, planes, mean, rstd, rows, eps, skip_group); break;
      case 3: hipLaunchKernelGGL((layernorm_fwd_planes_vec_kernel<3, R>), grid, block, 0, s, x, gamma, beta, yp, plane_stride, planes, mean, rstd, rows, eps, skip_group); break;
      case 4: hipLaunchKernelGGL((layernorm_fwd_planes_vec_kernel<4, R>), grid, block, 0, s, x, gamma, beta, yp, plane_stride, planes, mean, rstd, rows, eps, skip_group); break;
      case 6: hipLaunchKernelGGL((layernorm_fwd_planes_vec_kernel<6, R>), grid, block, 0, s, x, gamma, beta, yp, plane_stride, planes, mean, rstd, rows, eps, skip_group); break;
      default: hipLaunchKernelGGL((layernorm_fwd_planes_vec_kernel<8, R>), grid, block, 0, s, x, gamma, beta, yp, plane_stride, planes, mean, rstd, rows, eps, skip_group); break;
    }
    TT_CHECK_LAUNCH("layernorm_fwd_planes");
    return TT_OK;
  }
  hipLaunchKernelGGL(layernorm_fwd_planes_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), x, gamma, beta,
                     static_cast<__bf16*>(y_planes), plane_stride, planes, mean, rstd, rows, D, eps, skip_group);
  TT_CHECK_LAUNCH("layernorm_fwd_planes");
  return TT_OK;
}

extern "C" int tt_layernorm_fwd_pairs(const float* x, const float* gamma, const float* beta, void* y_pairs, float* mean, float* rstd, int rows,
                                      int D, float eps, int skip_group, int* range_flag, tt_stream_t stream) {
  TT_REQUIRE(skip_group == 0 || (skip_group >= 2 && rows % (skip_group - 1) == 0), "layernorm_fwd_pairs: rows must be a multiple of skip_group - 1");
  TT_REQUIRE(x && gamma && beta && y_pairs, "layernorm_fwd_pairs: null pointer");
  TT_REQUIRE(rows > 0 && D > 0 && D % 32 == 0 && D <= 64 * kMaxPerLane, "layernorm_fwd_pairs: need D %% 32 == 0 and 0 < D <= %d (got %d)",
             64 * kMaxPerLane, D);
  const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 7u) == 0 &&
                  (reinterpret_cast<uintptr_t>(y_pairs) & 3u) == 0;
  _Float16* yp = static_cast<_Float16*>(y_pairs);
  hipStream_t s = as_stream(stream);
  if (al && (D == 384 || D == 768 || D == 128 || D == 256 || D == 512 || D == 1024)) {
    constexpr int R = 2;
    const dim3 grid((rows + 4 * R - 1) / (4 * R)), block(256);
    switch (D / 128) {
      case 1: hipLaunchKernelGGL((layernorm_fwd_pairs_vec_kernel<1, R>), grid, block, 0, s, x, gamma, beta, yp, mean, rstd, rows, eps, skip_group, range_flag); break;
      case 2: hipLaunchKernelGGL((layernorm_fwd_pairs_vec_kernel<2, R>), grid, block, 0, s, x, gamma, beta, yp, mean, rstd, rows, eps, skip_group, range_flag); break;
      case 3: hipLaunchKernelGGL((layernorm_fwd_pairs_vec_kernel<3, R>), grid, block, 0, s, x, gamma, beta, yp, mean, rstd, rows, eps, skip_group, range_flag); break;
      case 4: hipLaunchKernelGGL((layernorm_fwd_pairs_vec_kernel<4, R>), grid, block, 0, s, x, gamma, beta, yp, mean, rstd, rows, eps, skip_group, range_flag); break;
      case 6: hipLaunchKernelGGL((layernorm_fwd_pairs_vec_kernel<6, R>), grid, block, 0, s, x, gamma, beta, yp, mean, rstd, rows, eps, skip_group, range_flag); break;
      default: hipLaunchKernelGGL((layernorm_fwd_pairs_vec_kernel<8, R>), grid, block, 0, s, x, gamma, beta, yp, mean, rstd, rows, eps, skip_group, range_flag); break;
    }
    TT_CHECK_LAUNCH("layernorm_fwd_pairs");
    return TT_OK;
  }
  hipLaunchKernelGGL(layernorm_fwd_pairs_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, gamma, beta, yp, mean, rstd, rows, D, eps, skip_group, range_flag);
  TT_CHECK_LAUNCH("layernorm_fwd_pairs");
  return TT_OK;
}

static int ln_bwd_wgs(int rows) {
  // 8 rows per workgroup (two per wave, both in flight) until the fold of the per-workgroup dgamma / dbeta partials would outgrow
  // the pass itself: 6304 rows -> 788 workgroups (26 -> 11 us at D = 384 against 197 workgroups of 32 rows)
  int w = (rows + 7) / 8;
  return w > 1024 ? 1024 : (w < 1 ? 1 : w);
}
extern "C" size_t tt_layernorm_bwd_workspace_bytes(int rows, int D) { return (size_t)ln_bwd_wgs(rows) * 2 * D * sizeof(float); }

extern "C" int tt_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                float* dx, float* dgamma, float* dbeta, int rows, int D, int add_to_dx, int skip_group,
                                void* workspace, size_t workspace_bytes, float* amax_out, tt_stream_t stream) {
  TT_REQUIRE(skip_group == 0 || (skip_group >= 2 && rows % (skip_group - 1) == 0), "layernorm_bwd: rows must be a multiple of skip_group - 1");
  TT_REQUIRE(dy && x && gamma && mean && rstd && dx, "layernorm_bwd: null pointer");
  TT_REQUIRE(rows > 0 && D > 0 && D <= 64 * kMaxPerLane, "layernorm_bwd: need 0 < D <= %d", 64 * kMaxPerLane);
  const bool want = dgamma || dbeta;
  TT_REQUIRE(!want || (dgamma && dbeta), "layernorm_bwd: dgamma and dbeta must be given together");
  const int wgs = ln_bwd_wgs(rows);
  const int rpw = (rows + wgs - 1) / wgs;
  if (want) TT_REQUIRE(workspace && workspace_bytes >= tt_layernorm_bwd_workspace_bytes(rows, D), "layernorm_bwd: workspace too small");
  float* partial = want ? static_cast<float*>(workspace) : nullptr;
  const bool al8 = ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) |
                     reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(partial)) & 7u) == 0;
  hipStream_t s = as_stream(stream);
  const dim3 grid(wgs), block(256);
  if (al8 && D == 384) hipLaunchKernelGGL((layernorm_bwd_vec_kernel<3>), grid, block, 0, s, dy, x, gamma, mean, rstd, dx, partial, rows, rpw, add_to_dx, skip_group, amax_out);
  else if (al8 && D == 768) hipLaunchKernelGGL((layernorm_bwd_vec_kernel<6>), grid, block, 0, s, dy, x, gamma, mean, rstd, dx, partial, rows, rpw, add_to_dx, skip_group, amax_out);
  else if (al8 && D == 128) hipLaunchKernelGGL((layernorm_bwd_vec_kernel<1>), grid, block, 0, s, dy, x, gamma, mean, rstd, dx, partial, rows, rpw, add_to_dx, skip_group, amax_out);
  else
    hipLaunchKernelGGL(layernorm_bwd_kernel, grid, block, 0, s, dy, x, gamma, mean, rstd, dx, partial, rows, D, rpw, add_to_dx, skip_group, amax_out);
  TT_CHECK_LAUNCH("layernorm_bwd");
  if (want) {
    hipLaunchKernelGGL(colsum_stage2_pair, dim3((D + 63) / 64, 2), dim3(1024), 0, as_stream(stream), partial, dgamma, dbeta, wgs, D);
    TT_CHECK_LAUNCH("layernorm_bwd.reduce");
  }
  return TT_OK;
}

extern "C" size_t tt_colsum_workspace_bytes(int M, int N) { return (size_t)colsum_chunks(M) * N * sizeof(float); }

extern "C" int tt_colsum(const float* a, float* out, int M, int N, void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  TT_REQUIRE(a && out && M > 0 && N > 0, "colsum: bad arguments");
  TT_REQUIRE(workspace && workspace_bytes >= tt_colsum_workspace_bytes(M, N), "colsum: workspace too small");
  const int chunks = colsum_chunks(M);
  const int rpc = (M + chunks - 1) / chunks;
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(colsum_stage1, dim3((N + 63) / 64, chunks), dim3(256), 0, as_stream(stream), a, partial, M, N, N, rpc);
  hipLaunchKernelGGL(colsum_stage2, dim3((N + 63) / 64), dim3(256), 0, as_stream(stream), partial, out, chunks, N, N, 0);
  TT_CHECK_LAUNCH("colsum");
  return TT_OK;
}

extern "C" int tt_l2norm_fwd(const float* x, int ldx, float* xn, float* inv_norm, int rows, int D, tt_stream_t stream) {
  TT_REQUIRE(x && xn && rows > 0 && D > 0 && D <= 64 * kMaxPerLane && ldx >= D, "l2norm_fwd: bad arguments (D=%d ldx=%d)", D, ldx);
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), x, ldx, xn, inv_norm, rows, D);
  TT_CHECK_LAUNCH("l2norm_fwd");
  return TT_OK;
}

extern "C" int tt_l2norm_bwd(const float* dxn, const float* xn, const float* inv_norm, float* dx, int rows, int D, float* amax_out,
                             tt_stream_t stream) {
  TT_REQUIRE(dxn && xn && inv_norm && dx && rows > 0 && D > 0 && D <= 64 * kMaxPerLane, "l2norm_bwd: bad arguments");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), dxn, xn, inv_norm, dx, rows, D, amax_out);
  TT_CHECK_LAUNCH("l2norm_bwd");
  return TT_OK;
}

extern "C" int tt_normalize_rows_inplace(float* w, int rows, int D, tt_stream_t stream) {
  return tt_l2norm_fwd(w, D, w, nullptr, rows, D, stream);
}

extern "C" int tt_patch_embed_gemm(const float* img, const int32_t* frame_map, const float* w, const float* bias, const float* pos,
                                   float* tokens, int F, int C, int H, int W, int P, int D, tt_stream_t stream);

extern "C" int tt_patch_embed_fwd(const float* img, const int32_t* frame_map, const float* w, const float* bias, const float* cls,
                                  const float* pos, float* tokens, int F, int C, int H, int W, int P, int D, tt_stream_t stream) {
  TT_REQUIRE(img && w && bias && cls && pos && tokens, "patch_embed: null pointer");
  TT_REQUIRE(F > 0 && P > 0 && H % P == 0 && W % P == 0, "patch_embed: H, W must be multiples of the patch size");
  const int n = (H / P) * (W / P);
  int rc = tt_patch_embed_gemm(img, frame_map, w, bias, pos, tokens, F, C, H, W, P, D, stream);
  if (rc != TT_OK) return rc;
  hipLaunchKernelGGL(cls_rows_kernel, dim3((F * D + 255) / 256), dim3(256), 0, as_stream(stream), cls, pos, tokens, F, D,
                     (long long)(n + 1) * D);
  TT_CHECK_LAUNCH("patch_embed.cls");
  return TT_OK;
}

extern "C" int tt_add_inplace(float* dst, const float* src, long long n, tt_stream_t stream) {
  TT_REQUIRE(dst && src && n > 0, "add_inplace: bad arguments");
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), dst, src, n);
  TT_CHECK_LAUNCH("add_inplace");
  return TT_OK;
}

__global__ __launch_bounds__(256) void count_mismatch_kernel(const unsigned* __restrict__ a, const unsigned* __restrict__ b, long long n,
                                                              unsigned long long* __restrict__ out) {
  unsigned long long c = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) c += a[i] != b[i];
  unsigned lo = (unsigned)c;  // per-thread count < 2^32 for any n this is used on
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) lo += __shfl_xor(lo, o, 64);
  if ((threadIdx.x & 63) == 0 && lo) atomicAdd(out, (unsigned long long)lo);
}

extern "C" int tt_count_mismatch(const float* a, const float* b, long long n, long long* count_out, tt_stream_t stream) {
  TT_REQUIRE(a && b && count_out && n > 0, "count_mismatch: bad arguments");
  hipError_t e = hipMemsetAsync(count_out, 0, sizeof(long long), as_stream(stream));
  if (e != hipSuccess) {
    set_error("count_mismatch: memset failed: %s", hipGetErrorString(e));
    return TT_ELAUNCH;
  }
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(count_mismatch_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), reinterpret_cast<const unsigned*>(a),
                     reinterpret_cast<const unsigned*>(b), n, reinterpret_cast<unsigned long long*>(count_out));
  TT_CHECK_LAUNCH("count_mismatch");
  return TT_OK;
}

extern "C" int tt_adamw_step(const tt_adamw_tensor* tensors, int count, int step, float beta1, float beta2, float eps,
                             tt_stream_t stream) {
  TT_REQUIRE(tensors && count > 0 && count <= TT_MAX_TENSORS && step >= 1, "adamw: need 1..%d tensors and step >= 1", TT_MAX_TENSORS);
  AdamTable tab{};
  long long maxn = 0;
  for (int i = 0; i < count; ++i) {
    TT_REQUIRE(tensors[i].p && tensors[i].g && tensors[i].m && tensors[i].v && tensors[i].n > 0, "adamw: tensor %d has a null pointer", i);
    tab.t[i] = tensors[i];
    if (tensors[i].n > maxn) maxn = tensors[i].n;
  }
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  long long bx = (maxn + 255) / 256;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)bx, count), dim3(256), 0, as_stream(stream), tab, beta1, beta2, eps, (float)bc1,
                     (float)sqrt(bc2));
  TT_CHECK_LAUNCH("adamw");
  return TT_OK;
}

extern "C" int tt_scale_tensors(const tt_adamw_tensor* tensors, int count, const float* scale_device, tt_stream_t stream) {
  TT_REQUIRE(tensors && scale_device && count > 0 && count <= TT_MAX_TENSORS, "scale_tensors: need 1..%d tensors", TT_MAX_TENSORS);
  AdamTable tab{};
  long long maxn = 0;
  for (int i = 0; i < count; ++i) {
    TT_REQUIRE(tensors[i].g && tensors[i].n > 0, "scale_tensors: tensor %d has a null gradient pointer", i);
    tab.t[i] = tensors[i];
    if (tensors[i].n > maxn) maxn = tensors[i].n;
  }
  long long bx = (maxn + 255) / 256;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(scale_tensors_kernel, dim3((unsigned)bx, count), dim3(256), 0, as_stream(stream), tab, scale_device);
  TT_CHECK_LAUNCH("scale_tensors");
  return TT_OK;
}

extern "C" int tt_ema_update(float* teacher, const float* student, long long n, double momentum, tt_stream_t stream) {
  TT_REQUIRE(teacher && student && n > 0, "ema: bad arguments");
  TT_REQUIRE(aligned16(teacher) && aligned16(student), "ema: buffers must be 16-byte aligned");
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(ema_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), teacher, student, n, (float)momentum,
                     (float)(1.0 - momentum));
  TT_CHECK_LAUNCH("ema");
  return TT_OK;
}

extern "C" int tt_queue_push(float* queue, float* scratch, const float* feats, const int64_t* idx, int Q, int D, int m,
                             tt_stream_t stream) {
  TT_REQUIRE(queue && scratch && feats && idx && Q > 0 && D > 0 && m > 0 && m <= Q, "queue_push: bad arguments");
  const long long n = (long long)Q * D;
  hipLaunchKernelGGL(queue_push_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), queue, scratch, feats, idx,
                     Q, D, m);
  TT_CHECK_LAUNCH("queue_push");
  hipError_t e = hipMemcpyAsync(queue, scratch, n * sizeof(float), hipMemcpyDeviceToDevice, as_stream(stream));
  if (e != hipSuccess) {
    set_error("queue_push: copy failed: %s", hipGetErrorString(e));
    return TT_ELAUNCH;
  }
  return TT_OK;
}
