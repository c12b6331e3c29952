// Fused multi-head self-attention for ViT token counts (N <= 256, head_dim = 64), fp32 on
// v_mfma_f32_16x16x4_f32.  Replaces q@k^T*scale -> softmax -> @v (dino_vision_transformer.py:125-129)
// without ever writing the N x N matrix to HBM (the reference materialises attn[F,h,N,N] per layer).
//
// Layout trick: each wave owns 16 query rows and computes S^T = K Q^T, so the MFMA C/D layout puts the
// query on the lane (col = lane & 15) and the keys on registers (key = 16*tile + 4*(lane >> 4) + e).  The whole
// score row of a query (<= 256 keys = 64 registers) therefore lives in one lane-column: the softmax needs two
// cross-lane steps (xor 16, xor 32) instead of a 64-lane reduction, and P^T is already in B-operand layout for
// O^T = V^T P^T (the accumulator feeds the next MFMA with no LDS round trip and no shuffle).
//
// A workgroup = 4 waves = 64 queries of one (frame, head); K and V stream through LDS in 32-key chunks
// (double-buffered, next chunk's global loads in flight during the current chunk's MFMAs).  q/k/v are read
// straight out of the qkv Linear's [F, N, 3*H*64] output (256-byte contiguous head rows): no permute kernel.
#include "common.hpp"
#include <cstdlib>

namespace tt {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HD = 64;
constexpr int KCH = 32;        // keys per LDS chunk
constexpr int KSTR = 66;       // K row stride (floats): (2*i + g) distinct banks for the A-operand read
constexpr int VSTR = 68;       // V row stride: 16*g + i distinct banks

template <int NT>  // NT = number of 16-key tiles kept in registers (N <= 16 * NT)
__global__ __launch_bounds__(256) void attention_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            float* __restrict__ lse, float* __restrict__ probs, int N, int H,
                                                            int FH, float scale) {
  constexpr int NC = (NT + 1) / 2;  // chunks of 32 keys
  __shared__ __attribute__((aligned(16))) float smem[2 * KCH * KSTR + 2 * KCH * VSTR];
  float* Ks = smem;                   // [2][KCH][KSTR]
  float* Vs = smem + 2 * KCH * KSTR;  // [2][KCH][VSTR]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = lane & 15, g = lane >> 4;
  int fh, qtile;
  if (!xcd_group_decode(blockIdx.x, (N + 63) / 64, FH, fh, qtile)) return;  // q-tiles of one (frame, head) share an XCD
  const int f = fh / H, h = fh - f * H;
  const int D3 = 3 * H * HD;
  const float* base = qkv + (long long)f * N * D3 + h * HD;
  const int q0 = qtile * 64 + wave * 16;
  const bool wave_active = q0 < N;

  // ---- stage the 64 x 64 Q tile through LDS (coalesced 16-B loads), then pull this lane's 16 operands
  {
    float* Qs = smem;  // [64][KSTR], aliases the K buffers before the main loop
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4;
      const int q = min(qtile * 64 + row, N - 1);   // clamped, not branched around: rows >= N are never stored
      const float4 v = *reinterpret_cast<const float4*>(base + (long long)q * D3 + c4);
      float2* d = reinterpret_cast<float2*>(Qs + row * KSTR + c4);
      d[0] = make_float2(v.x, v.y);
      d[1] = make_float2(v.z, v.w);
    }
  }
  __syncthreads();
  float qreg[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) qreg[s] = smem[(wave * 16 + qi) * KSTR + 4 * s + g] * scale;
  __syncthreads();

  float4 st[2];
  auto gload = [&](int chunk, int which) {  // which: 1 = K, 2 = V
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4;
      const int key = min(chunk * KCH + row, N - 1);   // keys >= N: clamped here, masked to -inf before the softmax
      st[i] = *reinterpret_cast<const float4*>(base + (long long)key * D3 + which * H * HD + c4);
    }
  };
  auto swrite_k = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4;
      float2* d = reinterpret_cast<float2*>(Ks + (buf * KCH + row) * KSTR + c4);
      d[0] = make_float2(st[i].x, st[i].y);
      d[1] = make_float2(st[i].z, st[i].w);
    }
  };
  auto swrite_v = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4;
      *reinterpret_cast<float4*>(Vs + (buf * KCH + row) * VSTR + c4) = st[i];
    }
  };

  // ---- phase 1: S^T = K Q^T, all key tiles kept in registers
  f32x4 sacc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) sacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  gload(0, 1);
  swrite_k(0);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int buf = c & 1;
    if (c + 1 < NC) gload(c + 1, 1);
    if (wave_active) {
      const float* kp = Ks + (buf * KCH + qi) * KSTR + g;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        // two independent accumulator chains (16x16x4 f32: 32-cycle issue, 40-cycle dependent latency)
        sacc[2 * c] = __builtin_amdgcn_mfma_f32_16x16x4f32(kp[4 * s], qreg[s], sacc[2 * c], 0, 0, 0);
        if (2 * c + 1 < NT)
          sacc[2 * c + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kp[16 * KSTR + 4 * s], qreg[s], sacc[2 * c + 1], 0, 0, 0);
      }
    }
    if (c + 1 < NC) swrite_k(buf ^ 1);
    __syncthreads();
  }

  // first V chunk's loads fly under the softmax
  gload(0, 2);

  // ---- softmax over the key axis (registers + 2 cross-lane steps)
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int key = 16 * t + 4 * g + e;
      if (key >= N) sacc[t][e] = -INFINITY;
      mx = fmaxf(mx, sacc[t][e]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float p = fast_exp(sacc[t][e] - mx);
      sacc[t][e] = p;
      sum += p;
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  const int q = q0 + qi;
  if (wave_active && q < N) {
    if (lse && g == 0) lse[((long long)f * H + h) * N + q] = mx + logf(sum);
    if (probs) {
      float* pr = probs + (((long long)f * H + h) * N + q) * N;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int key = 16 * t + 4 * g + e;
          if (key < N) pr[key] = sacc[t][e] * inv;
        }
    }
  }

  // ---- phase 2: O^T = V^T P^T
  f32x4 oacc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) oacc[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  swrite_v(0);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int buf = c & 1;
    if (c + 1 < NC) gload(c + 1, 2);
    if (wave_active) {
#pragma unroll
      for (int tt2 = 0; tt2 < 2; ++tt2) {
        const int t = 2 * c + tt2;
        if (t < NT) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float* vp = Vs + (buf * KCH + 16 * tt2 + 4 * g + e) * VSTR + qi;
            const float pb = sacc[t][e];
#pragma unroll
            for (int d = 0; d < 4; ++d) oacc[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[16 * d], pb, oacc[d], 0, 0, 0);
          }
        }
      }
    }
    if (c + 1 < NC) swrite_v(buf ^ 1);
    __syncthreads();
  }

  if (wave_active && q < N) {
    float* o = out + ((long long)f * N + q) * (H * HD) + h * HD + 4 * g;
#pragma unroll
    for (int d = 0; d < 4; ++d)
      *reinterpret_cast<float4*>(o + 16 * d) = make_float4(oacc[d][0] * inv, oacc[d][1] * inv, oacc[d][2] * inv, oacc[d][3] * inv);
  }
}

// KV-tiled variant for token grids that do not fit a register-resident score row (ViT-S/8: N = 785).  Same wave layout
// (16 queries per wave, S^T = K Q^T, P^T straight into O^T = V^T P^T), but keys arrive in 32-key chunks and the softmax is
// the online one: running max m and sum l per query, O rescaled by exp(m_old - m_new) when the max moves.  K and V of a
// chunk are staged together, double-buffered.  No probability output (only the register-resident kernel serves
// get_last_selfattention).
__global__ __launch_bounds__(256) void attention_fwd_flash_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                  float* __restrict__ lse, int N, int H, int FH, float scale) {
  __shared__ __attribute__((aligned(16))) float smem[2 * KCH * KSTR + 2 * KCH * VSTR];
  float* Ks = smem;
  float* Vs = smem + 2 * KCH * KSTR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = lane & 15, g = lane >> 4;
  int fh, qtile;
  if (!xcd_group_decode(blockIdx.x, (N + 63) / 64, FH, fh, qtile)) return;  // q-tiles of one (frame, head) share an XCD
  const int f = fh / H, h = fh - f * H;
  const int D3 = 3 * H * HD;
  const float* base = qkv + (long long)f * N * D3 + h * HD;
  const int q0 = qtile * 64 + wave * 16;
  const bool wave_active = q0 < N;
  {
    float* Qs = smem;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4;
      const int q = min(qtile * 64 + row, N - 1);   // clamped, not branched around: rows >= N are never stored
      const float4 v = *reinterpret_cast<const float4*>(base + (long long)q * D3 + c4);
      float2* d = reinterpret_cast<float2*>(Qs + row * KSTR + c4);
      d[0] = make_float2(v.x, v.y);
      d[1] = make_float2(v.z, v.w);
    }
  }
  __syncthreads();
  float qreg[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) qreg[s] = smem[(wave * 16 + qi) * KSTR + 4 * s + g] * scale;
  __syncthreads();

  float4 stk[2], stv[2];
  auto gload = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4;
      const int key = min(chunk * KCH + row, N - 1);   // keys >= N: clamped here, masked to -inf before the softmax
      stk[i] = *reinterpret_cast<const float4*>(base + (long long)key * D3 + H * HD + c4);
      stv[i] = *reinterpret_cast<const float4*>(base + (long long)key * D3 + 2 * H * HD + c4);
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4;
      float2* d = reinterpret_cast<float2*>(Ks + (buf * KCH + row) * KSTR + c4);
      d[0] = make_float2(stk[i].x, stk[i].y);
      d[1] = make_float2(stk[i].z, stk[i].w);
      *reinterpret_cast<float4*>(Vs + (buf * KCH + row) * VSTR + c4) = stv[i];
    }
  };

  f32x4 oacc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) oacc[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const int nchunks = (N + KCH - 1) / KCH;
  gload(0);
  swrite(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunks) gload(c + 1);
    if (wave_active) {
      f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0;
      const float* kp = Ks + (buf * KCH + qi) * KSTR + g;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kp[4 * s], qreg[s], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kp[16 * KSTR + 4 * s], qreg[s], s1, 0, 0, 0);
      }
      float mx = -INFINITY;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (c * KCH + 4 * g + e >= N) s0[e] = -INFINITY;
        if (c * KCH + 16 + 4 * g + e >= N) s1[e] = -INFINITY;
        mx = fmaxf(mx, fmaxf(s0[e], s1[e]));
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);          // finite: chunk 0 always holds key 0
      const float alpha = fast_exp(m_run - m_new);   // exp(-inf) = 0 on the first chunk
      float ps = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s0[e] = fast_exp(s0[e] - m_new);
        s1[e] = fast_exp(s1[e] - m_new);
        ps += s0[e] + s1[e];
      }
      ps += __shfl_xor(ps, 16, 64);
      ps += __shfl_xor(ps, 32, 64);
      l_run = l_run * alpha + ps;
      m_run = m_new;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int e = 0; e < 4; ++e) oacc[d][e] *= alpha;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* v0 = Vs + (buf * KCH + 4 * g + e) * VSTR + qi;
        const float* v1 = Vs + (buf * KCH + 16 + 4 * g + e) * VSTR + qi;
#pragma unroll
        for (int d = 0; d < 4; ++d) oacc[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0[16 * d], s0[e], oacc[d], 0, 0, 0);
#pragma unroll
        for (int d = 0; d < 4; ++d) oacc[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1[16 * d], s1[e], oacc[d], 0, 0, 0);
      }
    }
    if (c + 1 < nchunks) swrite(buf ^ 1);
    __syncthreads();
  }
  const int q = q0 + qi;
  if (wave_active && q < N) {
    const float inv = 1.0f / l_run;
    if (lse && g == 0) lse[((long long)f * H + h) * N + q] = m_run + logf(l_run);
    float* o = out + ((long long)f * N + q) * (H * HD) + h * HD + 4 * g;
#pragma unroll
    for (int d = 0; d < 4; ++d)
      *reinterpret_cast<float4*>(o + 16 * d) = make_float4(oacc[d][0] * inv, oacc[d][1] * inv, oacc[d][2] * inv, oacc[d][3] * inv);
  }
}

int launch_attention_fwd_q2(const float* qkv, float* out, float* lse, int F, int N, int H, float scale, hipStream_t s);  // attention_q2.hip

// Attention probabilities for sequences the register-resident kernel does not cover (N > 256, e.g. ViT-S/8's 785 tokens or
// 256x256 inputs at patch 16): only callers that ask for the reference's attn[F,h,N,N] pay for it (FeatureExtractor with
// return_attention=True; the training step never does).  One wave per query row, lane j handles keys j, j+64, ...: the
// logits go straight into the output row, are reduced to max / sum across the wave and normalised in place.
__global__ __launch_bounds__(256) void attention_probs_rows_kernel(const float* __restrict__ qkv, float* __restrict__ probs, int N, int H,
                                                                  float scale) {
  __shared__ float qs[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave, fh = blockIdx.y, f = fh / H, h = fh - f * H;
  if (row >= N) return;
  const int D3 = 3 * H * 64;
  const float* base = qkv + (size_t)f * N * D3;
  qs[wave][lane] = base[(size_t)row * D3 + h * 64 + lane];
  __builtin_amdgcn_wave_barrier();
  float* prow = probs + ((size_t)fh * N + row) * N;
  float mx = -INFINITY;
  for (int j = lane; j < N; j += 64) {
    const float4* kr = reinterpret_cast<const float4*>(base + (size_t)j * D3 + H * 64 + h * 64);
    float sdot = 0.f;
#pragma unroll
    for (int d = 0; d < 16; ++d) {
      const float4 kv = kr[d];
      sdot += qs[wave][4 * d] * kv.x + qs[wave][4 * d + 1] * kv.y + qs[wave][4 * d + 2] * kv.z + qs[wave][4 * d + 3] * kv.w;
    }
    sdot *= scale;
    prow[j] = sdot;
    mx = fmaxf(mx, sdot);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < N; j += 64) {
    const float e = expf(prow[j] - mx);
    prow[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  for (int j = lane; j < N; j += 64) prow[j] = prow[j] / sum;
}

template <int NT>
static int launch_fwd(const float* qkv, float* out, float* lse, float* probs, int F, int N, int H, float scale, hipStream_t s) {
  // 1-D over (frame*head, q-tile), ordered so that the q-tiles sharing K/V sit on one XCD (common.hpp xcd_group_decode)
  dim3 grid(xcd_group_grid(F * H, (N + 63) / 64));
  hipLaunchKernelGGL((attention_fwd_kernel<NT>), grid, dim3(256), 0, s, qkv, out, lse, probs, N, H, F * H, scale);
  TT_CHECK_LAUNCH("attention_fwd");
  return TT_OK;
}

}  // namespace tt

extern "C" int tt_attention_fwd(const float* qkv, float* out, float* lse, float* probs, int F, int N, int H, int hd,
                                float scale, tt_stream_t stream) {
  using namespace tt;
  TT_REQUIRE(qkv && out, "attention_fwd: null pointer");
  TT_REQUIRE(hd == 64, "attention_fwd: head_dim must be 64 (got %d)", hd);
  TT_REQUIRE(F > 0 && H > 0 && N > 0, "attention_fwd: bad shape");
  TT_REQUIRE(aligned16(qkv) && aligned16(out), "attention_fwd: buffers must be 16-byte aligned");
  hipStream_t s = as_stream(stream);
  if (N > 256) {  // KV-tiled online-softmax kernel (ViT-S/8: 785 tokens)
    hipLaunchKernelGGL(attention_fwd_flash_kernel, dim3(xcd_group_grid(F * H, (N + 63) / 64)), dim3(256), 0, s, qkv, out, lse, N, H,
                       F * H, scale);
    TT_CHECK_LAUNCH("attention_fwd_flash");
    if (probs) {
      hipLaunchKernelGGL(attention_probs_rows_kernel, dim3((N + 3) / 4, F * H), dim3(256), 0, s, qkv, probs, N, H, scale);
      TT_CHECK_LAUNCH("attention_probs_rows");
    }
    return TT_OK;
  }
  const int nt = (N + 15) / 16;
  if (probs == nullptr && N > 64) return launch_attention_fwd_q2(qkv, out, lse, F, N, H, scale, s);  // two q-tiles per wave
  if (nt <= 4) return launch_fwd<4>(qkv, out, lse, probs, F, N, H, scale, s);
  if (nt <= 8) return launch_fwd<8>(qkv, out, lse, probs, F, N, H, scale, s);
  if (nt <= 13) return launch_fwd<13>(qkv, out, lse, probs, F, N, H, scale, s);
  return launch_fwd<16>(qkv, out, lse, probs, F, N, H, scale, s);
}
