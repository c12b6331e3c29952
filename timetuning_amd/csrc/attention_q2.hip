// Fused attention forward, two query tiles per wave (N <= 256, head_dim 64, f32 MFMA 16x16x4).
//
// Same scheme as attention.hip (S^T = K Q^T keeps a query's score row in one lane-column, P^T feeds O^T = V^T P^T
// from registers) but each wave owns 32 queries = two 16-query tiles that SHARE every K / V fragment read from LDS:
// one ds_read_b32 feeds two MFMAs, a workgroup (4 waves) covers 128 queries, and the work between two barriers
// doubles (64 MFMAs per wave per 32-key chunk), which is what the one-tile version was short of (44 % MFMA busy).
// The one-tile kernel stays for the probability output.
//
// Measured alternative, not shipped (round 2): a register-streaming form with NO LDS and NO barriers - every wave an independent
// stream that loads K / Q fragments straight from global memory as 16-byte loads (the contraction order over d permuted to
// d = 16 g + s so that a lane's sixteen values are consecutive) and the V^T operand as sixteen dwords per key tile, next tile in
// flight under the MFMAs (223 VGPRs, counted vmcnt throughout, bit-compatible results): 129 us per ViT-S/16 layer of 128 frames
// against 112 us for this kernel on the same box.  Sharing K / V through LDS (one fetch per workgroup instead of four) is
// worth more than the sixteen barriers cost.  A hybrid - this kernel with the d-permuted Q loads (no Q staging, no barriers
// for it) and a 256-byte-row K image XOR-swizzled for conflict-free ds_read_b128 fragments (4 reads per key tile instead of
// 16 ds_read_b32) - was correct to 1.4e-6 and 10 % SLOWER (118 vs 108 us, 128 bytes of scratch per lane): also not shipped.
// Rotating which 32-query slice a wave owns with the (frame, head) index, so that the slices beyond N (wave 3 of every second
// workgroup at N = 197: no MFMA work) do not always fall on the same SIMD: 108.1 vs 108.3 us - the SIMDs are not the imbalance.
// Round 3: the three workgroups a CU receives in the first dispatch wave started 0 / 1 / 2 x {4 k, 10 k, 20 k} cycles apart (so that
// their load / softmax / store phases do not coincide): 96.8 / 97.7 / 98.9 vs 97.4 us - lockstep is not it either.
#include "common.hpp"

namespace tt {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int Q2_HD = 64, Q2_KCH = 32, Q2_KSTR = 66, Q2_VSTR = 68;

// 3 resident workgroups per CU (168 VGPRs, 10 of them spilled) beat 2 (190 VGPRs, no spill) by 3 %: tools/ab_attn.py,
// 107.4 vs 111.0 us per ViT-S/16 layer of 128 frames - the load / softmax / store phases of a third workgroup fill MFMA gaps.
#ifndef TT_Q2_WAVES_PER_SIMD
#define TT_Q2_WAVES_PER_SIMD 3
#endif

template <int NT>
__global__ __launch_bounds__(256, TT_Q2_WAVES_PER_SIMD) void attention_fwd_q2_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                  float* __restrict__ lse, int N, int H, int FH, float scale) {
  constexpr int NC = (NT + 1) / 2;
  __shared__ __attribute__((aligned(16))) float smem[2 * Q2_KCH * Q2_KSTR + 2 * Q2_KCH * Q2_VSTR];
  float* Ks = smem;
  float* Vs = smem + 2 * Q2_KCH * Q2_KSTR;
  // (the wave index through readfirstlane: the compiler then KNOWS that q0 / wave_active are wave-uniform and branches on SCC
  // instead of saving and restoring EXEC around every guarded MFMA block)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 15, g = lane >> 4;
  int fh, qblk;
  if (!xcd_group_decode(blockIdx.x, (N + 127) / 128, FH, fh, qblk)) return;
  const int f = fh / H, h = fh - f * H;
  const int D3 = 3 * H * Q2_HD;
  const float* base = qkv + (long long)f * N * D3 + h * Q2_HD;
  const int q0 = qblk * 128 + wave * 32;
  const bool wave_active = q0 < N;

  // stage the 128 x 64 Q block through LDS (aliases the whole K/V staging area), pull 2 x 16 operands per lane
  {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4;
      // rows beyond N are CLAMPED, not branched around (queries >= N are never stored, keys >= N are masked to -inf below and
      // their V rows meet P = 0): no exec-mask juggling around the loads
      const int q = min(qblk * 128 + row, N - 1);
      const float4 v = *reinterpret_cast<const float4*>(base + (long long)q * D3 + c4);
      float2* d = reinterpret_cast<float2*>(smem + row * Q2_KSTR + c4);
      d[0] = make_float2(v.x, v.y);
      d[1] = make_float2(v.z, v.w);
    }
  }
  __syncthreads();
  // scores in LOG2 units (q pre-multiplied by scale log2(e)): the softmax is then a subtract and a bare v_exp_f32 per element
  const float scale2 = scale * 1.44269504088896340736f;
  float qreg[2][16];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s = 0; s < 16; ++s) qreg[t][s] = smem[(wave * 32 + t * 16 + qi) * Q2_KSTR + 4 * s + g] * scale2;
  __syncthreads();

  float4 st[2];
  auto gload = [&](int chunk, int which) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4;
      const int key = min(chunk * Q2_KCH + row, N - 1);
      st[i] = *reinterpret_cast<const float4*>(base + (long long)key * D3 + which * H * Q2_HD + c4);
    }
  };
  auto swrite_k = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4;
      float2* d = reinterpret_cast<float2*>(Ks + (buf * Q2_KCH + row) * Q2_KSTR + c4);
      d[0] = make_float2(st[i].x, st[i].y);
      d[1] = make_float2(st[i].z, st[i].w);
    }
  };
  auto swrite_v = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4;
      *reinterpret_cast<float4*>(Vs + (buf * Q2_KCH + row) * Q2_VSTR + c4) = st[i];
    }
  };

  f32x4 sacc[2][NT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int k = 0; k < NT; ++k) sacc[t][k] = (f32x4){0.f, 0.f, 0.f, 0.f};

  gload(0, 1);
  swrite_k(0);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int buf = c & 1;
    if (c + 1 < NC) gload(c + 1, 1);
    if (wave_active) {
      const float* kp = Ks + (buf * Q2_KCH + qi) * Q2_KSTR + g;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float k0 = kp[4 * s];
        sacc[0][2 * c] = __builtin_amdgcn_mfma_f32_16x16x4f32(k0, qreg[0][s], sacc[0][2 * c], 0, 0, 0);
        sacc[1][2 * c] = __builtin_amdgcn_mfma_f32_16x16x4f32(k0, qreg[1][s], sacc[1][2 * c], 0, 0, 0);
        if (2 * c + 1 < NT) {
          const float k1 = kp[16 * Q2_KSTR + 4 * s];
          sacc[0][2 * c + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(k1, qreg[0][s], sacc[0][2 * c + 1], 0, 0, 0);
          sacc[1][2 * c + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(k1, qreg[1][s], sacc[1][2 * c + 1], 0, 0, 0);
        }
      }
    }
    if (c + 1 < NC) swrite_k(buf ^ 1);
    __syncthreads();
  }
  gload(0, 2);

  float inv[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      if (16 * k + 15 >= N) {   // (wave-uniform) only the last key tiles can hold keys >= N
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (16 * k + 4 * g + e >= N) sacc[t][k][e] = -INFINITY;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) mx = fmaxf(mx, sacc[t][k][e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p = __builtin_amdgcn_exp2f(sacc[t][k][e] - mx);
        sacc[t][k][e] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    inv[t] = 1.0f / sum;
    const int q = q0 + 16 * t + qi;
    if (lse && g == 0 && q < N) lse[((long long)f * H + h) * N + q] = (mx + log2f(sum)) * 0.69314718055994530942f;   // natural-log units
  }

  f32x4 oacc[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int d = 0; d < 4; ++d) oacc[t][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  swrite_v(0);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int buf = c & 1;
    if (c + 1 < NC) gload(c + 1, 2);
    if (wave_active) {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const int k = 2 * c + t2;
        if (k < NT) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float* vp = Vs + (buf * Q2_KCH + 16 * t2 + 4 * g + e) * Q2_VSTR + qi;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              const float v = vp[16 * d];
              oacc[0][d] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, sacc[0][k][e], oacc[0][d], 0, 0, 0);
              oacc[1][d] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, sacc[1][k][e], oacc[1][d], 0, 0, 0);
            }
          }
        }
      }
    }
    if (c + 1 < NC) swrite_v(buf ^ 1);
    __syncthreads();
  }
  // The 32 x 64 output tile of a wave leaves through LDS (the K / V staging area is idle behind the last barrier; 2144 floats per wave)
  // so that a store instruction writes FOUR WHOLE 256-byte rows instead of sixteen 64-byte pieces: 16-byte chunk c of row r sits at
  // chunk position c ^ (r & 15).  (Round 3: the direct stores cost 4.8 of the kernel's 102 us, tools/ab_attn.py.)
#ifndef TT_Q2_NOSTORE   // (timing-study builds only compile the stores out)
  if (wave_active) {
    float* scr = smem + wave * 2144;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int d = 0; d < 4; ++d)
        *reinterpret_cast<float4*>(scr + (16 * t + qi) * 64 + (((4 * d + g) ^ qi) << 2)) =
            make_float4(oacc[t][d][0] * inv[t], oacc[t][d][1] * inv[t], oacc[t][d][2] * inv[t], oacc[t][d][3] * inv[t]);
    const int c16 = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = (lane >> 4) + 4 * i;
      const float4 v = *reinterpret_cast<const float4*>(scr + row * 64 + ((c16 ^ (row & 15)) << 2));
      if (q0 + row < N) *reinterpret_cast<float4*>(out + ((long long)f * N + q0 + row) * (H * Q2_HD) + h * Q2_HD + 4 * c16) = v;
    }
  }
#endif
}

template <int NT>
static int launch_q2(const float* qkv, float* out, float* lse, int F, int N, int H, float scale, hipStream_t s) {
  hipLaunchKernelGGL((attention_fwd_q2_kernel<NT>), dim3(xcd_group_grid(F * H, (N + 127) / 128)), dim3(256), 0, s, qkv, out, lse, N, H,
                     F * H, scale);
  TT_CHECK_LAUNCH("attention_fwd_q2");
  return TT_OK;
}

int launch_attention_fwd_q2(const float* qkv, float* out, float* lse, int F, int N, int H, float scale, hipStream_t s) {
  const int nt = (N + 15) / 16;
  if (nt <= 8) return launch_q2<8>(qkv, out, lse, F, N, H, scale, s);
  if (nt <= 13) return launch_q2<13>(qkv, out, lse, F, N, H, scale, s);
  return launch_q2<16>(qkv, out, lse, F, N, H, scale, s);
}

}  // namespace tt
