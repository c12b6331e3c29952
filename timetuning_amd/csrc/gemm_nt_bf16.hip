// bf16-MFMA instances of the forward nn.Linear GEMM (y = act(x @ w^T + b) (+ residual)), opt-in via tt_linear_fwd's `precision` argument (ABI 8):
//
//   mode 2  "bf16"    operands rounded to bf16 (RNE), f32 accumulate: the "MFMA bf16 path" of BASELINE config C4.  Does NOT
//                     meet the 1e-3 fp32 contract of the default path (bf16 has 8 significant bits) and is reported as such.
//   mode 1  "bf16x3"  split precision: x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 significant bits together);
//                     x*y ~= hi*hi' + hi*lo' + lo*hi' as three bf16 MFMAs into the same f32 accumulator (the dropped lo*lo'
//                     term is 2^-16 relative).  Products carry ~2^-16 relative error instead of f32's 2^-24.
//
// Inputs and outputs stay fp32 in HBM; the conversion happens while a slab is staged into LDS, so nothing else in the
// pipeline changes.  v_mfma_f32_32x32x16_bf16: lane l (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j] and
// B[k = 8h + j][col r], j = 0..7 - eight CONSECUTIVE k of a k-contiguous source, so the LDS image is [row][k] (row stride
// 40 bf16 = 80 B: conflict-free ds_read_b128) and a fragment is one ds_read_b128; no transpose anywhere.
// Tile 64WM x 64WN, BK = 32, 4 waves (2 x 2) of (32WM x 32WN); one LDS buffer + register prefetch of the next slab
// (global loads fly under the MFMAs; two barriers per slab).  Epilogue as gemm_nt_fast.hip (through LDS, 16-byte stores).
#include "common.hpp"

namespace tt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct Bf16Args {
  const float* A;  // [M][K]
  const float* B;  // [N][K]
  float* C;        // [M][N]
  int M, N, K;
  const float* bias;
  const float* residual;
  float* pre_out;
  int act;
};

__device__ __forceinline__ void split4(const float4 v, bf16x4& hi, bf16x4& lo, bool want_lo) {
  const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
  const bf16x2 ha = __builtin_convertvector(a, bf16x2), hb = __builtin_convertvector(b, bf16x2);  // v_cvt_pk_bf16_f32 (RNE)
  hi = (bf16x4){ha[0], ha[1], hb[0], hb[1]};
  if (want_lo) {
    const f32x2 ra = a - __builtin_convertvector(ha, f32x2), rb = b - __builtin_convertvector(hb, f32x2);  // exact in f32
    const bf16x2 la = __builtin_convertvector(ra, bf16x2), lb = __builtin_convertvector(rb, bf16x2);
    lo = (bf16x4){la[0], la[1], lb[0], lb[1]};
  }
}

template <int WM, int WN, int NPASS>
__global__ __launch_bounds__(256) void gemm_nt_bf16_kernel(Bf16Args g) {
  constexpr int BM = 64 * WM, BN = 64 * WN, BK = 32;
  constexpr int RS = 40;                       // LDS row stride in bf16 elements (80 B)
  constexpr int PL = (NPASS == 3) ? 2 : 1;     // planes per operand: hi (+ lo)
  constexpr int A_EL = BM * RS, B_EL = BN * RS;
  constexpr int LDS_BYTES_PIPE = (A_EL + B_EL) * PL * 2;
  constexpr int CH = 32 * WM, LDCS = BN + 4;
  constexpr int LDS_BYTES_EPI = CH * LDCS * 4;
  constexpr int LDS_BYTES = LDS_BYTES_PIPE > LDS_BYTES_EPI ? LDS_BYTES_PIPE : LDS_BYTES_EPI;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
  __bf16* Ah = reinterpret_cast<__bf16*>(smem);
  __bf16* Al = Ah + A_EL;                       // only when PL == 2
  __bf16* Bh = Ah + A_EL * PL;
  __bf16* Bl = Bh + B_EL;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int ntn = g.N / BN, ntm = (g.M + BM - 1) / BM;   // partial last row tile: loads clamp, stores mask
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  const int K = g.K;

  // staging: 8 float4 per 32-k row; thread owns row (tid >> 3) + 32 i, k-chunk (tid & 7) * 4
  constexpr int NA = BM / 32, NB = BN / 32;
  const int srow = tid >> 3, skc = (tid & 7) * 4;
  const float* pa[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = m0 + srow + 32 * i;
    pa[i] = g.A + (size_t)(row < g.M ? row : g.M - 1) * K + skc;
  }
  const float* pb = g.B + (size_t)(n0 + srow) * K + skc;
  const size_t step32 = (size_t)32 * K;
  float4 ra[NA], rb[NB];
  auto gload = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      ra[i] = *reinterpret_cast<const float4*>(pa[i]);
      pa[i] += BK;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const float4*>(pb + i * step32);
    pb += BK;
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      bf16x4 hi, lo;
      split4(ra[i], hi, lo, NPASS == 3);
      *reinterpret_cast<bf16x4*>(Ah + (srow + 32 * i) * RS + skc) = hi;
      if (NPASS == 3) *reinterpret_cast<bf16x4*>(Al + (srow + 32 * i) * RS + skc) = lo;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      bf16x4 hi, lo;
      split4(rb[i], hi, lo, NPASS == 3);
      *reinterpret_cast<bf16x4*>(Bh + (srow + 32 * i) * RS + skc) = hi;
      if (NPASS == 3) *reinterpret_cast<bf16x4*>(Bl + (srow + 32 * i) * RS + skc) = lo;
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = K / BK;
  gload();
  sstore();
  __syncthreads();
  const int fa_off = (wm * (32 * WM) + r) * RS + 8 * h;
  const int fb_off = (wn * (32 * WN) + r) * RS + 8 * h;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload();
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(Ah + fa_off + i * 32 * RS + 16 * ks);
        if (NPASS == 3) al[i] = *reinterpret_cast<const bf16x8*>(Al + fa_off + i * 32 * RS + 16 * ks);
      }
#pragma unroll
      for (int n = 0; n < WN; ++n) {
        bh[n] = *reinterpret_cast<const bf16x8*>(Bh + fb_off + n * 32 * RS + 16 * ks);
        if (NPASS == 3) bl[n] = *reinterpret_cast<const bf16x8*>(Bl + fb_off + n * 32 * RS + 16 * ks);
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int n = 0; n < WN; ++n) {
          if (NPASS == 3) {  // small terms first
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[n], acc[i][n], 0, 0, 0);
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[n], acc[i][n], 0, 0, 0);
          }
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[n], acc[i][n], 0, 0, 0);
        }
    }
    __syncthreads();            // every wave is done reading this slab
    if (kt + 1 < nk) sstore();  // convert + write the prefetched slab
    __syncthreads();
  }

  // ---- epilogue through LDS (same C/D layout as the f32 32x32 MFMA)
  float* Cs = reinterpret_cast<float*>(smem);
  constexpr int TPR = BN / 4, RPP = 256 / TPR;
  const int c4 = (tid % TPR) * 4;
  const int n = n0 + c4;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g.bias) bias4 = *reinterpret_cast<const float4*>(g.bias + n);
#pragma unroll
  for (int wmi = 0; wmi < 2; ++wmi) {
    if (wm == wmi) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            Cs[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * LDCS + wn * (32 * WN) + j * 32 + r] = acc[i][j][e];
    }
    __syncthreads();
    for (int rr = tid / TPR; rr < CH; rr += RPP) {
      if (m0 + wmi * CH + rr >= g.M) break;
      const size_t off = (size_t)(m0 + wmi * CH + rr) * g.N + n;
      float4 v = *reinterpret_cast<const float4*>(Cs + rr * LDCS + c4);
      v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
      if (g.pre_out) *reinterpret_cast<float4*>(g.pre_out + off) = v;
      if (g.act == 1) {
        v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w);
      }
      if (g.residual) {
        const float4 rs = *reinterpret_cast<const float4*>(g.residual + off);
        v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w;
      }
      *reinterpret_cast<float4*>(g.C + off) = v;
    }
    __syncthreads();
  }
}

template <int WM, int WN>
static int launch_bf16(const Bf16Args& g, int npass, hipStream_t s) {
  const int tiles = ((g.M + 64 * WM - 1) / (64 * WM)) * (g.N / (64 * WN));
  if (npass == 3) hipLaunchKernelGGL((gemm_nt_bf16_kernel<WM, WN, 3>), dim3(tiles), dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_nt_bf16_kernel<WM, WN, 1>), dim3(tiles), dim3(256), 0, s, g);
  TT_CHECK_LAUNCH("gemm_nt_bf16");
  return TT_OK;
}

int gemm_tile_choice(int M, int N, int batch);

// Returns TT_OK if launched, 1 if the shape is not eligible (caller falls back to the f32 kernels).
int try_launch_gemm_nt_bf16(const float* A, const float* B, float* C, int M, int N, int K, const float* bias, const float* residual,
                            float* pre_out, int act, int npass, hipStream_t s) {
  auto ok16 = [](const void* p) { return p == nullptr || aligned16(p); };
  if (K % 32 != 0 || K < 32 || !aligned16(A) || !aligned16(B) || !aligned16(C) || !ok16(bias) || !ok16(residual) || !ok16(pre_out)) return 1;
  Bf16Args g{A, B, C, M, N, K, bias, residual, pre_out, act};
  const int tile = gemm_tile_choice(M, N, 1);
  const int bn = (tile == 0 || tile == 1) ? 128 : 64;
  if (N % bn != 0) {   // (any M)
    if (N % 64 == 0) return launch_bf16<1, 1>(g, npass, s);
    return 1;
  }
  switch (tile) {
    case 0: return launch_bf16<2, 2>(g, npass, s);
    case 1: return launch_bf16<1, 2>(g, npass, s);
    case 2: return launch_bf16<2, 1>(g, npass, s);
    default: return launch_bf16<1, 1>(g, npass, s);
  }
}

}  // namespace tt
