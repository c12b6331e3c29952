// Lean instances of the f32-MFMA GEMM for the backward products (see gemm_nt_fast.hip for the scheme):
//   dgrad  dx[M,K] = dy[M,N] @ w[N,K]  (* gelu'(pre))   A = dy k-contiguous, B = w n-contiguous; rows guarded (the target-frame
//                                                      slice has M = bs * 197 rows, not a multiple of 64), N % 16 == 0
//   wgrad  dw[N,K] = dy[M,N]^T @ x[M,K]                 both operands m-major ([reduction][extent]); split-K over M with
//                                                      partial tiles to a workspace (folded by splitk_reduce_kernel)
// Output extents must fill whole tiles along the unguarded axes; everything else falls back to the general kernel.
#include "common.hpp"

namespace tt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct BwdArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;           // output [M][N], reduction K
  int lda, ldb;          // leading dimensions of the stored operands
  const float* gelu_pre; // dgrad only: [M][N]
  int kchunk;            // wgrad: reduction slice per grid.z
  long long strideS;     // wgrad: elements between split-K partial outputs
  float* colpart;        // wgrad, optional: [grid.z][M] partial column sums of A (= the bias gradient of the same nn.Linear)
};

// Shared main loop.  A_MMAJOR: A stored [K][lda] (else [M][lda], k contiguous).  B is always stored [K][ldb].
// COLSUM (wgrad): the workgroup also sums the A rows it stages (A = dy [reduction][M], so these are column sums of dy over
// the slice) into csum - the bias gradient rides on loads the weight gradient needs anyway, instead of a second pass over dy.
template <int WM, int WN, bool A_MMAJOR, bool GUARD_M, bool COLSUM = false>
__device__ __forceinline__ void bwd_mainloop(const BwdArgs& g, float* lds, int m0, int n0, int kbeg, int kend, f32x16 (&acc)[WM][WN],
                                             float4* csum = nullptr, int cs_mod = 1, int cs_sel = 0) {
  constexpr int BM = 64 * WM, BN = 64 * WN, BK = 16;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int ASZ = BK * LDA, BSZ = BK * LDB;
  constexpr int NA = WM, NB = WN;  // float4 per thread per slab
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  float4 ra[NA], rb[NB];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int u = tid + 256 * i;
      if (A_MMAJOR) {
        const int kr = u / (BM / 4), mc = (u % (BM / 4)) * 4;
        ra[i] = *reinterpret_cast<const float4*>(g.A + (size_t)(k0 + kr) * g.lda + m0 + mc);
      } else {
        const int row = u >> 2, kc = (u & 3) * 4;
        int m = m0 + row;
        if (GUARD_M) m = min(m, g.M - 1);  // clamped rows are computed but never stored
        ra[i] = *reinterpret_cast<const float4*>(g.A + (size_t)m * g.lda + k0 + kc);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int u = tid + 256 * i;
      const int kr = u / (BN / 4), nc = (u % (BN / 4)) * 4;
      rb[i] = *reinterpret_cast<const float4*>(g.B + (size_t)(k0 + kr) * g.ldb + n0 + nc);
    }
  };
  int cs_phase = 0;  // COLSUM: the column tiles of one row block share the sums - slab s belongs to tile s % cs_mod
  auto sstore = [&](int buf) {
    const bool cs_now = COLSUM && csum && (cs_phase % cs_mod) == cs_sel;
    ++cs_phase;
    float* da = lds + buf * ASZ;
    float* db = lds + 2 * ASZ + buf * BSZ;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int u = tid + 256 * i;
      if (A_MMAJOR) {
        const int kr = u / (BM / 4), mc = (u % (BM / 4)) * 4;
        *reinterpret_cast<float4*>(da + kr * LDA + mc) = ra[i];
        if (cs_now) {
          csum->x += ra[i].x; csum->y += ra[i].y; csum->z += ra[i].z; csum->w += ra[i].w;
        }
      } else {
        const int row = u >> 2, kc = (u & 3) * 4;
        da[(kc + 0) * LDA + row] = ra[i].x;
        da[(kc + 1) * LDA + row] = ra[i].y;
        da[(kc + 2) * LDA + row] = ra[i].z;
        da[(kc + 3) * LDA + row] = ra[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int u = tid + 256 * i;
      const int kr = u / (BN / 4), nc = (u % (BN / 4)) * 4;
      *reinterpret_cast<float4*>(db + kr * LDB + nc) = rb[i];
    }
  };
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int nk = (kend - kbeg) / BK;
  if (nk <= 0) return;  // empty split-K slice: the tile is all zeros (uniform across the workgroup)
  gload(kbeg);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kbeg + (kt + 1) * BK);
    const float* fa = lds + buf * ASZ + (4 * h) * LDA + wm * (32 * WM) + r;
    const float* fb = lds + 2 * ASZ + buf * BSZ + (4 * h) * LDB + wn * (32 * WN) + r;
#pragma unroll
    for (int j = 0; j < BK / 8; ++j) {
      float a[WM][4], b[WN][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int i = 0; i < WM; ++i) a[i][q] = fa[(8 * j + q) * LDA + i * 32];
#pragma unroll
        for (int n = 0; n < WN; ++n) b[n][q] = fb[(8 * j + q) * LDB + n * 32];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int n = 0; n < WN; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[n][q], acc[i][n], 0, 0, 0);
    }
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }
}

// Tile leaves through LDS as 16-byte row-major stores (see gemm_nt_fast.hip); optional gelu'(pre) factor and row guard.
template <int WM, int WN, bool GUARD_M>
__device__ __forceinline__ void bwd_epilogue(float* lds, const f32x16 (&acc)[WM][WN], float* __restrict__ C, int ldc, int m0, int n0, int M,
                                             const float* __restrict__ gelu_pre) {
  constexpr int BN = 64 * WN, CH = 32 * WM, LDCS = BN + 4, TPR = BN / 4, RPP = 256 / TPR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int c4 = (tid % TPR) * 4;
#pragma unroll
  for (int wmi = 0; wmi < 2; ++wmi) {
    if (wm == wmi) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            lds[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * LDCS + wn * (32 * WN) + j * 32 + r] = acc[i][j][e];
    }
    __syncthreads();
    for (int rr = tid / TPR; rr < CH; rr += RPP) {
      const int m = m0 + wmi * CH + rr;
      if (GUARD_M && m >= M) break;
      const size_t off = (size_t)m * ldc + n0 + c4;
      float4 v = *reinterpret_cast<const float4*>(lds + rr * LDCS + c4);
      if (gelu_pre) {
        const float4 gp = *reinterpret_cast<const float4*>(gelu_pre + off);
        v.x *= gelu_grad_f(gp.x); v.y *= gelu_grad_f(gp.y); v.z *= gelu_grad_f(gp.z); v.w *= gelu_grad_f(gp.w);
      }
      *reinterpret_cast<float4*>(C + off) = v;
    }
    __syncthreads();
  }
}

// The two tile bodies, shared by the stand-alone kernels and by the fused launch below.  `id` = linear workgroup index among the
// tiles of its kind, z = split-K slice (wgrad).
template <int WM, int WN>
__device__ __forceinline__ void dgrad_tile(const BwdArgs& g, float* lds, int id) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  const int ntn = g.N / BN, ntm = (g.M + BM - 1) / BM;
  const int tile = xcd_remap(id, ntm * ntn);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  f32x16 acc[WM][WN];
  bwd_mainloop<WM, WN, false, true>(g, lds, m0, n0, 0, g.K, acc);
  bwd_epilogue<WM, WN, true>(lds, acc, g.C, g.N, m0, n0, g.M, g.gelu_pre);
}

template <int WM, int WN>
__device__ __forceinline__ void wgrad_tile(const BwdArgs& g, float* lds, int id, int z) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  const int ntn = g.N / BN, ntm = g.M / BM;
  const int tile = xcd_remap(id, ntm * ntn);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  const int kbeg = z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
  f32x16 acc[WM][WN];
  // the ntn workgroups of a row block each sum the slabs s with s % ntn == their column-tile index: the bias gradient's
  // extra adds are spread evenly instead of making one workgroup per row block the straggler
  const bool do_cs = g.colpart != nullptr;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  bwd_mainloop<WM, WN, true, false, true>(g, lds, m0, n0, kbeg, kend, acc, do_cs ? &cs : nullptr, ntn, n0 / BN);
  if (do_cs) {
    // thread tid staged columns 4 (tid % (BM/4)) .. +3 of reduction rows tid / (BM/4) (+ 256 i / (BM/4)): fold the row groups
    constexpr int G = BM / 4, R = 256 / G;
    float* red = lds;  // the pipeline buffers are idle between the main loop and the epilogue
    const int tid = threadIdx.x;
    *reinterpret_cast<float4*>(red + (tid / G) * BM + (tid % G) * 4) = cs;
    __syncthreads();
    if (tid < BM) {
      float sum = 0.f;
#pragma unroll
      for (int rr = 0; rr < R; ++rr) sum += red[rr * BM + tid];
      g.colpart[((size_t)z * ntn + n0 / BN) * g.M + m0 + tid] = sum;
    }
    __syncthreads();
  }
  bwd_epilogue<WM, WN, false>(lds, acc, g.C + (long long)z * g.strideS, g.N, m0, n0, g.M, nullptr);
}

template <int WM, int WN>
__global__ __launch_bounds__(256) void gemm_dgrad_fast_kernel(BwdArgs g) {
  constexpr int BM = 64 * WM, BN = 64 * WN, BK = 16;
  __shared__ __attribute__((aligned(16))) float lds[2 * BK * (BM + 4 + BN + 4)];
  dgrad_tile<WM, WN>(g, lds, blockIdx.x);
}

template <int WM, int WN>
__global__ __launch_bounds__(256) void gemm_wgrad_fast_kernel(BwdArgs g) {
  constexpr int BM = 64 * WM, BN = 64 * WN, BK = 16;
  __shared__ __attribute__((aligned(16))) float lds[2 * BK * (BM + 4 + BN + 4)];
  wgrad_tile<WM, WN>(g, lds, blockIdx.x, blockIdx.z);
}

// BOTH backward products of an nn.Linear in one launch - they read the same dy and do not depend on each other: workgroups
// [0, nd) are the dgrad tiles (WM x WN), the rest the 64 x 64 split-K tiles of the weight gradient (slice-major).  Neither product
// fills the chip on the 32 target frames (594 dgrad tiles of width 384 = 2.3 per CU; ~1000 weight-gradient slices): as two kernels
// each ends in a tail of idle CUs (and two kernels on two streams did not fill each other's tails, round 2), one grid does.
template <int WM, int WN>
__global__ __launch_bounds__(256) void gemm_bwd_fused_kernel(BwdArgs gd, BwdArgs gw, int nd, int wtiles) {
  constexpr int BM = 64 * WM, BN = 64 * WN, BK = 16;   // (>= the 64 x 64 tile of the weight-gradient part)
  __shared__ __attribute__((aligned(16))) float lds[2 * BK * (BM + 4 + BN + 4)];
  const int b = blockIdx.x;
  if (b < nd) {
    dgrad_tile<WM, WN>(gd, lds, b);
  } else {
    const int w = b - nd;
    wgrad_tile<1, 1>(gw, lds, w % wtiles, w / wtiles);
  }
}

int gemm_tile_choice(int M, int N, int batch);

// dx[M,K] = dy[M,N] @ w[N,K]: output [M][K], reduction N.  Returns 1 when not eligible.
int try_launch_dgrad_fast(const float* dy, const float* w, const float* gelu_pre, float* dx, int M, int N, int K, hipStream_t s) {
  auto ok16 = [](const void* p) { return p == nullptr || aligned16(p); };
  if (N % 16 != 0 || N < 16 || K % 64 != 0 || !aligned16(dy) || !aligned16(w) || !aligned16(dx) || !ok16(gelu_pre)) return 1;
  BwdArgs g{dy, w, dx, M, K, N, N, K, gelu_pre, N, 0, nullptr};
  const int tile = gemm_tile_choice(M, K, 1);
  const bool bn128 = (tile == 0 || tile == 1) && K % 128 == 0;
  const bool bm128 = (tile == 0 || tile == 2);
  const int bm = bm128 ? 128 : 64, bn = bn128 ? 128 : 64;
  dim3 grid(((M + bm - 1) / bm) * (K / bn));
  if (bm128 && bn128) hipLaunchKernelGGL((gemm_dgrad_fast_kernel<2, 2>), grid, dim3(256), 0, s, g);
  else if (bm128) hipLaunchKernelGGL((gemm_dgrad_fast_kernel<2, 1>), grid, dim3(256), 0, s, g);
  else if (bn128) hipLaunchKernelGGL((gemm_dgrad_fast_kernel<1, 2>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_dgrad_fast_kernel<1, 1>), grid, dim3(256), 0, s, g);
  TT_CHECK_LAUNCH("gemm_dgrad_fast");
  return TT_OK;
}

// partial[z][N][K] = dy[zslice,N]^T @ x[zslice,K]: output [N][K], reduction M (split `splits` ways, kchunk % 16 == 0).
// colpart (optional): [splits * (K / tile columns)][N] partial column sums of dy, written by the same launch (the caller folds
// them into db); *colparts receives that leading extent.
int try_launch_wgrad_fast(const float* dy, const float* x, float* out, int M, int N, int K, int splits, int kchunk, float* colpart,
                          int* colparts, hipStream_t s) {
  if (M % 16 != 0 || kchunk % 16 != 0 || N % 64 != 0 || K % 64 != 0 || !aligned16(dy) || !aligned16(x) || !aligned16(out)) return 1;
  BwdArgs g{dy, x, out, N, K, M, N, K, nullptr, kchunk, (long long)N * K, colpart};
  const int tile = splits > 1 ? 3 : gemm_tile_choice(N, K, 1);  // split plans are made for 64x64 tiles (gemm_splitk_choice)
  const bool bm128 = (tile == 0 || tile == 2) && N % 128 == 0;
  const bool bn128 = (tile == 0 || tile == 1) && K % 128 == 0;
  const int bm = bm128 ? 128 : 64, bn = bn128 ? 128 : 64;
  dim3 grid((N / bm) * (K / bn), 1, splits);
  if (colparts) *colparts = splits * (K / bn);
  if (bm128 && bn128) hipLaunchKernelGGL((gemm_wgrad_fast_kernel<2, 2>), grid, dim3(256), 0, s, g);
  else if (bm128) hipLaunchKernelGGL((gemm_wgrad_fast_kernel<2, 1>), grid, dim3(256), 0, s, g);
  else if (bn128) hipLaunchKernelGGL((gemm_wgrad_fast_kernel<1, 2>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_wgrad_fast_kernel<1, 1>), grid, dim3(256), 0, s, g);
  TT_CHECK_LAUNCH("gemm_wgrad_fast");
  return TT_OK;
}

// dx = dy @ w (* gelu'(pre)) and the split-K partials of dw = dy^T @ x (+ the column partials of db) in ONE launch.  Returns 1 when
// either product is not eligible for its lean kernel (the caller then launches them separately).
int try_launch_bwd_fused(const float* dy, const float* w, const float* x, const float* gelu_pre, float* dx, float* wpart, int M, int N, int K,
                         int splits, int kchunk, float* colpart, int* colparts, hipStream_t s) {
  auto ok16 = [](const void* p) { return p == nullptr || aligned16(p); };
  // dgrad: dx[M,K] = dy[M,N] @ w[N,K]
  if (N % 16 != 0 || N < 16 || K % 64 != 0 || !aligned16(dy) || !aligned16(w) || !aligned16(dx) || !ok16(gelu_pre)) return 1;
  // wgrad: partial[z][N][K] = dy[zslice,N]^T @ x[zslice,K], 64 x 64 tiles
  if (splits < 2 || M % 16 != 0 || kchunk % 16 != 0 || N % 64 != 0 || !aligned16(x) || !aligned16(wpart)) return 1;
  BwdArgs gd{dy, w, dx, M, K, N, N, K, gelu_pre, N, 0, nullptr};
  BwdArgs gw{dy, x, wpart, N, K, M, N, K, nullptr, kchunk, (long long)N * K, colpart};
  const int tile = gemm_tile_choice(M, K, 1);
  const bool bn128 = (tile == 0 || tile == 1) && K % 128 == 0;
  const bool bm128 = (tile == 0 || tile == 2);
  const int bm = bm128 ? 128 : 64, bn = bn128 ? 128 : 64;
  const int nd = ((M + bm - 1) / bm) * (K / bn), wtiles = (N / 64) * (K / 64);
  if (colparts) *colparts = splits * (K / 64);
  dim3 grid(nd + wtiles * splits);
  if (bm128 && bn128) hipLaunchKernelGGL((gemm_bwd_fused_kernel<2, 2>), grid, dim3(256), 0, s, gd, gw, nd, wtiles);
  else if (bm128) hipLaunchKernelGGL((gemm_bwd_fused_kernel<2, 1>), grid, dim3(256), 0, s, gd, gw, nd, wtiles);
  else if (bn128) hipLaunchKernelGGL((gemm_bwd_fused_kernel<1, 2>), grid, dim3(256), 0, s, gd, gw, nd, wtiles);
  else hipLaunchKernelGGL((gemm_bwd_fused_kernel<1, 1>), grid, dim3(256), 0, s, gd, gw, nd, wtiles);
  TT_CHECK_LAUNCH("gemm_bwd_fused");
  return TT_OK;
}

}  // namespace tt
