// fp32 GEMM on the CDNA4 matrix cores: v_mfma_f32_32x32x2_f32 (exact f32 in / f32 accumulate).
//
// Why f32 MFMA and not bf16: the north-star contract is <= 1e-3 relative fp32 on patch embeddings
// after 12 residual blocks, and gfx950 has no TF32/xf32 path; the f32 MFMA is bit-for-bit an fmaf
// chain at the f32 vector rate (157 TFLOP/s) while leaving the VALU free for epilogues.
//
// Tiling: 256 threads = 4 waves in a 2x2 grid; each wave owns WM x WN accumulator tiles of 32x32
// (block tile 64*WM x 64*WN).  K is consumed in slabs of BK through a double-buffered LDS image; the next slab's
// 16-byte global loads are issued before the current slab's MFMAs and written to the other buffer afterwards
// (one barrier per slab).
//
// Operand staging.  The MFMA wants, per lane l, A[row = l & 31][k = l >> 5] and B[k = l >> 5][col = l & 31]; WHICH two k
// values the two half-waves supply is free as long as A and B agree, so MFMA number i of k-group j uses
// k = 8 j + 4 (l >> 5) + i.  The LDS image is k-major ([k][m], stride BM + 4) for every operand and a fragment is one
// ds_read_b32 per MFMA operand: an m/n-contiguous source (dgrad / wgrad operands) is copied with 16-byte LDS writes, a
// k-contiguous source (activations [M][K], nn.Linear weights [N][K]) is transposed on the way in (four scalar writes per
// float4).  The alternative [row][k] image with ds_read_b128 fragments (TT_KLAYOUT) measured slower and is off.
// This is the GENERAL kernel (any extents, alignments and layouts); whole-tile products take the lean instances in
// gemm_nt_fast.hip / gemm_bwd_fast.hip / gemm_nt_bf16.hip.
//
// Epilogue (fused): alpha, bias, pre-activation store, exact GELU, row scale,
// GELU-derivative multiply (dgrad through fc1 / head activations), residual add, and for the patch-embed instance the
// row remap to token order plus the pos-embed add.  Split-K (grid.z) serves the weight-gradient products, whose
// outputs are too small to fill 256 CUs: partial tiles go to a workspace and a second kernel folds them in a fixed
// order (deterministic, no atomics).
#include "common.hpp"
#include <cstdlib>

namespace tt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  int lda, ldb, ldc;
  float alpha;
  const float* bias;       // [N]
  const float* residual;   // [M][ldc]
  float* pre_out;          // [M][ldc]
  const float* gelu_pre;   // [M][ldc]: C *= gelu'(gelu_pre)
  const float* row_scale;  // [M]
  int act;                 // 1 = GELU
  long long strideA, strideB, strideC;  // batch strides (grid.y)
  int batch_inner;                      // two-level batch: grid.y = inner + batch_inner * outer (0: one level)
  long long strideA2, strideB2, strideC2;  // strides of the outer level
  int splits, kchunk;                   // split-K over grid.z: slice z covers k in [z*kchunk, (z+1)*kchunk)
  long long strideS;                    // element stride between split-K partial outputs
  int vecA, vecB;                       // 16-byte loads legal for A / B (alignment, leading dimension and extents % 4)
  // patch-embed (AMODE == 2)
  const int* frame_map;
  int Cin, H, W, P, gw, n_patch;
  const float* pos;  // [(n_patch+1)][N]
  int bf16;          // NT products on bf16 MFMA (operands rounded as they leave LDS); set by launch_gemm_plain2 in the "bf16" mode
};


// BF16 (the "bf16" precision mode's batched products - the label propagation's cosine similarities): every group of four k = 2 f32 MFMAs
// becomes ONE v_mfma_f32_32x32x8_bf16 on operands rounded to bf16 as they leave LDS (element q of a lane's operand is the fp32 loop's
// register q: any assignment of the contraction index to (lane half, element) is valid as long as both operands use it).
typedef __bf16 gemm_bf16x4 __attribute__((ext_vector_type(4)));
typedef short gemm_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ gemm_s16x4 gemm_pack_bf16(const float (&v)[4]) {
  const gemm_bf16x4 p = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  return __builtin_bit_cast(gemm_s16x4, p);
}

template <int WM, int WN, int AMODE, int BMODE, int BK, bool BF16 = false>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  // TT_KLAYOUT selects the [row][k] image (b128 reads) for k-contiguous sources.  Measured on MI355X (tools/ab_gemm.py):
  // it needs 41 KB of LDS per workgroup against 33.8 KB for the k-major image and loses one resident workgroup per
  // CU, which costs more (qkv 105 -> 90 TFLOP/s) than the 4x fewer LDS instructions buy; the default is off.
#ifdef TT_KLAYOUT
  constexpr bool AK = (AMODE != 1), BKL = (BMODE == 0);
#else
  constexpr bool AK = false, BKL = false;
#endif
  constexpr int LDA = AK ? BK + 4 : BM + 4, LDB = BKL ? BK + 4 : BN + 4;
  constexpr int ASZ = AK ? BM * LDA : BK * LDA, BSZ = BKL ? BN * LDB : BK * LDB;
  constexpr int NA = BM * BK / 1024, NB = BN * BK / 1024;  // float4 per thread per slab
  __shared__ __attribute__((aligned(16))) float lds[2 * (ASZ + BSZ)];
  float* As = lds;
  float* Bs = lds + 2 * ASZ;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  const int ntn = (g.N + BN - 1) / BN;
  const int ntm = (g.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  const int by_outer = g.batch_inner ? (int)blockIdx.y / g.batch_inner : 0;
  const int by = g.batch_inner ? (int)blockIdx.y - by_outer * g.batch_inner : (int)blockIdx.y;
  const float* __restrict__ A = g.A + (long long)by * g.strideA + (long long)by_outer * g.strideA2;
  const float* __restrict__ B = g.B + (long long)by * g.strideB + (long long)by_outer * g.strideB2;
  float* __restrict__ C = g.C + (long long)by * g.strideC + (long long)by_outer * g.strideC2 + (long long)blockIdx.z * g.strideS;
  const int kbeg = blockIdx.z * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);

  float4 ra[NA], rb[NB];

  auto load_a = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int u = tid + 256 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (AMODE == 1) {  // stored [K][lda], m contiguous
        const int kr = u / (BM / 4), mc = (u % (BM / 4)) * 4;
        const int k = k0 + kr, m = m0 + mc;
        if (k < kend && m < g.M) {
          const float* p = A + (long long)k * g.lda + m;
          if (g.vecA) {
            v = *reinterpret_cast<const float4*>(p);
          } else {  // leading dimension / extent not a multiple of 4: element loads with per-element bounds
            v.x = p[0];
            if (m + 1 < g.M) v.y = p[1];
            if (m + 2 < g.M) v.z = p[2];
            if (m + 3 < g.M) v.w = p[3];
          }
        }
      } else {
        const int row = u / (BK / 4), kc = (u % (BK / 4)) * 4;
        const int m = m0 + row, k = k0 + kc;
        if (m < g.M && k < kend) {
          if (AMODE == 0) {
            const float* p = A + (long long)m * g.lda + k;
            if (g.vecA) {
              v = *reinterpret_cast<const float4*>(p);
            } else {
              v.x = p[0];
              if (k + 1 < kend) v.y = p[1];
              if (k + 2 < kend) v.z = p[2];
              if (k + 3 < kend) v.w = p[3];
            }
          } else {  // patch gather: row = (frame, py, px), k = (c, i, j)
            const int f = m / g.n_patch, pi = m - f * g.n_patch;
            const int py = pi / g.gw, px = pi - py * g.gw;
            const int pp = g.P * g.P;
            const int c = k / pp, rem = k - c * pp;
            const int ii = rem / g.P, jj = rem - ii * g.P;
            const int src = g.frame_map ? g.frame_map[f] : f;
            v = *reinterpret_cast<const float4*>(
                A + (((long long)src * g.Cin + c) * g.H + (py * g.P + ii)) * g.W + px * g.P + jj);
          }
        }
      }
      ra[i] = v;
    }
  };
  auto load_b = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int u = tid + 256 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (BMODE == 1) {  // stored [K][ldb], n contiguous
        const int kr = u / (BN / 4), nc = (u % (BN / 4)) * 4;
        const int k = k0 + kr, n = n0 + nc;
        if (k < kend && n < g.N) {
          const float* p = B + (long long)k * g.ldb + n;
          if (g.vecB) {
            v = *reinterpret_cast<const float4*>(p);
          } else {
            v.x = p[0];
            if (n + 1 < g.N) v.y = p[1];
            if (n + 2 < g.N) v.z = p[2];
            if (n + 3 < g.N) v.w = p[3];
          }
        }
      } else {  // stored [N][ldb], k contiguous
        const int row = u / (BK / 4), kc = (u % (BK / 4)) * 4;
        const int n = n0 + row, k = k0 + kc;
        if (n < g.N && k < kend) {
          const float* p = B + (long long)n * g.ldb + k;
          if (g.vecB) {
            v = *reinterpret_cast<const float4*>(p);
          } else {
            v.x = p[0];
            if (k + 1 < kend) v.y = p[1];
            if (k + 2 < kend) v.z = p[2];
            if (k + 3 < kend) v.w = p[3];
          }
        }
      }
      rb[i] = v;
    }
  };
  auto store_a = [&](int buf) {
    float* dst = As + buf * ASZ;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int u = tid + 256 * i;
      if (AMODE == 1) {
        const int kr = u / (BM / 4), mc = (u % (BM / 4)) * 4;
        *reinterpret_cast<float4*>(dst + kr * LDA + mc) = ra[i];
      } else {
        const int row = u / (BK / 4), kc = (u % (BK / 4)) * 4;
        if (AK) {
          *reinterpret_cast<float4*>(dst + row * LDA + kc) = ra[i];
        } else {  // transpose into the k-major image
          dst[(kc + 0) * LDA + row] = ra[i].x;
          dst[(kc + 1) * LDA + row] = ra[i].y;
          dst[(kc + 2) * LDA + row] = ra[i].z;
          dst[(kc + 3) * LDA + row] = ra[i].w;
        }
      }
    }
  };
  auto store_b = [&](int buf) {
    float* dst = Bs + buf * BSZ;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int u = tid + 256 * i;
      if (BMODE == 1) {
        const int kr = u / (BN / 4), nc = (u % (BN / 4)) * 4;
        *reinterpret_cast<float4*>(dst + kr * LDB + nc) = rb[i];
      } else {
        const int row = u / (BK / 4), kc = (u % (BK / 4)) * 4;
        if (BKL) {
          *reinterpret_cast<float4*>(dst + row * LDB + kc) = rb[i];
        } else {
          dst[(kc + 0) * LDB + row] = rb[i].x;
          dst[(kc + 1) * LDB + row] = rb[i].y;
          dst[(kc + 2) * LDB + row] = rb[i].z;
          dst[(kc + 3) * LDB + row] = rb[i].w;
        }
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk > 0) {
    load_a(kbeg);
    load_b(kbeg);
    store_a(0);
    store_b(0);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) {
      load_a(kbeg + (kt + 1) * BK);
      load_b(kbeg + (kt + 1) * BK);
    }
    const float* pa = As + buf * ASZ + (AK ? (wm * (32 * WM) + r) * LDA + 4 * h : (4 * h) * LDA + wm * (32 * WM) + r);
    const float* pb = Bs + buf * BSZ + (BKL ? (wn * (32 * WN) + r) * LDB + 4 * h : (4 * h) * LDB + wn * (32 * WN) + r);
#pragma unroll
    for (int j = 0; j < BK / 8; ++j) {
      float a[WM][4], b[WN][4];
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        if (AK) {
          const float4 v = *reinterpret_cast<const float4*>(pa + i * 32 * LDA + 8 * j);
          a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) a[i][q] = pa[(8 * j + q) * LDA + i * 32];
        }
      }
#pragma unroll
      for (int n = 0; n < WN; ++n) {
        if (BKL) {
          const float4 v = *reinterpret_cast<const float4*>(pb + n * 32 * LDB + 8 * j);
          b[n][0] = v.x; b[n][1] = v.y; b[n][2] = v.z; b[n][3] = v.w;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) b[n][q] = pb[(8 * j + q) * LDB + n * 32];
        }
      }
      if constexpr (BF16) {
        gemm_s16x4 ap[WM], bp[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) ap[i] = gemm_pack_bf16(a[i]);
#pragma unroll
        for (int n = 0; n < WN; ++n) bp[n] = gemm_pack_bf16(b[n]);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int n = 0; n < WN; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ap[i], bp[n], acc[i][n], 0, 0, 0);
      } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int n = 0; n < WN; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[n][q], acc[i][n], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) {
      store_a(buf ^ 1);
      store_b(buf ^ 1);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 32x32 tile: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < WM; ++i) {
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = n0 + wn * (32 * WN) + j * 32 + r;
      if (n >= g.N) continue;
      const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * (32 * WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m >= g.M) continue;
        float v = acc[i][j][e] * g.alpha + bias;
        long long off;
        if (AMODE == 2) {
          const int f = m / g.n_patch, pi = m - f * g.n_patch;
          off = ((long long)f * (g.n_patch + 1) + 1 + pi) * g.ldc + n;
          v += g.pos[(long long)(1 + pi) * g.N + n];
        } else {
          off = (long long)m * g.ldc + n;
        }
        if (g.pre_out) g.pre_out[off] = v;
        if (g.act == 1) v = gelu_f(v);
        if (g.row_scale) v *= g.row_scale[m];
        if (g.gelu_pre) v *= gelu_grad_f(g.gelu_pre[off]);
        if (g.residual) v += g.residual[off];
        C[off] = v;
      }
    }
  }
}

// out[i] = sum_s partial[s][i] (fixed order), 16 B per lane
// the partials of one float4 in slice order (the sum's order is the slices': run-to-run and layout-independent bits), the loads of four
// slices in flight at a time: a thread's `splits` loads were one dependent chain of L2 / HBM latencies (round 6: the 1536 x 384 weight
// gradient's fold of 10 slices took 15.5 us for 26 MB)
__device__ __forceinline__ float4 splitk_fold4(const float* __restrict__ p, int splits, long long stride) {
  float4 s = *reinterpret_cast<const float4*>(p);
  int z = 1;
  for (; z + 4 <= splits; z += 4) {
    const float4 v0 = *reinterpret_cast<const float4*>(p + z * stride), v1 = *reinterpret_cast<const float4*>(p + (z + 1) * stride),
                 v2 = *reinterpret_cast<const float4*>(p + (z + 2) * stride), v3 = *reinterpret_cast<const float4*>(p + (z + 3) * stride);
    s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
    s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
    s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
    s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
  }
  for (; z < splits; ++z) {
    const float4 v = *reinterpret_cast<const float4*>(p + z * stride);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  return s;
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, long long n,
                                                            int splits, long long stride) {
  long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  const long long step = (long long)gridDim.x * 256 * 4;
  for (; i < n; i += step) {
    *reinterpret_cast<float4*>(out + i) = splitk_fold4(partial + i, splits, stride);
  }
}

// The same launch also folds the bias gradient's column partials [colparts][N] (written by the weight-gradient kernel) into db:
// workgroups >= blocks_main take 64 columns each, in colsum_stage2's order (common.hpp: colsum_fold_block).
__global__ __launch_bounds__(256) void splitk_reduce_colfold_kernel(const float* __restrict__ partial, float* __restrict__ out, long long n,
                                                                    int splits, long long stride, int blocks_main,
                                                                    const float* __restrict__ colpart, float* __restrict__ db, int colparts,
                                                                    int N) {
  __shared__ float red[4][64];
  if ((int)blockIdx.x >= blocks_main) {
    colsum_fold_block(colpart, db, colparts, N, N, 0, (int)blockIdx.x - blocks_main, red);
    return;
  }
  long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  const long long step = (long long)blocks_main * 256 * 4;
  for (; i < n; i += step) {
    *reinterpret_cast<float4*>(out + i) = splitk_fold4(partial + i, splits, stride);
  }
}

int launch_splitk_reduce(const float* partial, float* out, long long n, int splits, long long stride, hipStream_t s) {
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, partial, out, n, splits, stride);
  TT_CHECK_LAUNCH("splitk_reduce");
  return TT_OK;
}

int launch_splitk_reduce_colfold(const float* partial, float* out, long long n, int splits, long long stride, const float* colpart, float* db,
                                 int colparts, int N, hipStream_t s) {
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(splitk_reduce_colfold_kernel, dim3((unsigned)blocks + (N + 63) / 64), dim3(256), 0, s, partial, out, n, splits, stride, (int)blocks,
                     colpart, db, colparts, N);
  TT_CHECK_LAUNCH("splitk_reduce_colfold");
  return TT_OK;
}

int gemm_tile_choice(int M, int N, int batch) {
  static const int forced = [] { const char* e = getenv("TT_FORCE_TILE"); return e ? atoi(e) : -1; }();  // tuning aid
  if (forced >= 0 && forced <= 3) return forced;
  struct Cfg { int wm, wn; double pen; };
  // (128 x 64 beats 64 x 128 at equal tile counts on every ViT-S/16 block shape - 813.6 vs 836.3 us per block of 128 frames,
  // tools/bench_linear.py with TT_FORCE_TILE: the streamed activation rows are the taller operand, the L2-resident weights the narrower)
  const Cfg cfgs[4] = {{2, 2, 1.00}, {1, 2, 1.04}, {2, 1, 1.03}, {1, 1, 1.10}};
  int best = 0;
  double best_cost = 1e300;
  for (int c = 0; c < 4; ++c) {
    const long long bm = 64 * cfgs[c].wm, bn = 64 * cfgs[c].wn;
    const long long tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * (long long)batch;
    const double cost = (double)((tiles + 255) / 256) * (double)(bm * bn) * cfgs[c].pen;
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  // A grid of half a chip to a chip of workgroups is latency-bound by their K loops, and several of these small workgroups share a CU:
  // the smallest tile then halves / quarters that loop at the same residency (round 6, in the step: the prototype-score products -
  // 6272 x 200 x 256, 196 tiles of 128 x 64 - and their gradients: C2 -0.25 %, C2 in f32 -0.7 %; below 128 tiles - C1's launches -
  // it measured no better: left alone; TT_TILE_RULE=0 restores the plain rule; profiles/r06_step_knob_sweeps.txt).
  static const bool small_rule = [] { const char* e = getenv("TT_TILE_RULE"); return e ? atoi(e) != 0 : true; }();
  if (small_rule) {
    const long long bm = 64 * cfgs[best].wm, bn = 64 * cfgs[best].wn;
    const long long t = ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * (long long)batch;
    if (t >= 128 && t < 256) best = 3;
  }
  return best;
}

#ifndef TT_BK
#define TT_BK 16
#endif
constexpr int kBK = TT_BK;

template <int WM, int WN, int AMODE, int BMODE>
static int launch_cfg(const GemmArgs& g, int batch, hipStream_t s) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
  dim3 grid(ntm * ntn, batch, g.splits);
  if constexpr (AMODE == 0 && BMODE == 0) {
    if (g.bf16) {
      // (a 64-deep-slab instance of this one - a quarter of the barriers - was measured on the C4 label propagation: 795 vs 700 us for the
      // whole call, tools/lp_time.py; f32: 775 - the extra LDS costs a resident workgroup)
      hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, AMODE, BMODE, kBK, true>), grid, dim3(256), 0, s, g);
      TT_CHECK_LAUNCH("gemm_f32(bf16 products)");
      return TT_OK;
    }
  }
  hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, AMODE, BMODE, kBK>), grid, dim3(256), 0, s, g);
  TT_CHECK_LAUNCH("gemm_f32");
  return TT_OK;
}

template <int AMODE, int BMODE>
static int launch_mode(const GemmArgs& g, int batch, int tile, hipStream_t s) {
  switch (tile) {
    case 0: return launch_cfg<2, 2, AMODE, BMODE>(g, batch, s);
    case 1: return launch_cfg<1, 2, AMODE, BMODE>(g, batch, s);
    case 2: return launch_cfg<2, 1, AMODE, BMODE>(g, batch, s);
    default: return launch_cfg<1, 1, AMODE, BMODE>(g, batch, s);
  }
}

// Split-K plan for a product whose output is too small to fill the chip: returns the number of K slices (1 = none).
int gemm_splitk_choice(int M, int N, int K, int* tile_out) {
  // Measured on the block / head weight-gradient shapes (tools/bench_kernels.py, tile x target sweep): 64x64 output tiles
  // with about one resident wave of workgroups (~1024) win over larger tiles with more, shorter K slices
  // (fc2 110 vs 140 us, qkv 96 vs 117, head 157 vs 187).  The launcher must use the tile this plan was made for.
  const int tile = 3;
  const long long tiles = (long long)((M + 63) / 64) * ((N + 63) / 64);
  if (tile_out) *tile_out = tile;
  if (tiles >= 768 || K < 1024) return 1;
  static const int target_env = [] { const char* e = getenv("TT_SPLIT_TARGET"); return e ? atoi(e) : 0; }();  // tuning aids
  static const int mink = [] { const char* e = getenv("TT_SPLIT_MINK"); return e ? atoi(e) : 256; }();
  const int target = target_env ? target_env : 1024;
  int s = (int)((target + tiles - 1) / tiles);
  const int smax = K / mink;                         // keep >= mink of K per slice
  if (s > smax) s = smax;
  if (s > 32) s = 32;
  return s < 1 ? 1 : s;
}

int launch_gemm(const GemmArgs& g_in, int amode, int bmode, int batch, hipStream_t s) {
  GemmArgs g = g_in;
  TT_REQUIRE(g.A && g.B && g.C, "gemm: null operand");
  TT_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && batch > 0, "gemm: bad shape M=%d N=%d K=%d batch=%d", g.M, g.N, g.K, batch);
  if (amode == 2) TT_REQUIRE(g.P % 4 == 0 && g.W % 4 == 0 && aligned16(g.A), "gemm: patch size and image width must be multiples of 4");
  if (g.splits <= 1) {
    g.splits = 1;
    g.kchunk = g.K;
    g.strideS = 0;
  }
  // 16-byte loads need an aligned base, a leading dimension that keeps rows aligned, and whole float4s inside the
  // extent along the contiguous axis; otherwise (e.g. 50 prototypes) the kernel falls back to element loads
  const bool batch_ok = batch == 1 || ((g.strideA % 4 == 0) && (g.strideB % 4 == 0) && (g.strideA2 % 4 == 0) && (g.strideB2 % 4 == 0));
  const int kq = (g.K % 4 == 0) && (g.kchunk % 4 == 0);
  g.vecA = aligned16(g.A) && batch_ok && g.lda % 4 == 0 && (amode == 1 ? g.M % 4 == 0 : kq);
  g.vecB = aligned16(g.B) && batch_ok && g.ldb % 4 == 0 && (bmode == 1 ? g.N % 4 == 0 : kq);
  if (amode == 2) g.vecA = 1;
  const int tile = gemm_tile_choice(g.M, g.N, batch * g.splits);
  if (amode == 0 && bmode == 0) return launch_mode<0, 0>(g, batch, tile, s);
  if (amode == 0 && bmode == 1) return launch_mode<0, 1>(g, batch, tile, s);
  if (amode == 1 && bmode == 1) return launch_mode<1, 1>(g, batch, tile, s);
  if (amode == 1 && bmode == 0) return launch_mode<1, 0>(g, batch, tile, s);
  if (amode == 2 && bmode == 0) return launch_mode<2, 0>(g, batch, tile, s);
  set_error("gemm: unsupported operand layout (%d, %d)", amode, bmode);
  return TT_EUNSUPPORTED;
}

static GemmArgs base_args(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc) {
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.alpha = 1.f;
  g.splits = 1;
  return g;
}

// plain NT product with a two-level batch, used by other translation units (label propagation): problem (i, o), i < batch_inner, o < batch_outer, at i * s?1 + o * s?2
int launch_gemm_plain2(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int batch_inner,
                       int batch_outer, long long sA, long long sB, long long sC, long long sA2, long long sB2, long long sC2, int bf16,
                       hipStream_t s) {
  GemmArgs g = base_args(A, B, C, M, N, K, lda, ldb, ldc);
  g.strideA = sA; g.strideB = sB; g.strideC = sC;
  g.batch_inner = batch_inner;
  g.strideA2 = sA2; g.strideB2 = sB2; g.strideC2 = sC2;
  g.bf16 = bf16 != 0;   // precision 2, the "bf16" mode (BASELINE C4's path): these products on bf16 MFMA too, as torch.autocast would run them
  return launch_gemm(g, 0, 0, batch_inner * batch_outer, s);
}

}  // namespace tt

using tt::GemmArgs;
using tt::base_args;

namespace tt {
int try_launch_gemm_nt_fast(const float* A, const float* B, float* C, int M, int N, int K, const float* bias,
                            const float* residual, float* pre_out, int act, hipStream_t s);
int try_launch_gemm_nt_bf16(const float* A, const float* B, float* C, int M, int N, int K, const float* bias, const float* residual,
                            float* pre_out, int act, int npass, hipStream_t s);
int try_launch_dgrad_fast(const float* dy, const float* w, const float* gelu_pre, float* dx, int M, int N, int K, hipStream_t s);
int try_launch_wgrad_fast(const float* dy, const float* x, float* out, int M, int N, int K, int splits, int kchunk, float* colpart,
                          int* colparts, hipStream_t s);
int launch_colsum_fold(const float* partial, float* out, int chunks, int N, hipStream_t s);  // rowops.hip
int try_launch_bwd_fused(const float* dy, const float* w, const float* x, const float* gelu_pre, float* dx, float* wpart, int M, int N, int K,
                         int splits, int kchunk, float* colpart, int* colparts, hipStream_t s);    // gemm_bwd_fast.hip
}

extern "C" int tt_gemm_tile_choice(int M, int N, int batch) { return tt::gemm_tile_choice(M, N, batch); }

// Which kernel an fp32 tt_linear_fwd of this shape runs (16-byte aligned operands assumed): bits 0-1 = tile
// (0: 128x128, 1: 64x128, 2: 128x64, 3: 64x64), bit 8 set = the lean whole-tile instance gemm_nt_fast_kernel, clear = the
// general bounds-checked gemm_f32_kernel.  Mirrors try_launch_gemm_nt_fast; profilers label launches with it.
extern "C" int tt_linear_fwd_route(int M, int N, int K) {
  const int tile = tt::gemm_tile_choice(M, N, 1);
  if (K % 16 != 0 || K < 16) return tile;
  const int bn = (tile == 0 || tile == 1) ? 128 : 64;
  if (N % bn == 0) return tile | 256;
  if (N % 64 == 0) return 3 | 256;
  return tile;
}

extern "C" int tt_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                           int a_mmajor, int b_nmajor, float alpha, int batch, long long strideA, long long strideB,
                           long long strideC, tt_stream_t stream) {
  GemmArgs g = base_args(A, B, C, M, N, K, lda, ldb, ldc);
  g.alpha = alpha;
  g.strideA = strideA; g.strideB = strideB; g.strideC = strideC;
  return tt::launch_gemm(g, a_mmajor ? 1 : 0, b_nmajor ? 1 : 0, batch, tt::as_stream(stream));
}

extern "C" int tt_linear_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y,
                             float* pre_act, int M, int N, int K, int act, int precision, tt_stream_t stream) {
  if (precision < TT_PRECISION_F32 || precision > TT_PRECISION_BF16) {
    tt::set_error("linear_fwd: precision must be 0 (f32), 1 (bf16x3) or 2 (bf16), got %d", precision);
    return TT_EINVAL;
  }
  if (x && w && y && M > 0 && N > 0 && K > 0) {
    if (precision != TT_PRECISION_F32) {  // opt-in bf16 / split-bf16 MFMA instances (gemm_nt_bf16.hip)
      const int rc = tt::try_launch_gemm_nt_bf16(x, w, y, M, N, K, bias, residual, pre_act, act, precision == TT_PRECISION_BF16X3 ? 3 : 1,
                                                 tt::as_stream(stream));
      if (rc <= 0) return rc;
    }
    // whole-tile, aligned shapes take the lean f32 kernel (gemm_nt_fast.hip)
    const int rc = tt::try_launch_gemm_nt_fast(x, w, y, M, N, K, bias, residual, pre_act, act, tt::as_stream(stream));
    if (rc <= 0) return rc;
  }
  GemmArgs g = base_args(x, w, y, M, N, K, K, K, N);
  g.bias = bias; g.residual = residual; g.pre_out = pre_act; g.act = act;
  return tt::launch_gemm(g, 0, 0, 1, tt::as_stream(stream));
}

extern "C" int tt_linear_bwd_data(const float* dy, const float* w, const float* gelu_pre, float* dx, int M, int N,
                                  int K, tt_stream_t stream) {
  // dx[M,K] = dy[M,N] @ w[N,K]: reduction over N; w is "n-major" for this product (stored [N_red][K_out]).
  if (dy && w && dx && M > 0 && N > 0 && K > 0) {  // lean instance for whole-tile output columns (gemm_bwd_fast.hip)
    const int rc = tt::try_launch_dgrad_fast(dy, w, gelu_pre, dx, M, N, K, tt::as_stream(stream));
    if (rc <= 0) return rc;
  }
  GemmArgs g = base_args(dy, w, dx, M, K, N, N, K, K);
  g.gelu_pre = gelu_pre;
  return tt::launch_gemm(g, 0, 1, 1, tt::as_stream(stream));
}

extern "C" size_t tt_linear_bwd_weight_workspace_bytes(int M, int N, int K) {
  const int s = (((long long)N * K) % 4 == 0) ? tt::gemm_splitk_choice(N, K, M, nullptr) : 1;
  // split-K partials of dw, then the [s * K / 64][N] partial column sums of dy that the same launch produces for db
  const size_t split = (s > 1 ? (size_t)s * N * K * sizeof(float) : 0) + (size_t)s * ((K + 63) / 64) * N * sizeof(float);
  const size_t cs = tt_colsum_workspace_bytes(M, N);
  return split > cs ? split : cs;
}

extern "C" int tt_linear_bwd_weight(const float* dy, const float* x, float* dw, float* db, int M, int N, int K,
                                    void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  // dw[N,K] = dy[M,N]^T @ x[M,K]: reduction over M; both operands are stored [M_red][*].
  GemmArgs g = base_args(dy, x, dw, N, K, M, N, K, K);
  const int s = (((long long)N * K) % 4 == 0) ? tt::gemm_splitk_choice(N, K, M, nullptr) : 1;
  int rc;
  TT_REQUIRE(!db || (workspace && workspace_bytes >= tt_linear_bwd_weight_workspace_bytes(M, N, K)), "linear_bwd_weight: workspace too small");
  // where the lean kernel runs, the bias gradient (column sums of dy) is produced by the same launch: [s][N] partials
  // behind the split-K area, folded below; otherwise tt_colsum makes its own pass over dy
  float* colpart = db ? static_cast<float*>(workspace) + (s > 1 ? (size_t)s * N * K : 0) : nullptr;
  bool bias_fused = false;
  int colparts = 0;
  if (s > 1) {
    TT_REQUIRE(workspace && workspace_bytes >= (size_t)s * N * K * sizeof(float), "linear_bwd_weight: workspace too small for split-K");
    g.C = static_cast<float*>(workspace);
    g.splits = s;
    g.kchunk = ((M + s - 1) / s + tt::kBK - 1) / tt::kBK * tt::kBK;
    g.strideS = (long long)N * K;
    rc = tt::try_launch_wgrad_fast(dy, x, g.C, M, N, K, s, g.kchunk, colpart, &colparts, tt::as_stream(stream));
    bias_fused = rc == TT_OK && colpart != nullptr;
    if (rc > 0) rc = tt::launch_gemm(g, 1, 1, 1, tt::as_stream(stream));
    if (rc != TT_OK) return rc;
    const long long n = (long long)N * K;
    TT_REQUIRE(n % 4 == 0, "linear_bwd_weight: N*K must be a multiple of 4");
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (bias_fused && db) {   // one launch folds the split-K partials of dw AND the column partials of db
      hipLaunchKernelGGL(tt::splitk_reduce_colfold_kernel, dim3((unsigned)blocks + (N + 63) / 64), dim3(256), 0, tt::as_stream(stream),
                         static_cast<const float*>(workspace), dw, n, s, (long long)N * K, (int)blocks, colpart, db, colparts, N);
      TT_CHECK_LAUNCH("splitk_reduce_colfold");
      return TT_OK;
    }
    hipLaunchKernelGGL(tt::splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, tt::as_stream(stream),
                       static_cast<const float*>(workspace), dw, n, s, (long long)N * K);
    TT_CHECK_LAUNCH("splitk_reduce");
  } else {
    rc = (M % 16 == 0) ? tt::try_launch_wgrad_fast(dy, x, dw, M, N, K, 1, M, colpart, &colparts, tt::as_stream(stream)) : 1;
    bias_fused = rc == TT_OK && colpart != nullptr;
    if (rc > 0) rc = tt::launch_gemm(g, 1, 1, 1, tt::as_stream(stream));
    if (rc != TT_OK) return rc;
  }
  if (!db) return TT_OK;
  if (bias_fused) return tt::launch_colsum_fold(colpart, db, colparts, N, tt::as_stream(stream));
  return tt_colsum(dy, db, M, N, workspace, workspace_bytes, stream);
}

extern "C" int tt_linear_bwd(const float* dy, const float* w, const float* x, const float* gelu_pre, float* dx, float* dw, float* db, int M,
                             int N, int K, void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  TT_REQUIRE(dy && w && x && dx && dw && M > 0 && N > 0 && K > 0, "linear_bwd: null operand / bad shape");
  // ONE launch when the weight gradient is split along M (its plan: gemm_splitk_choice) and both lean kernels take the shapes
  const int s = (((long long)N * K) % 4 == 0) ? tt::gemm_splitk_choice(N, K, M, nullptr) : 1;
  static const bool no_fuse = getenv("TT_BWD_NO_FUSE") != nullptr;   // tuning aid
  if (s > 1 && !no_fuse && workspace && workspace_bytes >= tt_linear_bwd_weight_workspace_bytes(M, N, K)) {
    const int kchunk = ((M + s - 1) / s + tt::kBK - 1) / tt::kBK * tt::kBK;
    float* part = static_cast<float*>(workspace);
    float* colpart = db ? part + (size_t)s * N * K : nullptr;
    int colparts = 0;
    const int rc = tt::try_launch_bwd_fused(dy, w, x, gelu_pre, dx, part, M, N, K, s, kchunk, colpart, &colparts, tt::as_stream(stream));
    if (rc < 0) return rc;
    if (rc == TT_OK) {
      const long long n = (long long)N * K;
      long long blocks = (n / 4 + 255) / 256;
      if (blocks > 1024) blocks = 1024;
      if (db) {
        hipLaunchKernelGGL(tt::splitk_reduce_colfold_kernel, dim3((unsigned)blocks + (N + 63) / 64), dim3(256), 0, tt::as_stream(stream),
                           static_cast<const float*>(workspace), dw, n, s, (long long)N * K, (int)blocks, colpart, db, colparts, N);
        TT_CHECK_LAUNCH("splitk_reduce_colfold");
      } else {
        hipLaunchKernelGGL(tt::splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, tt::as_stream(stream),
                           static_cast<const float*>(workspace), dw, n, s, (long long)N * K);
        TT_CHECK_LAUNCH("splitk_reduce");
      }
      return TT_OK;
    }
  }
  const int rc = tt_linear_bwd_weight(dy, x, dw, db, M, N, K, workspace, workspace_bytes, stream);
  if (rc != TT_OK) return rc;
  return tt_linear_bwd_data(dy, w, gelu_pre, dx, M, N, K, stream);
}

namespace tt {
int try_launch_patch_embed_fast(const float* img, const int* frame_map, const float* w, const float* bias, const float* pos, float* tokens,
                                int F, int C, int H, int W, int P, int D, hipStream_t s);
}

extern "C" int tt_patch_embed_gemm(const float* img, const int32_t* frame_map, const float* w, const float* bias,
                                   const float* pos, float* tokens, int F, int C, int H, int W, int P, int D,
                                   tt_stream_t stream) {
  static const bool lean = [] { const char* e = getenv("TT_PATCH_LEAN"); return !e || atoi(e) != 0; }();   // tuning aid
  if (lean) {
    const int rc = tt::try_launch_patch_embed_fast(img, frame_map, w, bias, pos, tokens, F, C, H, W, P, D, tt::as_stream(stream));
    if (rc <= 0) return rc;
  }
  const int gw = W / P, gh = H / P, n = gw * gh;
  GemmArgs g = base_args(img, w, tokens, F * n, D, C * P * P, 0, C * P * P, D);
  g.bias = bias; g.frame_map = frame_map; g.Cin = C; g.H = H; g.W = W; g.P = P; g.gw = gw; g.n_patch = n; g.pos = pos;
  return tt::launch_gemm(g, 2, 0, 1, tt::as_stream(stream));
}
