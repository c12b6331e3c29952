// Backward of the fused attention core (autograd of dino_vision_transformer.py:125-129) for the two trainable
// blocks, fp32 on v_mfma_f32_16x16x4_f32.  Probabilities are recomputed from q, k and the forward's
// log-sum-exp (nothing N x N is stored).  Two launches:
//   dq kernel  - a wave owns 16 queries; per 16-key tile: S^T = K Q^T, dP^T = V dO^T, dS^T = P^T (dP^T - delta),
//                dQ^T += K^T dS^T.  It also produces delta_q = <dO_q, O_q> and stores it for the second kernel.
//   dkv kernel - a wave owns 16 keys; per 16-query tile: S = Q K^T, dP = dO V^T, P, dS, dV^T += dO^T P, dK^T += Q^T dS.
// In both, the freshly computed accumulator tile (P / dS) is used as the B operand of the next MFMA directly from
// registers: the C/D layout (col = lane & 15, row = 4 * (lane >> 4) + e) is the B layout for k = row, so no LDS
// round trip or shuffle is needed.  No atomics: each output element has exactly one owner (deterministic).
//
// BF16 = true (tt_attention_bwd_bf16, BASELINE C4's bf16 path): the same kernels with every group of four k = 4 f32 MFMAs replaced by ONE
// v_mfma_f32_16x16x16_bf16 on operands rounded to bf16 - what torch.autocast makes of the backward of q k^T and attn v (bf16 products,
// fp32 accumulation; softmax statistics, delta, P and dS computed in fp32, P and dS rounded where they enter a product).  It is the
// SAME data movement: element j of a lane's bf16 operand for k-step t is the fp32 kernel's register s = 4 t + j (any assignment of the
// contraction index to (lane group, element) is valid as long as both operands use it), and a 16 x 16 accumulator tile is the B operand
// of the wider MFMA as it stands (k = row = 4 g + e, four elements per lane).
// PAIR = true (tt_attention_bwd_pairs, round 6: the "f16x3" mode's backward): all five products of a tile on fp16 (hi, lo) pairs -
// three v_mfma_f32_16x16x32_f16 per group of EIGHT fp32 MFMAs (hi hi into one accumulator, hi lo + lo hi into a second one, folded with
// the exact 2^-11: gemm_pairs8.hip), 48 matrix-pipe cycles instead of 256.  What makes it pay is the operand layout (a first form that
// kept the fp32 LDS layout and unzipped {hi, lo} words with v_perm_b32 was slower than the fp32 kernels):
//   * the ROW-WISE products (S = Q K^T, dP = dO V^T) contract over the head dimension: their chunk operands are split ONCE on the way
//     into LDS, into a hi and a lo plane of fp16 rows in natural order (128-byte rows, 16-byte chunks XOR-swizzled by row & 7), and the
//     contraction index is assigned so that a lane's sixteen elements are CONTIGUOUS (lane group g owns head dimensions 16 g .. 16 g +
//     15; element j of k-step t is dimension 16 g + 8 t + j - any assignment is valid as long as both operands use it): a k-step's
//     fragment is ONE ds_read_b128 per plane where the fp32 form issues sixteen ds_read_b32;
//   * the TRANSPOSED products (dQ^T += K^T dS^T; dV^T += dO^T P, dK^T += Q^T dS) contract over the chunk's ROWS: their chunk operands are
//     written a second time, transposed ([dim][row], the rows in the order the K = 32 fragment wants them - tplane_at); P and dS of BOTH
//     16-row tiles of a chunk are split where they leave the accumulators and are the B fragment as they stand: one K = 32 group per 16
//     head dimensions and chunk;
//   * dO is a GRADIENT (1e-3 ... 1e-8): it is multiplied by the power of two S that brings max |dO| into [2^13, 2^14) before the split
//     (exact; the maximum comes from the amax slot of the kernel that wrote dO, or from one small launch in front) and dP is divided by
//     it again; dS = P (dP - delta) is a sum over the 64 head dimensions of dO V, up to 64 max |V| times larger than dO: it is split as
//     dS S 2^-12 (a dS beyond fp16's range - |V| of a few hundred - raises the caller's range flag);
//   * registers are capped so that the dq kernel runs three and the dkv kernel two waves per SIMD (LDS: 49 KB / 66 KB per workgroup).
// profiles/r06_attention_bwd_pairs.txt has every step of that with its measurement.
#include "common.hpp"

namespace tt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ s16x4 pack_bf16(float a, float b, float c, float d) {
  const bf16x4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
  return __builtin_bit_cast(s16x4, v);
}
__device__ __forceinline__ f32x4 mma_bf16(s16x4 a, s16x4 b, f32x4 acc) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0); }

// ---- fp16 pairs (PAIR = true)
__device__ __forceinline__ bool split4(float a, float b, float c, float d, f16x4& hi, f16x4& lo) {   // returns "a hi is not finite"
  _Float16 h0, l0, h1, l1, h2, l2, h3, l3;
  split_pair(a, h0, l0); split_pair(b, h1, l1); split_pair(c, h2, l2); split_pair(d, h3, l3);
  hi = (f16x4){h0, h1, h2, h3};
  lo = (f16x4){l0, l1, l2, l3};
  return pair_hi_bad(h0) || pair_hi_bad(h1) || pair_hi_bad(h2) || pair_hi_bad(h3);
}
// acc1 += ah bh, acc2 += ah bl + al bh (the lo halves carry 2^11: the product is acc1 + 2^-11 acc2) on v_mfma_f32_16x16x32_f16 (a lane
// holds eight consecutive k)
__device__ __forceinline__ void mma_pair8(f16x8 ah, f16x8 al, f16x8 bh, f16x8 bl, f32x4& acc1, f32x4& acc2) {
  acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc1, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc2, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc2, 0, 0, 0);
}
__device__ __forceinline__ bool split8(const float* v, float s, f16x8& hi, f16x8& lo) {
  bool bad = false;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    _Float16 h, l;
    split_pair(v[j] * s, h, l);
    hi[j] = h;
    lo[j] = l;
    bad |= pair_hi_bad(h);
  }
  return bad;
}
__device__ __forceinline__ f32x4 fold_pair(f32x4 a1, f32x4 a2, float m) {   // (acc1 + 2^-11 acc2) m
  return (f32x4){fmaf(a2[0], kPairInvScale, a1[0]) * m, fmaf(a2[1], kPairInvScale, a1[1]) * m, fmaf(a2[2], kPairInvScale, a1[2]) * m,
                 fmaf(a2[3], kPairInvScale, a1[3]) * m};
}
// max |dO| (per-workgroup partials; both kernels fold them themselves - max is order-independent) and the scale it gives
constexpr int kBwdAmaxParts = 256;
__global__ __launch_bounds__(256) void attention_bwd_amax_kernel(const float* __restrict__ x, long long n, float* __restrict__ part) {
  __shared__ float red[4];
  float m = 0.f;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * 1024) {
    const float4 v = *reinterpret_cast<const float4*>(x + i);   // (n % 4 == 0: head_dim 64)
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
// (n_part < 0: the -n_part ways of a producer's amax slot, kAmaxStride floats apart - common.hpp amax_publish)
__device__ __forceinline__ float bwd_grad_scale(const float* __restrict__ part, int n_part, float* sred) {   // every thread of the workgroup calls it
  float m = n_part < 0 ? ((int)threadIdx.x < -n_part ? part[threadIdx.x * kAmaxStride] : 0.f) : ((int)threadIdx.x < n_part ? part[threadIdx.x] : 0.f);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
  if (!(m > 0.f) || !(m < INFINITY)) return 1.0f;
  int e = 13 - ilogbf(m);
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return ldexpf(1.0f, e);
}
// fp16 elements per row of a pair plane: 128 bytes, no padding (with 16 bytes of it the dq kernel's 54 KB fit two workgroups per CU where
// three of the fp32 form fit); the 16-byte chunks of row r sit at chunk ^ (r & 7) instead - rows r and r + 8 then meet in a bank (a
// two-way conflict on eight reads per tile step) and every other pair of rows does not
constexpr int PSTR = 64;

constexpr int BHD = 64, BCH = 32, BSTR = 68;  // chunk rows, LDS row stride (16 g + i banks for the transposed reads)

// Loads a 64 x 64 tile (rows row0.., row stride ld) through LDS and returns this lane's 16 operand values
// r[s] = tile[wave * 16 + (lane & 15)][4 s + (lane >> 4)].
__device__ __forceinline__ void tile_to_regs(const float* __restrict__ src, long long ld, int row0, int nrows, float* stage,
                                             float (&r)[16], int tid) {
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + row < nrows) v = *reinterpret_cast<const float4*>(src + (long long)(row0 + row) * ld + c4);
    *reinterpret_cast<float4*>(stage + row * BSTR + c4) = v;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 16; ++s) r[s] = stage[(wave * 16 + li) * BSTR + 4 * s + g];
}
// ... with the contraction index assigned as the PAIR products use it: r[4 t + j] = tile[wave * 16 + (lane & 15)][16 (lane >> 4) + 4 t + j]
__device__ __forceinline__ void tile_to_regs_contig(const float* __restrict__ src, long long ld, int row0, int nrows, float* stage,
                                                    float (&r)[16], int tid) {
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + row < nrows) v = *reinterpret_cast<const float4*>(src + (long long)(row0 + row) * ld + c4);
    *reinterpret_cast<float4*>(stage + row * BSTR + c4) = v;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float4 v = *reinterpret_cast<const float4*>(stage + (wave * 16 + li) * BSTR + 16 * g + 4 * t);
    r[4 * t] = v.x; r[4 * t + 1] = v.y; r[4 * t + 2] = v.z; r[4 * t + 3] = v.w;
  }
}
// a float4 of a chunk row -> four fp16 his and four los (times s first) at the same position of the two planes
__device__ __forceinline__ bool plane_write4(_Float16* hi_row, _Float16* lo_row, int row, int c4, float4 v, float s) {
  f16x4 h, l;
  const bool bad = split4(v.x * s, v.y * s, v.z * s, v.w * s, h, l);
  const int at = (((c4 >> 3) ^ (row & 7)) << 3) + (c4 & 7);
  *reinterpret_cast<f16x4*>(hi_row + at) = h;
  *reinterpret_cast<f16x4*>(lo_row + at) = l;
  return bad;
}
// TRANSPOSED planes [dim 0 .. 63][32 chunk rows] (the operands of the products that contract over the chunk's rows: K^T, Q^T, dO^T): chunk
// row r = 16 t2 + 4 g + e sits at position 8 g + 4 t2 + e of a plane row - the eight elements lane group g needs as its K = 32 fragment
// (the B operand, P / dS, holds rows 4 g + e of tile t2 = 0 and of tile t2 = 1 in its accumulators) are ONE 16-byte chunk, chunk g, stored
// at g ^ ((dim >> 2) & 3): conflict-free ds_read_b128 (16 lanes = 16 dims: four bank quarters x four chunk slots).
constexpr int TSTR = 32;
__device__ __forceinline__ int tplane_at(int dim, int r) {
  const int gq = (r >> 2) & 3;
  return dim * TSTR + ((gq ^ ((dim >> 2) & 3)) << 3) + ((r >> 4) << 2) + (r & 3);
}
__device__ __forceinline__ bool tplane_write4(_Float16* hi, _Float16* lo, int r, int c4, float4 v, float s) {
  f16x4 h, l;
  const bool bad = split4(v.x * s, v.y * s, v.z * s, v.w * s, h, l);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int at = tplane_at(c4 + i, r);
    hi[at] = h[i];
    lo[at] = l[i];
  }
  return bad;
}
__device__ __forceinline__ f16x8 tplane_frag(const _Float16* plane, int dim, int g) {
  return *reinterpret_cast<const f16x8*>(plane + dim * TSTR + ((g ^ ((dim >> 2) & 3)) << 3));
}
// this lane's two k-step fragments of a plane row: elements 16 g .. 16 g + 7 and 16 g + 8 .. 16 g + 15 (one 16-byte chunk each)
__device__ __forceinline__ void plane_frags(const _Float16* row_ptr, int row, int g, f16x8 (&f)[2]) {
  f[0] = *reinterpret_cast<const f16x8*>(row_ptr + (((2 * g) ^ (row & 7)) << 3));
  f[1] = *reinterpret_cast<const f16x8*>(row_ptr + (((2 * g + 1) ^ (row & 7)) << 3));
}

template <bool BF16, bool PAIR = false>
__global__ __launch_bounds__(256, PAIR ? 3 : 1) void attention_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                               const float* __restrict__ dout, const float* __restrict__ lse,
                                                               float* __restrict__ dqkv, float* __restrict__ delta, float* __restrict__ amax_out, int N, int H,
                                                               int FH, float scale, const float* __restrict__ amax_part = nullptr, int n_part = 0,
                                                               int* range_flag = nullptr) {
  // PAIR: four pair planes [2][32][64] fp16 (K hi, K lo, V hi, V lo; the prologue's stage on top of them) | two transposed planes
  // [2][64][32] fp16 (K^T hi, lo)
  __shared__ __attribute__((aligned(16))) float smem[PAIR ? 4 * BCH * PSTR + 2 * BHD * TSTR : 4 * BCH * BSTR + 64 * BSTR];
  __shared__ float sred[4];
  float S = 1.0f;
  if constexpr (PAIR) S = bwd_grad_scale(amax_part, n_part, sred);
  bool bad = false;
  float* Ks = smem;                   // [2][32][68]
  float* Vs = smem + 2 * BCH * BSTR;  // [2][32][68]  (PAIR: the planes start here)
  float* stage = PAIR ? smem : smem + 4 * BCH * BSTR;
  _Float16* KH = reinterpret_cast<_Float16*>(smem);
  _Float16* KL = KH + 2 * BCH * PSTR;
  _Float16* VH = KL + 2 * BCH * PSTR;
  _Float16* VL = VH + 2 * BCH * PSTR;
  _Float16* KTH = VL + 2 * BCH * PSTR;   // [2][64][32]
  _Float16* KTL = KTH + 2 * BHD * TSTR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, qi = lane & 15, g = lane >> 4;
  int fh, tile64;
  if (!xcd_group_decode(blockIdx.x, (N + 63) / 64, FH, fh, tile64)) return;  // tiles of one (frame, head) share an XCD
  const int f = fh / H, h = fh - f * H, D = H * BHD, D3 = 3 * D;
  const float* base = qkv + (long long)f * N * D3 + h * BHD;
  const int q0 = tile64 * 64, q = q0 + wave * 16 + qi;
  const bool wave_active = q0 + wave * 16 < N;

  float qreg[16], doreg[16], oreg[16];
  if constexpr (PAIR) {
    tile_to_regs_contig(base, D3, q0, N, stage, qreg, tid);
    tile_to_regs_contig(dout + (long long)f * N * D + h * BHD, D, q0, N, stage, doreg, tid);
    tile_to_regs_contig(out + (long long)f * N * D + h * BHD, D, q0, N, stage, oreg, tid);
  } else {
    tile_to_regs(base, D3, q0, N, stage, qreg, tid);
    tile_to_regs(dout + (long long)f * N * D + h * BHD, D, q0, N, stage, doreg, tid);
    tile_to_regs(out + (long long)f * N * D + h * BHD, D, q0, N, stage, oreg, tid);
  }
  float dl = 0.f;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    dl += doreg[s] * oreg[s];
    qreg[s] *= scale;
  }
  dl += __shfl_xor(dl, 16, 64);
  dl += __shfl_xor(dl, 32, 64);
  s16x4 qpk[4], dopk[4];   // (BF16) the query-side operands, rounded once
  f16x8 qh[2], ql[2], doh[2], dol[2];   // (PAIR) the same, split once; dO times S
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    qpk[t] = pack_bf16(qreg[4 * t], qreg[4 * t + 1], qreg[4 * t + 2], qreg[4 * t + 3]);
    dopk[t] = pack_bf16(doreg[4 * t], doreg[4 * t + 1], doreg[4 * t + 2], doreg[4 * t + 3]);
  }
  if constexpr (PAIR) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bad |= split8(qreg + 8 * t, 1.0f, qh[t], ql[t]);
      bad |= split8(doreg + 8 * t, S, doh[t], dol[t]);
    }
  }
  const float inv_s = 1.0f / S;   // (a power of two)
  const float S2 = S * 0.000244140625f;   // dS is split as dS S 2^-12: |dS| <= 2 x 64 max |dO| max |V|, i.e. dS S2 <= 512 max |V| - inside fp16 up to |V| = 128
  const float lse_q = (q < N) ? lse[((long long)f * H + h) * N + q] : 0.f;
  if (q < N && g == 0) delta[((long long)f * H + h) * N + q] = dl;

  float4 stk[2], stv[2];
  auto gload = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4;
      const int key = chunk * BCH + row;
      stk[i] = stv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (key < N) {
        stk[i] = *reinterpret_cast<const float4*>(base + (long long)key * D3 + D + c4);
        stv[i] = *reinterpret_cast<const float4*>(base + (long long)key * D3 + 2 * D + c4);
      }
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4;
      if constexpr (PAIR) {
        bad |= plane_write4(KH + (buf * BCH + row) * PSTR, KL + (buf * BCH + row) * PSTR, row, c4, stk[i], 1.0f);
        bad |= plane_write4(VH + (buf * BCH + row) * PSTR, VL + (buf * BCH + row) * PSTR, row, c4, stv[i], 1.0f);
        tplane_write4(KTH + buf * BHD * TSTR, KTL + buf * BHD * TSTR, row, c4, stk[i], 1.0f);
      } else {
        *reinterpret_cast<float4*>(Ks + (buf * BCH + row) * BSTR + c4) = stk[i];
        *reinterpret_cast<float4*>(Vs + (buf * BCH + row) * BSTR + c4) = stv[i];
      }
    }
  };

  f32x4 dq[4], dq2[4];   // (dq2: PAIR's cross-term accumulators)
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    dq[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dq2[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int nchunks = (N + BCH - 1) / BCH;
  gload(0);
  if constexpr (PAIR) __syncthreads();   // (the planes lie on the stage the prologue was still reading)
  swrite(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunks) gload(c + 1);
    if (wave_active) {
      float ds_all[8];   // (PAIR) dS of both 16-key tiles of the chunk: the K = 32 fragment of dQ^T += K^T dS^T
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const float* kp = Ks + (buf * BCH + 16 * t2 + qi) * BSTR + g;
        const float* vp = Vs + (buf * BCH + 16 * t2 + qi) * BSTR + g;
        f32x4 sa = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (PAIR) {
          f16x8 ah[2], al[2], bh[2], bl[2];
          plane_frags(KH + (buf * BCH + 16 * t2 + qi) * PSTR, qi, g, ah);
          plane_frags(KL + (buf * BCH + 16 * t2 + qi) * PSTR, qi, g, al);
          plane_frags(VH + (buf * BCH + 16 * t2 + qi) * PSTR, qi, g, bh);
          plane_frags(VL + (buf * BCH + 16 * t2 + qi) * PSTR, qi, g, bl);
          f32x4 sa2 = (f32x4){0.f, 0.f, 0.f, 0.f}, dp2 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            mma_pair8(ah[t], al[t], qh[t], ql[t], sa, sa2);
            mma_pair8(bh[t], bl[t], doh[t], dol[t], dp, dp2);
          }
          sa = fold_pair(sa, sa2, 1.0f);
          dp = fold_pair(dp, dp2, inv_s);
        } else if constexpr (BF16) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            sa = mma_bf16(pack_bf16(kp[16 * t], kp[16 * t + 4], kp[16 * t + 8], kp[16 * t + 12]), qpk[t], sa);
            dp = mma_bf16(pack_bf16(vp[16 * t], vp[16 * t + 4], vp[16 * t + 8], vp[16 * t + 12]), dopk[t], dp);
          }
        } else {
#pragma unroll
          for (int s = 0; s < 16; ++s) {
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(kp[4 * s], qreg[s], sa, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[4 * s], doreg[s], dp, 0, 0, 0);
          }
        }
        float ds[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int key = c * BCH + 16 * t2 + 4 * g + e;
          const float p = (key < N) ? fast_exp(sa[e] - lse_q) : 0.f;
          ds[e] = p * (dp[e] - dl);
          ds_all[4 * t2 + e] = ds[e];
        }
        if constexpr (PAIR) {
          // (dQ^T += K^T dS^T behind the second tile, below)
        } else if constexpr (BF16) {
          const float* kt = Ks + (buf * BCH + 16 * t2 + 4 * g) * BSTR + qi;
          const s16x4 dsp = pack_bf16(ds[0], ds[1], ds[2], ds[3]);
#pragma unroll
          for (int d = 0; d < 4; ++d)
            dq[d] = mma_bf16(pack_bf16(kt[16 * d], kt[BSTR + 16 * d], kt[2 * BSTR + 16 * d], kt[3 * BSTR + 16 * d]), dsp, dq[d]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float* kt = Ks + (buf * BCH + 16 * t2 + 4 * g + e) * BSTR + qi;
#pragma unroll
            for (int d = 0; d < 4; ++d) dq[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(kt[16 * d], ds[e], dq[d], 0, 0, 0);
          }
        }
      }
      if constexpr (PAIR) {   // dQ^T += K^T dS^T over the chunk's 32 keys: one K = 32 group per 16 head dimensions
        f16x8 dsh, dsl;
        bad |= split8(ds_all, S2, dsh, dsl);
#pragma unroll
        for (int d = 0; d < 4; ++d)
          mma_pair8(tplane_frag(KTH + buf * BHD * TSTR, 16 * d + qi, g), tplane_frag(KTL + buf * BHD * TSTR, 16 * d + qi, g), dsh, dsl, dq[d], dq2[d]);
      }
    }
    if (c + 1 < nchunks) swrite(buf ^ 1);
    __syncthreads();
  }
  float am = 0.f;
  if constexpr (PAIR) {
    scale /= S2;   // (a power of two: exact)
#pragma unroll
    for (int d = 0; d < 4; ++d) dq[d] = fold_pair(dq[d], dq2[d], 1.0f);
  }
  if (wave_active && q < N) {
    float* o = dqkv + ((long long)f * N + q) * D3 + h * BHD + 4 * g;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const float4 w = make_float4(dq[d][0] * scale, dq[d][1] * scale, dq[d][2] * scale, dq[d][3] * scale);
      *reinterpret_cast<float4*>(o + 16 * d) = w;
      am = fmaxf(fmaxf(am, fmaxf(fabsf(w.x), fabsf(w.y))), fmaxf(fabsf(w.z), fabsf(w.w)));
    }
  }
  if (amax_out) amax_publish(amax_out, am);   // (every lane of the wave is here: the max of what dqkv received, for the pair split of it)
  if constexpr (PAIR) range_flag_raise(range_flag, bad);
}

template <bool BF16, bool PAIR = false>
__global__ __launch_bounds__(256, PAIR ? 2 : 1) void attention_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                float* __restrict__ dqkv, float* __restrict__ amax_out, int N, int H, int FH, float scale,
                                                                const float* __restrict__ amax_part = nullptr, int n_part = 0, int* range_flag = nullptr) {
  // PAIR: four pair planes [2][32][64] fp16 (Q hi, Q lo, dO hi, dO lo - dO times S; the prologue's stage on top of them) | four transposed
  // planes [2][64][32] fp16 (Q^T hi, lo, dO^T hi, lo) | lse, delta
  __shared__ __attribute__((aligned(16))) float smem[(PAIR ? 4 * BCH * PSTR + 4 * BHD * TSTR : 4 * BCH * BSTR + 64 * BSTR) + 4 * BCH];
  __shared__ float sred[4];
  float S = 1.0f;
  if constexpr (PAIR) S = bwd_grad_scale(amax_part, n_part, sred);
  bool bad = false;
  float* Qs = smem;                    // [2][32][68]
  float* Os = smem + 2 * BCH * BSTR;   // dO chunks [2][32][68]
  float* stage = PAIR ? smem : smem + 4 * BCH * BSTR;
  float* Ls = PAIR ? smem + 4 * BCH * PSTR + 4 * BHD * TSTR : stage + 64 * BSTR;       // [2][32] lse, then [2][32] delta
  float* Dl = Ls + 2 * BCH;
  _Float16* QH = reinterpret_cast<_Float16*>(smem);
  _Float16* QL = QH + 2 * BCH * PSTR;
  _Float16* OH = QL + 2 * BCH * PSTR;
  _Float16* OL = OH + 2 * BCH * PSTR;
  _Float16* QTH = OL + 2 * BCH * PSTR;   // [2][64][32] each
  _Float16* QTL = QTH + 2 * BHD * TSTR;
  _Float16* OTH = QTL + 2 * BHD * TSTR;
  _Float16* OTL = OTH + 2 * BHD * TSTR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ki = lane & 15, g = lane >> 4;
  int fh, tile64;
  if (!xcd_group_decode(blockIdx.x, (N + 63) / 64, FH, fh, tile64)) return;
  const int f = fh / H, h = fh - f * H, D = H * BHD, D3 = 3 * D;
  const float* base = qkv + (long long)f * N * D3 + h * BHD;
  const float* dob = dout + (long long)f * N * D + h * BHD;
  const int k0 = tile64 * 64, key = k0 + wave * 16 + ki;
  const bool wave_active = k0 + wave * 16 < N;

  float kreg[16], vreg[16];
  if constexpr (PAIR) {
    tile_to_regs_contig(base + D, D3, k0, N, stage, kreg, tid);
    tile_to_regs_contig(base + 2 * D, D3, k0, N, stage, vreg, tid);
  } else {
    tile_to_regs(base + D, D3, k0, N, stage, kreg, tid);
    tile_to_regs(base + 2 * D, D3, k0, N, stage, vreg, tid);
  }
#pragma unroll
  for (int s = 0; s < 16; ++s) kreg[s] *= scale;
  s16x4 kpk[4], vpk[4];   // (BF16) the key-side operands, rounded once
  f16x8 kh[2], kl[2], vh[2], vl[2];   // (PAIR) the same, split once
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    kpk[t] = pack_bf16(kreg[4 * t], kreg[4 * t + 1], kreg[4 * t + 2], kreg[4 * t + 3]);
    vpk[t] = pack_bf16(vreg[4 * t], vreg[4 * t + 1], vreg[4 * t + 2], vreg[4 * t + 3]);
  }
  if constexpr (PAIR) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bad |= split8(kreg + 8 * t, 1.0f, kh[t], kl[t]);
      bad |= split8(vreg + 8 * t, 1.0f, vh[t], vl[t]);
    }
  }
  const float inv_s = 1.0f / S;   // (a power of two)
  const float S2 = S * 0.000244140625f;   // (dS is split as dS S 2^-12: the dq kernel)

  float4 stq[2], sto[2];
  float stl = 0.f, std_ = 0.f;
  auto gload = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4;
      const int q = chunk * BCH + row;
      stq[i] = sto[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < N) {
        stq[i] = *reinterpret_cast<const float4*>(base + (long long)q * D3 + c4);
        sto[i] = *reinterpret_cast<const float4*>(dob + (long long)q * D + c4);
      }
    }
    if (tid < BCH) {
      const int q = chunk * BCH + tid;
      stl = (q < N) ? lse[((long long)f * H + h) * N + q] : 0.f;
      std_ = (q < N) ? delta[((long long)f * H + h) * N + q] : 0.f;
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4;
      if constexpr (PAIR) {
        bad |= plane_write4(QH + (buf * BCH + row) * PSTR, QL + (buf * BCH + row) * PSTR, row, c4, stq[i], 1.0f);
        bad |= plane_write4(OH + (buf * BCH + row) * PSTR, OL + (buf * BCH + row) * PSTR, row, c4, sto[i], S);
        tplane_write4(QTH + buf * BHD * TSTR, QTL + buf * BHD * TSTR, row, c4, stq[i], 1.0f);
        tplane_write4(OTH + buf * BHD * TSTR, OTL + buf * BHD * TSTR, row, c4, sto[i], S);
      } else {
        *reinterpret_cast<float4*>(Qs + (buf * BCH + row) * BSTR + c4) = stq[i];
        *reinterpret_cast<float4*>(Os + (buf * BCH + row) * BSTR + c4) = sto[i];
      }
    }
    if (tid < BCH) {
      Ls[buf * BCH + tid] = stl;
      Dl[buf * BCH + tid] = std_;
    }
  };

  f32x4 dk[4], dv[4], dk2[4], dv2[4];   // (dk2 / dv2: PAIR's cross-term accumulators)
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    dk[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dv[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dk2[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dv2[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int nchunks = (N + BCH - 1) / BCH;
  gload(0);
  if constexpr (PAIR) __syncthreads();   // (the planes lie on the stage the prologue was still reading)
  swrite(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunks) gload(c + 1);
    if (wave_active) {
      float p_all[8], ds_all[8];   // (PAIR) P and dS of both 16-query tiles of the chunk: the K = 32 fragments of dV^T += dO^T P, dK^T += Q^T dS
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const float* qp = Qs + (buf * BCH + 16 * t2 + ki) * BSTR + g;
        const float* op = Os + (buf * BCH + 16 * t2 + ki) * BSTR + g;
        f32x4 sa = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (PAIR) {
          f16x8 ah[2], al[2], bh[2], bl[2];
          plane_frags(QH + (buf * BCH + 16 * t2 + ki) * PSTR, ki, g, ah);
          plane_frags(QL + (buf * BCH + 16 * t2 + ki) * PSTR, ki, g, al);
          plane_frags(OH + (buf * BCH + 16 * t2 + ki) * PSTR, ki, g, bh);
          plane_frags(OL + (buf * BCH + 16 * t2 + ki) * PSTR, ki, g, bl);
          f32x4 sa2 = (f32x4){0.f, 0.f, 0.f, 0.f}, dp2 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            mma_pair8(ah[t], al[t], kh[t], kl[t], sa, sa2);
            mma_pair8(bh[t], bl[t], vh[t], vl[t], dp, dp2);
          }
          sa = fold_pair(sa, sa2, 1.0f);
          dp = fold_pair(dp, dp2, inv_s);
        } else if constexpr (BF16) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            sa = mma_bf16(pack_bf16(qp[16 * t], qp[16 * t + 4], qp[16 * t + 8], qp[16 * t + 12]), kpk[t], sa);
            dp = mma_bf16(pack_bf16(op[16 * t], op[16 * t + 4], op[16 * t + 8], op[16 * t + 12]), vpk[t], dp);
          }
        } else {
#pragma unroll
          for (int s = 0; s < 16; ++s) {
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(qp[4 * s], kreg[s], sa, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x4f32(op[4 * s], vreg[s], dp, 0, 0, 0);
          }
        }
        float p[4], ds[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ql = 16 * t2 + 4 * g + e;
          const bool ok = (c * BCH + ql < N) && (key < N);
          p[e] = ok ? fast_exp(sa[e] - Ls[buf * BCH + ql]) : 0.f;
          ds[e] = p[e] * (dp[e] - Dl[buf * BCH + ql]);
          p_all[4 * t2 + e] = p[e];
          ds_all[4 * t2 + e] = ds[e];
        }
        if constexpr (PAIR) {
          // (the transposed products behind the second tile, below)
        } else if constexpr (BF16) {
          const float* ot = Os + (buf * BCH + 16 * t2 + 4 * g) * BSTR + ki;
          const float* qt = Qs + (buf * BCH + 16 * t2 + 4 * g) * BSTR + ki;
          const s16x4 pp = pack_bf16(p[0], p[1], p[2], p[3]), dsp = pack_bf16(ds[0], ds[1], ds[2], ds[3]);
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            dv[d] = mma_bf16(pack_bf16(ot[16 * d], ot[BSTR + 16 * d], ot[2 * BSTR + 16 * d], ot[3 * BSTR + 16 * d]), pp, dv[d]);
            dk[d] = mma_bf16(pack_bf16(qt[16 * d], qt[BSTR + 16 * d], qt[2 * BSTR + 16 * d], qt[3 * BSTR + 16 * d]), dsp, dk[d]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float* ot = Os + (buf * BCH + 16 * t2 + 4 * g + e) * BSTR + ki;
            const float* qt = Qs + (buf * BCH + 16 * t2 + 4 * g + e) * BSTR + ki;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              dv[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(ot[16 * d], p[e], dv[d], 0, 0, 0);
              dk[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(qt[16 * d], ds[e], dk[d], 0, 0, 0);
            }
          }
        }
      }
      if constexpr (PAIR) {   // dV^T += dO^T P, dK^T += Q^T dS over the chunk's 32 queries: one K = 32 group per 16 head dimensions each
        f16x8 ph, pl, dsh, dsl;
        split8(p_all, 1.0f, ph, pl);
        bad |= split8(ds_all, S2, dsh, dsl);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          mma_pair8(tplane_frag(OTH + buf * BHD * TSTR, 16 * d + ki, g), tplane_frag(OTL + buf * BHD * TSTR, 16 * d + ki, g), ph, pl, dv[d], dv2[d]);
          mma_pair8(tplane_frag(QTH + buf * BHD * TSTR, 16 * d + ki, g), tplane_frag(QTL + buf * BHD * TSTR, 16 * d + ki, g), dsh, dsl, dk[d], dk2[d]);
        }
      }
    }
    if (c + 1 < nchunks) swrite(buf ^ 1);
    __syncthreads();
  }
  float am = 0.f;
  float vscale = 1.0f;
  if constexpr (PAIR) {
    scale /= S2;      // (powers of two: exact)
    vscale = inv_s;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      dk[d] = fold_pair(dk[d], dk2[d], 1.0f);
      dv[d] = fold_pair(dv[d], dv2[d], 1.0f);
    }
  }
  if (wave_active && key < N) {
    float* ok_ = dqkv + ((long long)f * N + key) * D3 + D + h * BHD + 4 * g;
    float* ov = ok_ + D;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const float4 wk_ = make_float4(dk[d][0] * scale, dk[d][1] * scale, dk[d][2] * scale, dk[d][3] * scale);
      const float4 wv_ = make_float4(dv[d][0] * vscale, dv[d][1] * vscale, dv[d][2] * vscale, dv[d][3] * vscale);
      *reinterpret_cast<float4*>(ok_ + 16 * d) = wk_;
      *reinterpret_cast<float4*>(ov + 16 * d) = wv_;
      am = fmaxf(fmaxf(am, fmaxf(fabsf(wk_.x), fabsf(wk_.y))), fmaxf(fabsf(wk_.z), fabsf(wk_.w)));
      am = fmaxf(fmaxf(am, fmaxf(fabsf(wv_.x), fabsf(wv_.y))), fmaxf(fabsf(wv_.z), fabsf(wv_.w)));
    }
  }
  if (amax_out) amax_publish(amax_out, am);
  if constexpr (PAIR) range_flag_raise(range_flag, bad);
}

}  // namespace tt

using namespace tt;

extern "C" size_t tt_attention_bwd_workspace_bytes(int F, int N, int H, int hd) {
  (void)hd;
  return (size_t)F * H * N * sizeof(float);
}

template <bool BF16>
static int attention_bwd_impl(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int F, int N,
                              int H, int hd, float scale, void* workspace, size_t workspace_bytes, float* amax_out, tt_stream_t stream) {
  TT_REQUIRE(qkv && out && dout && lse && dqkv && workspace, "attention_bwd: null pointer");
  TT_REQUIRE(hd == 64, "attention_bwd: head_dim must be 64 (got %d)", hd);
  TT_REQUIRE(F > 0 && H > 0 && N > 0, "attention_bwd: bad shape");
  TT_REQUIRE(workspace_bytes >= tt_attention_bwd_workspace_bytes(F, N, H, hd), "attention_bwd: workspace too small");
  TT_REQUIRE(aligned16(qkv) && aligned16(out) && aligned16(dout) && aligned16(dqkv), "attention_bwd: buffers must be 16-byte aligned");
  hipStream_t s = as_stream(stream);
  float* delta = static_cast<float*>(workspace);
  dim3 grid(xcd_group_grid(F * H, (N + 63) / 64));
  hipLaunchKernelGGL(attention_bwd_dq_kernel<BF16>, grid, dim3(256), 0, s, qkv, out, dout, lse, dqkv, delta, amax_out, N, H, F * H, scale);
  hipLaunchKernelGGL(attention_bwd_dkv_kernel<BF16>, grid, dim3(256), 0, s, qkv, dout, lse, delta, dqkv, amax_out, N, H, F * H, scale);
  TT_CHECK_LAUNCH("attention_bwd");
  return TT_OK;
}

extern "C" int tt_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int F, int N,
                                int H, int hd, float scale, void* workspace, size_t workspace_bytes, float* amax_out, tt_stream_t stream) {
  return attention_bwd_impl<false>(qkv, out, dout, lse, dqkv, F, N, H, hd, scale, workspace, workspace_bytes, amax_out, stream);
}

extern "C" size_t tt_attention_bwd_pairs_workspace_bytes(int F, int N, int H, int hd) {
  return tt_attention_bwd_workspace_bytes(F, N, H, hd) + kBwdAmaxParts * sizeof(float);
}

extern "C" int tt_attention_bwd_pairs(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int F, int N, int H, int hd,
                                      float scale, const float* dout_amax, void* workspace, size_t workspace_bytes, int* range_flag, float* amax_out,
                                      tt_stream_t stream) {
  TT_REQUIRE(qkv && out && dout && lse && dqkv && workspace, "attention_bwd_pairs: null pointer");
  TT_REQUIRE(hd == 64, "attention_bwd_pairs: head_dim must be 64 (got %d)", hd);
  TT_REQUIRE(F > 0 && H > 0 && N > 0, "attention_bwd_pairs: bad shape");
  TT_REQUIRE(workspace_bytes >= tt_attention_bwd_pairs_workspace_bytes(F, N, H, hd), "attention_bwd_pairs: workspace too small");
  TT_REQUIRE(aligned16(qkv) && aligned16(out) && aligned16(dout) && aligned16(dqkv) && aligned16(workspace), "attention_bwd_pairs: buffers must be 16-byte aligned");
  hipStream_t s = as_stream(stream);
  float* delta = static_cast<float*>(workspace);
  const float* part = delta + (size_t)F * H * N;
  const long long n = (long long)F * N * H * hd;
  int n_part = (int)((n + 4095) / 4096 < kBwdAmaxParts ? (n + 4095) / 4096 : kBwdAmaxParts);
  if (dout_amax) {   // the kernel that wrote dout left max |dout| in its amax slot: no pass
    part = const_cast<float*>(dout_amax);
    n_part = -kAmaxWays;
  } else {
    hipLaunchKernelGGL(attention_bwd_amax_kernel, dim3(n_part), dim3(256), 0, s, dout, n, const_cast<float*>(part));
  }
  dim3 grid(xcd_group_grid(F * H, (N + 63) / 64));
  hipLaunchKernelGGL((attention_bwd_dq_kernel<false, true>), grid, dim3(256), 0, s, qkv, out, dout, lse, dqkv, delta, amax_out, N, H, F * H, scale, part, n_part,
                     range_flag);
  hipLaunchKernelGGL((attention_bwd_dkv_kernel<false, true>), grid, dim3(256), 0, s, qkv, dout, lse, delta, dqkv, amax_out, N, H, F * H, scale, part, n_part,
                     range_flag);
  TT_CHECK_LAUNCH("attention_bwd_pairs");
  return TT_OK;
}

extern "C" int tt_attention_bwd_bf16(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int F, int N,
                                     int H, int hd, float scale, void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  return attention_bwd_impl<true>(qkv, out, dout, lse, dqkv, F, N, H, hd, scale, workspace, workspace_bytes, nullptr, stream);
}
