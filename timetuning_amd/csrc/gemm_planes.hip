// bf16-plane GEMM for the forward nn.Linear sites (dino_vision_transformer.py:94-103,115-130; models.py:915-926):
//     y = act(x @ w^T + bias) (+ residual)
// with BOTH operands already resident in HBM as P planes of bf16 ("pre-split" by whoever produced them):
//
//   P = 1   plain bf16 operands: BASELINE config C4's "MFMA bf16 path" - bf16 activations and weights in HBM, fp32 accumulate.
//   P = 2   x = x1 + x2 (16 significant bits), products x1w1 + x1w2 + x2w1.
//   P = 3   x = x1 + x2 + x3 EXACTLY (3 x 8 = 24 significant bits = fp32), the six products with i + j <= 4; the dropped
//           terms are <= 2^-24 relative, i.e. fp32-level accuracy at 6 bf16 MFMAs per product term (nominal peak = bf16 / 6).
//
// Unlike gemm_nt_bf16.hip (fp32 in HBM, converted while staged: HBM- and VALU-bound) nothing is converted here: a slab goes
// HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, no ds_write) and a fragment is one ds_read_b128.
//
// LDS image per plane and operand: [row][BK bf16] rows of BK*2 bytes, 16-byte chunks XOR-swizzled by the row so that the
// 16 lanes a ds_read_b128 services together (lanes {0-3,12-15,20-27}, ... = rows with distinct (row / W) & (C-1), W = rows
// per 256 B, C = chunks per row) hit 16 distinct 16-byte slots of the 64 banks.  The LDS-DMA destination is lane-linear (M0
// base + 16 * lane), so the swizzle is applied to the per-lane SOURCE address: LDS slot s of row r receives chunk s ^ f(r).
// v_mfma_f32_32x32x16_bf16: lane l (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j], j = 0..7 = chunk 2 ks + h of its row.
//
// Where the time goes (round-2 ablations on ViT-B/16's qkv product, 25216 x 2304 x 768, 128 x 128 tiles, interleaved A/B of
// tools/ab_planes.py + build_variant.sh): full kernel 122 us; epilogue removed 91; MFMAs and fragment reads removed (DMA +
// barriers + epilogue only) 99 - i.e. the 1.4 GB of L2 -> LDS operand traffic of a 128 x 128 tiling costs as much as the matrix
// work and the two overlap poorly; the epilogue is a quarter of the launch (a third at N = 3072 with GELU + bf16 stores).
// Ring depth (2 / 3 / 4 slabs), slab width (32 / 64 k), 2 or 3 workgroups per CU and loader / compute wave specialisation
// (LD = 1) all land within +-5 %; halving the traffic with 256-wide tiles (8 waves) is worth 6-16 % where the grid allows.
// Four waves with 128 x 128 (P = 1) / 128 x 64 (P = 3) wave tiles - half the LDS fragment traffic per MFMA, 256 accumulator
// registers, one wave per SIMD - are 10-25 % SLOWER at P = 1 and equal at P = 3 (tools/bench_planes.py): with one wave per SIMD
// nothing hides the fragment reads.  Register staging (16-byte global loads before the MFMAs of a slab, ds_write_b128 after them, same
// LDS image) instead of LDS-DMA: equal to 10 % slower on every shape and tiling; fragment reads of k-step ks + 1 issued before the
// MFMAs of k-step ks (two register sets, scheduler fenced): +-2 %; 3- and 4-slab rings on the 256 x 256 tile, 128 x 256 / 256 x 128
// tiles with two 4-wave workgroups per CU, a 32 KB-per-workgroup 128 x 128 tile at 4 workgroups per CU: all within +-5 % or slower
// (PMC reading of the 256 x 256 instance: profiles/r02_gemm_planes_pmc.json).  For scale: a bare v_mfma_f32_32x32x16_bf16 loop on random operands sustains 1.81 PFLOP/s on
// this part (2.39 on zeros; tools/mfma_bf16_shapes.hip, profiles/r02_mfma_bf16_shapes.txt) - this kernel reaches 0.33-0.48 of that.
//
// Tile (32 WM GM) x (32 WN GN), GM x GN waves (4 as 2 x 2 by default), LDS ring of NBUF slabs, ONE barrier per slab: the DMA of slab t+1 is issued before
// the MFMAs of slab t and retired by the barrier's vmcnt(0).  Rows beyond M are clamped on load and masked on store.
// Epilogue through LDS (8 columns per thread): bias, GELU (erf to 1.5e-7, common.hpp gelu_fast_f), residual, and any of: fp32 y, fp32 pre-activation, bf16 planes
// of y (so the consumer finds its operand pre-split).
#include "common.hpp"
#include <cstdlib>

namespace tt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct PlaneArgs {
  const __bf16* A;        // [P][M][K]
  const __bf16* B;        // [P][N][K]
  long long a_stride, b_stride;  // plane strides in elements
  int M, N, K;
  const float* bias;      // [N] or null
  const float* residual;  // [M][N] fp32 or null (may alias C)
  float* C;               // [M][N] fp32 or null
  float* pre_out;         // [M][N] fp32 or null
  __bf16* Cp;             // [PO][M][N] bf16 planes or null
  long long c_stride;
  int po;                 // number of output planes (0..3)
  int act;                // 1 = GELU
  const float* gelu_pre;  // backward-data use: y *= gelu'(gelu_pre[m][n]) (dino_vision_transformer.py:100 through autograd) or null
  int splits;             // split-K: grid.y slices of the reduction, slice z writes plain fp32 partials to C + z * split_stride
  long long split_stride;
  const float* out_scale; // PAIR: device scalar S (a power of two) by which an operand was scaled before its split (a gradient: see
                          // transpose_pairs_tile) - the product is divided by it; null: none
  int* range_flag;        // PAIR outputs: device word set to 1 when an output's hi leaves fp16's range (common.hpp pair_hi_bad); null: none
  float* amax_out;        // device float or null: max |C| is published into it (common.hpp amax_publish: a data gradient that is the next
                          // Linear's dy leaves its maximum for that dy's pair split)
};

// x -> up to three bf16 planes with x = p0 + p1 + p2 (exact when 3 planes are taken and no exponent underflow)
__device__ __forceinline__ void split3(float v, __bf16& p0, __bf16& p1, __bf16& p2) {
  p0 = (__bf16)v;
  const float r1 = v - (float)p0;
  p1 = (__bf16)r1;
  const float r2 = r1 - (float)p1;
  p2 = (__bf16)r2;
}

// s_waitcnt vmcnt(N) with a compile-time N (the LDS-DMA pieces this wave may leave in flight)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// GM x GN waves (default 2 x 2), each owning a (32 WM) x (32 WN) sub-tile: block tile (32 WM GM) x (32 WN GN).  The 8-wave
// 256 x 256 / 256 x 128 instances exist because the 4-wave 128 x 128 tile is bound by the L2 -> LDS operand traffic, not by the
// matrix pipe (ablation, tools/ab_planes.py: with the MFMAs and fragment reads REMOVED the ViT-B/16 qkv launch still takes 99 of
// its 122 us); traffic per flop falls as 1/BM + 1/BN.
// LD = 1 (wave specialisation): GM x GN MORE waves join the workgroup as LOADERS - they issue every LDS-DMA piece and wait for
// it, the compute waves issue only ds_read_b128 + MFMA.  A DMA piece costs its issuing wave 100+ cycles (address VALU, M0, the
// request itself) - as long as the MFMAs of the k-step it feeds when every wave does both; on its own wave it overlaps the
// other wave's matrix work instead of delaying it (one loader and one compute wave per SIMD at GM x GN = 2 x 2).
// PAIR (round 4): the operands are fp16 PAIRS (common.hpp split_pair; gemm_pairs8.hip) - P = 1, BK = 64: a "row" of 64 16-bit elements is
// one pair group [hi x 32][lo x 32] of a 32-deep K-tile, g.K counts 16-bit elements (2 x the reduction length), and a term costs three
// v_mfma_f32_32x32x16_f16 into two accumulator sets (hi hi | hi lo + lo hi, folded with 2^-11 in the epilogue).  The general-shape
// kernel of the "f16x3" mode: everything the persistent gemm_pairs8_kernel does not take (any M, N % 64 == 0, the projection head,
// pre-activation outputs, gelu' products, split-K).
template <int P, int BK, int WM, int WN, int NBUF, int GM = 2, int GN = 2, int LD = 0, bool PAIR = false>
__global__ __launch_bounds__(64 * GM * GN * (1 + LD)) void gemm_planes_kernel(PlaneArgs g) {
  static_assert(!PAIR || (P == 1 && BK == 64), "pair operands: one 128-byte row per 32-deep K-tile");
  constexpr int NC = GM * GN;                 // compute waves
  constexpr int NW = LD ? NC : NC;            // waves that share the DMA pieces (the loaders when LD, else everybody)
  constexpr int NT = 64 * NC * (1 + LD);      // threads
  constexpr int BM = 32 * WM * GM, BN = 32 * WN * GN;
  constexpr int ROWB = BK * 2;              // bytes per LDS row
  constexpr int CPR = ROWB / 16;            // 16-byte chunks per row
  constexpr int WIN = 256 / ROWB;           // rows per 256-byte bank window
  constexpr int RPI = 64 / CPR;             // rows one LDS-DMA wave-instruction fills
  constexpr int A_PL = BM * ROWB, B_PL = BN * ROWB;
  constexpr int BUF = P * (A_PL + B_PL);
  constexpr int CH = 32 * WM, LDCS = BN + 4;
  constexpr int EPI = CH * LDCS * 4;
  constexpr int LDS_BYTES = NBUF * BUF > EPI ? NBUF * BUF : EPI;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  static_assert(BM % RPI == 0 && BN % RPI == 0, "whole DMA pieces");
  // DMA pieces (wave-instructions) every wave issues per slab; the counted waits below need the same number in all waves
  constexpr int G = P * (BM / RPI / NW + BN / RPI / NW);
  static_assert(NBUF == 2 || ((BM / RPI) % NW == 0 && (BN / RPI) % NW == 0), "ring depths > 2 need equal piece counts per wave");
  static_assert((NBUF - 2) * G < 64, "vmcnt range");
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave_id = tid >> 6;
  const bool loader = LD ? wave_id >= NC : true, computes = LD ? wave_id < NC : true;
  const int wave = LD && wave_id >= NC ? wave_id - NC : wave_id;   // index among the loaders / among the compute waves
  const int wm = wave / GN, wn = wave % GN, r = lane & 31, h = lane >> 5;
  const int ntn = g.N / BN, ntm = (g.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  const int K = g.K;
  // split-K (weight gradients: few output tiles, long reduction): this workgroup owns slabs [kt_lo, kt_lo + nk)
  const int nk_all = K / BK;
  const int per = (nk_all + g.splits - 1) / g.splits;
  const int kt_lo = blockIdx.y * per;
  const int nk = kt_lo + per <= nk_all ? per : (nk_all > kt_lo ? nk_all - kt_lo : 0);

  // ---- LDS-DMA: piece i (RPI rows) of an operand plane; lane -> (row, slot), source chunk = slot ^ f(row)
  const int l_row = lane / CPR, l_slot = lane % CPR;
  auto issue = [&](int kt, int buf) {
    unsigned char* base = smem + buf * BUF;
#pragma unroll
    for (int p = 0; p < P; ++p) {
#pragma unroll
      for (int i = 0; i < (BM / RPI + NW - 1) / NW; ++i) {
        const int piece = wave + NW * i, row = piece * RPI + l_row;
        if ((BM / RPI) % NW != 0 && piece >= BM / RPI) break;   // wave-uniform
        const int chunk = l_slot ^ ((row / WIN) & (CPR - 1));
        int grow = m0 + row;
        grow = grow < g.M ? grow : g.M - 1;
        const __bf16* src = g.A + p * g.a_stride + (size_t)grow * K + (kt_lo + kt) * BK + chunk * 8;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                         (void __attribute__((address_space(3)))*)(base + p * A_PL + piece * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < (BN / RPI + NW - 1) / NW; ++i) {
        const int piece = wave + NW * i, row = piece * RPI + l_row;
        if ((BN / RPI) % NW != 0 && piece >= BN / RPI) break;   // wave-uniform
        const int chunk = l_slot ^ ((row / WIN) & (CPR - 1));
        const __bf16* src = g.B + p * g.b_stride + (size_t)(n0 + row) * K + (kt_lo + kt) * BK + chunk * 8;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                         (void __attribute__((address_space(3)))*)(base + P * A_PL + p * B_PL + piece * 1024), 16, 0, 0);
      }
    }
  };

  f32x16 acc[WM][WN];
  f32x16 acc2[PAIR ? WM : 1][PAIR ? WN : 1];   // PAIR: the cross terms hi lo + lo hi (x 2^11)
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc[i][j][e] = 0.f;
        if constexpr (PAIR) acc2[i][j][e] = 0.f;
      }

  // fragment addressing: row offsets are compile-time multiples of 32 rows, the swizzle term depends on (row / WIN) only
  int a_off[WM], b_off[WN], a_sw[WM], b_sw[WN];
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int row = wm * (32 * WM) + i * 32 + r;
    a_off[i] = row * ROWB;
    a_sw[i] = (row / WIN) & (CPR - 1);
  }
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int row = wn * (32 * WN) + j * 32 + r;
    b_off[j] = row * ROWB;
    b_sw[j] = (row / WIN) & (CPR - 1);
  }

  auto compute = [&](int buf) {
#ifdef TT_PLANES_NO_MAINLOOP   // timing study only: DMA and barriers stay, no fragment reads / MFMAs
    return;
#endif
    const unsigned char* base = smem + buf * BUF;
    if constexpr (PAIR) {
      typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int ch = 2 * ks + h, cl = 4 + 2 * ks + h;   // 16-byte chunks of the row: hi / lo of this k-step
        f16x8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          ah[i] = *reinterpret_cast<const f16x8*>(base + a_off[i] + ((ch ^ a_sw[i]) << 4));
          al[i] = *reinterpret_cast<const f16x8*>(base + a_off[i] + ((cl ^ a_sw[i]) << 4));
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          bh[j] = *reinterpret_cast<const f16x8*>(base + A_PL + b_off[j] + ((ch ^ b_sw[j]) << 4));
          bl[j] = *reinterpret_cast<const f16x8*>(base + A_PL + b_off[j] + ((cl ^ b_sw[j]) << 4));
        }
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc2[i][j], 0, 0, 0);
            acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc2[i][j], 0, 0, 0);
          }
      }
      return;
    }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int chunk = 2 * ks + h;
      bf16x8 a[P][WM], b[P][WN];
#pragma unroll
      for (int p = 0; p < P; ++p) {
#pragma unroll
        for (int i = 0; i < WM; ++i) a[p][i] = *reinterpret_cast<const bf16x8*>(base + p * A_PL + a_off[i] + ((chunk ^ a_sw[i]) << 4));
#pragma unroll
        for (int j = 0; j < WN; ++j)
          b[p][j] = *reinterpret_cast<const bf16x8*>(base + P * A_PL + p * B_PL + b_off[j] + ((chunk ^ b_sw[j]) << 4));
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int s = P - 1; s >= 0; --s)        // plane-index sum s = pa + pw: small terms first
#pragma unroll
            for (int pa = 0; pa <= s; ++pa)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa][i], b[s - pa][j], acc[i][j], 0, 0, 0);
    }
  };
  if constexpr (NBUF == 2) {
    // double buffer, one barrier per slab: the DMA of slab t+1 flies under the MFMAs of slab t and is retired by the barrier's
    // vmcnt(0)
    if (nk > 0 && loader) issue(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk && loader) issue(kt + 1, buf ^ 1);
      if (computes) compute(buf);
      __syncthreads();
    }
  } else {
    // ring of NBUF slabs, NBUF - 1 of them in flight: per slab ONE raw s_barrier behind a COUNTED vmcnt that leaves the younger
    // slabs' DMA pieces outstanding (a __syncthreads() would drain them: its fence waits vmcnt(0)).  Passing the barrier of
    // iteration kt means (a) every wave's pieces of slab kt have landed and (b) every wave is done reading slab kt - 1, whose
    // buffer the DMA issued right behind the barrier overwrites.
#pragma unroll
    for (int t = 0; t < NBUF - 1; ++t)
      if (t < nk && loader) issue(t, t);
    int buf = 0, nxt = NBUF - 1;
    for (int kt = 0; kt < nk; ++kt) {
      const int younger = nk - 1 - kt;   // slabs issued after kt that may stay in flight
      if (loader) {
        if (younger >= NBUF - 2) wait_vmcnt<(NBUF - 2) * G>();
        else if (NBUF > 3 && younger == 1) wait_vmcnt<G>();
        else wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      if (kt + NBUF - 1 < nk && loader) issue(kt + NBUF - 1, nxt);
      if (computes) compute(buf);
      buf = buf + 1 == NBUF ? 0 : buf + 1;
      nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
    }
    __builtin_amdgcn_s_barrier();   // all reads of the last slab done before the epilogue reuses the LDS
  }

#ifdef TT_PLANES_NO_EPILOGUE   // timing study only (tools/build_variant.sh): keep the accumulators alive with one store per lane
  {
    float sres = 0.f;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) sres += acc[i][j][e];
    if (g.C && sres == 12345.678f) g.C[tid] = sres;
    return;
  }
#endif
  // ---- epilogue through LDS, one wave-row (32 WM tile rows) at a time; a thread owns 8 consecutive columns
  float* Cs = reinterpret_cast<float*>(smem);
  constexpr int TPR = BN / 8, RPP = NT / TPR;
  const int c8 = (tid % TPR) * 8;
  const int n = n0 + c8;
  float bias8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = g.bias ? g.bias[n + e] : 0.f;
  const float inv_s = (PAIR && g.out_scale) ? 1.0f / *g.out_scale : 1.0f;   // exact: S is a power of two
  float am = 0.f;
#pragma unroll
  for (int wmi = 0; wmi < GM; ++wmi) {
    if (computes && wm == wmi) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            Cs[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * LDCS + wn * (32 * WN) + j * 32 + r] =
                PAIR ? fmaf(acc2[PAIR ? i : 0][PAIR ? j : 0][e], kPairInvScale, acc[i][j][e]) * inv_s : acc[i][j][e];
    }
    __syncthreads();
    for (int rr = tid / TPR; rr < CH; rr += RPP) {
      const int m = m0 + wmi * CH + rr;
      if (m < g.M) {
        const size_t off = (size_t)m * g.N + n;
        float v[8];
        *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(Cs + rr * LDCS + c8);
        *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(Cs + rr * LDCS + c8 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bias8[e];
        if (g.pre_out) {
          *reinterpret_cast<float4*>(g.pre_out + off) = *reinterpret_cast<const float4*>(v);
          *reinterpret_cast<float4*>(g.pre_out + off + 4) = *reinterpret_cast<const float4*>(v + 4);
        }
        if (g.act == 1) {
          // ONE GELU formula per route of the "bf16" mode: a bf16-only output (P = 1, no fp32 y, no pre-activation - the
          // frozen blocks' fc1) takes the tanh form here exactly as gemm_planes8_kernel<1> does, so the numbers do not depend on which of
          // the two kernels the tile count selects; every fp32-accurate output keeps the erf form.
          const bool tanh_form = P == 1 && !PAIR && g.C == nullptr && g.pre_out == nullptr;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = tanh_form ? gelu_bf16_f(v[e]) : gelu_fast_f(v[e]);
        }
        if (g.gelu_pre) {
          float pr[8];
          *reinterpret_cast<float4*>(pr) = *reinterpret_cast<const float4*>(g.gelu_pre + off);
          *reinterpret_cast<float4*>(pr + 4) = *reinterpret_cast<const float4*>(g.gelu_pre + off + 4);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= gelu_grad_fast_f(pr[e]);
        }
        if (g.residual) {
          float rs[8];
          *reinterpret_cast<float4*>(rs) = *reinterpret_cast<const float4*>(g.residual + off);
          *reinterpret_cast<float4*>(rs + 4) = *reinterpret_cast<const float4*>(g.residual + off + 4);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rs[e];
        }
        if (g.C) {
          float* c = g.C + (size_t)blockIdx.y * g.split_stride + off;
          *reinterpret_cast<float4*>(c) = *reinterpret_cast<const float4*>(v);
          *reinterpret_cast<float4*>(c + 4) = *reinterpret_cast<const float4*>(v + 4);
          if (g.amax_out) {
#pragma unroll
            for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(v[e]));
          }
        }
        if constexpr (PAIR) {
          if (g.po > 0) {   // pairs [M][2 N]: this thread's 8 columns lie in one group of 32
            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
            f16x8 qh, ql;
            bool bad = false;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              _Float16 hi_, lo_;
              split_pair(v[e], hi_, lo_);
              qh[e] = hi_;
              ql[e] = lo_;
              bad |= pair_hi_bad(hi_);
            }
            _Float16* dst = reinterpret_cast<_Float16*>(g.Cp) + (size_t)m * 2 * g.N + pair_index(n);
            *reinterpret_cast<f16x8*>(dst) = qh;
            *reinterpret_cast<f16x8*>(dst + 32) = ql;
            range_flag_raise(g.range_flag, bad);
          }
        } else if (g.po > 0) {
          bf16x8 q0, q1, q2;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            __bf16 p0, p1, p2;
            split3(v[e], p0, p1, p2);
            q0[e] = p0; q1[e] = p1; q2[e] = p2;
          }
          *reinterpret_cast<bf16x8*>(g.Cp + off) = q0;
          if (g.po > 1) *reinterpret_cast<bf16x8*>(g.Cp + g.c_stride + off) = q1;
          if (g.po > 2) *reinterpret_cast<bf16x8*>(g.Cp + 2 * g.c_stride + off) = q2;
        }
      }
    }
    __syncthreads();
  }
  if (g.amax_out) amax_publish(g.amax_out, am);   // (uniform; every lane is here)
}

template <int P, int BK, int WM, int WN, int NBUF = 2, int GM = 2, int GN = 2, int LD = 0, bool PAIR = false>
static int launch_planes(const PlaneArgs& g, hipStream_t s) {
  constexpr int BM = 32 * WM * GM, BN = 32 * WN * GN;
  const int tiles = ((g.M + BM - 1) / BM) * (g.N / BN);
  hipLaunchKernelGGL((gemm_planes_kernel<P, BK, WM, WN, NBUF, GM, GN, LD, PAIR>), dim3(tiles, g.splits), dim3(64 * GM * GN * (1 + LD)), 0, s, g);
  TT_CHECK_LAUNCH("gemm_planes");
  return TT_OK;
}

// ---- f32 -> bf16 planes (weights once per change, activations whose producer is an fp32 kernel)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long long stride, int planes,
                                                           long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    float v[8];
    *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(src + 8 * i);
    *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(src + 8 * i + 4);
    bf16x8 q0, q1, q2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      __bf16 p0, p1, p2;
      split3(v[e], p0, p1, p2);
      q0[e] = p0; q1[e] = p1; q2[e] = p2;
    }
    *reinterpret_cast<bf16x8*>(dst + 8 * i) = q0;
    if (planes > 1) *reinterpret_cast<bf16x8*>(dst + stride + 8 * i) = q1;
    if (planes > 2) *reinterpret_cast<bf16x8*>(dst + 2 * stride + 8 * i) = q2;
  }
}

// ---- fp32 [R][C] -> bf16 [C][Rpad] (transpose; columns R..Rpad-1 zero): the operands of the backward products in the layout
// gemm_planes_kernel reads (reduction index contiguous).  64 x 64 tiles through LDS.
// SUM: the fp32 column sums of the tile's 64 rows go to partial[blockIdx.x][c] as well (a bias gradient = column sums of dy, whose
// transpose the weight-gradient product needs anyway: one pass over dy instead of two; folded by colsum_fold in a fixed order)
template <bool SUM>
__global__ __launch_bounds__(256) void transpose_planes_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int R, int C, int Rpad,
                                                               float* __restrict__ partial) {
  __shared__ float t[64][65];
  __shared__ float red[4][64];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  float csum = 0.f;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    const float v = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
    t[i][tx] = v;
    csum += v;
  }
  if constexpr (SUM) red[ty][tx] = csum;
  __syncthreads();
  if constexpr (SUM) {
    if (ty == 0 && c0 + tx < C) partial[(size_t)blockIdx.x * C + c0 + tx] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
  }
  // out row c = 64 consecutive r: 16 lanes x 4 bf16 (8-byte stores; Rpad % 64 == 0 keeps them aligned), 16 rows per pass
  if (Rpad % 4 == 0) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const int q = threadIdx.x & 15, cr = threadIdx.x >> 4;
    for (int i = cr; i < 64; i += 16) {
      const int c = c0 + i, r = r0 + 4 * q;
      if (c < C && r < Rpad) {   // (Rpad % 4 == 0: the four r of a quad are inside or outside together)
        bf16x4 v = {(__bf16)t[4 * q][i], (__bf16)t[4 * q + 1][i], (__bf16)t[4 * q + 2][i], (__bf16)t[4 * q + 3][i]};
        *reinterpret_cast<bf16x4*>(dst + (size_t)c * Rpad + r) = v;
      }
    }
    return;
  }
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < Rpad) dst[(size_t)c * Rpad + r] = (__bf16)t[tx][i];
  }
}

// ---- f32 <-> fp16 pairs (common.hpp split_pair): groups of 32 elements as [hi x 32][lo x 32]; a thread converts 8 elements
__global__ __launch_bounds__(256) void split_pairs_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, long long n8, int* range_flag) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  bool bad = false;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    float v[8];
    *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(src + 8 * i);
    *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(src + 8 * i + 4);
    f16x8 qh, ql;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      _Float16 hi_, lo_;
      split_pair(v[e], hi_, lo_);
      qh[e] = hi_;
      ql[e] = lo_;
      bad |= pair_hi_bad(hi_);
    }
    _Float16* d = dst + pair_index(8 * i);
    *reinterpret_cast<f16x8*>(d) = qh;
    *reinterpret_cast<f16x8*>(d + 32) = ql;
  }
  range_flag_raise(range_flag, bad);
}
__global__ __launch_bounds__(256) void join_pairs_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, long long n8) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const _Float16* s = src + pair_index(8 * i);
    const f16x8 qh = *reinterpret_cast<const f16x8*>(s), ql = *reinterpret_cast<const f16x8*>(s + 32);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = join_pair(qh[e], ql[e]);
    *reinterpret_cast<float4*>(dst + 8 * i) = *reinterpret_cast<const float4*>(v);
    *reinterpret_cast<float4*>(dst + 8 * i + 4) = *reinterpret_cast<const float4*>(v + 4);
  }
}

int pairs8_try(const void* x_pairs, const void* w_pairs, const float* bias, const float* residual, float* y, float* pre_out, void* y_pairs,
               const float* gelu_pre, const float* out_scale, int M, int N, int K, int act, void* ksplit_ws, size_t ksplit_ws_bytes_, int* range_flag,
               float* amax_out, hipStream_t s);                                                                       // gemm_pairs8.hip
int pairs8_would_run(int M, int N, int K, int act, int has_residual, int has_y, int has_pairs, int has_pre, int has_gelu_pre);

// ---- transposed pairs: the operands of the backward products in the "f16x3" mode (reduction index contiguous, in pair groups).
// 64 x 64 tiles through LDS.  out row c holds groups of 32 consecutive r as [hi x 32][lo x 32]; rows R..Rpad-1 are zero (Rpad % 32 == 0).
//   SRC_PAIRS = false: src fp32 [R][C]; optionally ALSO the row-major pairs [R][2 C] (C % 32 == 0: the dgrad operand of a dy that the
//   weight gradient needs transposed) and the fp32 column sums of the tile's rows (a bias gradient) - one read of dy for all three.
//   SRC_PAIRS = true: src pairs [R][2 C] (a saved forward operand): a 16-bit transpose of the hi and the lo halves.
// amax -> the power of two S that brings it into [2^13, 2^14) (fp16: max 65504, smallest normal 2^-14); 1 for 0 / inf / nan
__device__ __forceinline__ float pair_scale_of(float amax) {
  if (!(amax > 0.f) || !(amax < INFINITY)) return 1.0f;
  int e = 13 - ilogbf(amax);
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return ldexpf(1.0f, e);
}
// per-workgroup maxima of |x| (stage 1 of the gradient scale: every consumer folds the <= 256 partials itself - max is order-independent)
__global__ __launch_bounds__(256) void amax_partial_kernel(const float* __restrict__ x, long long n, float* __restrict__ part) {
  __shared__ float red[4];
  float m = 0.f;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * 1024) {
    if (i + 3 < n) {
      const float4 v = *reinterpret_cast<const float4*>(x + i);
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    } else {
      for (long long j = i; j < n; ++j) m = fmaxf(m, fabsf(x[j]));
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// amax_part / n_part (fp32 source only): the source is a GRADIENT whose whole magnitude may sit below fp16's normal range (a C2 step's
// dy tensors peak at 1e-3 .. 1e-6 with medians down to 2e-8: split as they are, hi is an fp16 subnormal and the pair keeps ~15 bits) -
// it is multiplied by the power of two S = pair_scale_of(max |src|) before the split (exact) and the consumers divide their product by S
// (PlaneArgs::out_scale).  The column sums are taken of the unscaled values.  Workgroup (0, 0) publishes S.
template <bool SRC_PAIRS, bool ROW, bool SUM>
__device__ __forceinline__ void transpose_pairs_tile(const void* __restrict__ src_, _Float16* __restrict__ dst_t, _Float16* __restrict__ dst_row,
                                                     int R, int C, int Rpad, float* __restrict__ partial, int bx, int by,
                                                     const float* __restrict__ amax_part = nullptr, int n_part = 0,
                                                     float* __restrict__ scale_out = nullptr, int* range_flag = nullptr) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  __shared__ _Float16 th[64][66], tl[64][66];   // [r][c] halves of the tile (row stride 132 bytes: conflict-free column walks)
  __shared__ float red[4][64];
  const int r0 = bx * 64, c0 = by * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  if constexpr (SRC_PAIRS) {
    const _Float16* src = static_cast<const _Float16*>(src_);
    for (int i = ty; i < 64; i += 4) {
      const int r = r0 + i, c = c0 + tx;
      _Float16 hi = (_Float16)0.f, lo = (_Float16)0.f;
      if (r < R && c < C) {
        const _Float16* p = src + (size_t)r * 2 * C + pair_index(c);
        hi = p[0];
        lo = p[32];
      }
      th[i][tx] = hi;
      tl[i][tx] = lo;
    }
  } else {
    const float* src = static_cast<const float*>(src_);
    float S = 1.0f;
    if (amax_part) {   // (uniform) fold of the <= 256 partial maxima: one per thread, then across the four waves
      __shared__ float sred[4];
      // (n_part < 0: the -n_part ways of a producer's amax slot, kAmaxStride floats apart - common.hpp amax_publish)
      float m = n_part < 0 ? ((int)threadIdx.x < -n_part ? amax_part[threadIdx.x * kAmaxStride] : 0.f)
                           : ((int)threadIdx.x < n_part ? amax_part[threadIdx.x] : 0.f);
      m = wave_max(m);
      if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = m;
      __syncthreads();
      S = pair_scale_of(fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3])));
      if (scale_out && bx == 0 && by == 0 && threadIdx.x == 0) *scale_out = S;
    }
    float csum = 0.f;
    bool bad = false;
    for (int i = ty; i < 64; i += 4) {
      const int r = r0 + i, c = c0 + tx;
      const float v = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
      _Float16 hi, lo;
      split_pair(v * S, hi, lo);
      bad |= pair_hi_bad(hi);
      th[i][tx] = hi;
      tl[i][tx] = lo;
      csum += v;
      if constexpr (ROW) {
        if (r < R && c < C) {
          _Float16* p = dst_row + (size_t)r * 2 * C + pair_index(c);
          p[0] = hi;
          p[32] = lo;
        }
      }
    }
    if constexpr (SUM) red[ty][tx] = csum;
    range_flag_raise(range_flag, bad);
  }
  __syncthreads();
  if constexpr (SUM) {
    if (ty == 0 && c0 + tx < C) partial[(size_t)bx * C + c0 + tx] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
  }
  // out row c: the tile's 64 r = two pair groups; 16 lanes x 4 r per row (8-byte stores), 16 rows per pass
  if (!dst_t) return;   // (uniform) row pairs / column sums only
  const int q = threadIdx.x & 15, cr = threadIdx.x >> 4;
  for (int i = cr; i < 64; i += 16) {
    const int c = c0 + i, r = r0 + 4 * q;
    if (c < C && r < Rpad) {   // (Rpad % 32 == 0: the four r of a quad are inside or outside together)
      f16x4 vh = {th[4 * q][i], th[4 * q + 1][i], th[4 * q + 2][i], th[4 * q + 3][i]};
      f16x4 vl = {tl[4 * q][i], tl[4 * q + 1][i], tl[4 * q + 2][i], tl[4 * q + 3][i]};
      _Float16* p = dst_t + (size_t)c * 2 * Rpad + pair_index(r);
      *reinterpret_cast<f16x4*>(p) = vh;
      *reinterpret_cast<f16x4*>(p + 32) = vl;
    }
  }
}
template <bool SRC_PAIRS, bool ROW, bool SUM>
__global__ __launch_bounds__(256) void transpose_pairs_kernel(const void* __restrict__ src_, _Float16* __restrict__ dst_t, _Float16* __restrict__ dst_row,
                                                              int R, int C, int Rpad, float* __restrict__ partial,
                                                              const float* __restrict__ amax_part, int n_part, float* __restrict__ scale_out,
                                                              int* range_flag) {
  transpose_pairs_tile<SRC_PAIRS, ROW, SUM>(src_, dst_t, dst_row, R, C, Rpad, partial, blockIdx.x, blockIdx.y, amax_part, n_part, scale_out, range_flag);
}

// Row pairs (+ column partial sums) of an fp32 matrix WITHOUT a transposed output - what the transpose-free weight gradient leaves of a
// dy's split (tt_split_pairs_dual_parts with dst_t == NULL: twelve launches of a C2 step, on the data-gradient chain).  The tile kernel
// above moves such a matrix through LDS 4 bytes in and 2 + 2 bytes out per thread; this one streams it: a workgroup takes 64 rows x 128
// columns, a lane 8 consecutive columns (two 16-byte loads, one 16-byte store of his and one of los: a pair group is 32 columns) of four
// rows.  Same scale (the producer's amax slot or the max pass's partials), same partial layout [64-row block][C]; the column sums add
// a thread's four rows, then the four row quarters of a wave (shuffles), then the four waves, in that fixed order.
template <bool SUM>
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ src, _Float16* __restrict__ dst_row, int R, int C,
                                                         float* __restrict__ partial, const float* __restrict__ amax_part, int n_part,
                                                         float* __restrict__ scale_out, int* range_flag) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  __shared__ float sred[4];
  __shared__ float red[4][128];
  float S = 1.0f;
  if (amax_part) {   // (uniform) as transpose_pairs_tile
    float m = n_part < 0 ? ((int)threadIdx.x < -n_part ? amax_part[threadIdx.x * kAmaxStride] : 0.f)
                         : ((int)threadIdx.x < n_part ? amax_part[threadIdx.x] : 0.f);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = m;
    __syncthreads();
    S = pair_scale_of(fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3])));
    if (scale_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *scale_out = S;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.y * 128 + 8 * (lane & 15);
  const int r0 = blockIdx.x * 64 + 16 * wave + (lane >> 4);
  float cs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = 0.f;
  bool bad = false;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + 4 * i;
    if (r < R) {
      float v[8];
      *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(src + (size_t)r * C + c);
      *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(src + (size_t)r * C + c + 4);
      f16x8 qh, ql;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        _Float16 hi, lo;
        split_pair(v[j] * S, hi, lo);
        bad |= pair_hi_bad(hi);
        qh[j] = hi;
        ql[j] = lo;
        cs[j] += v[j];
      }
      _Float16* p = dst_row + (size_t)r * 2 * C + pair_index(c);
      *reinterpret_cast<f16x8*>(p) = qh;
      *reinterpret_cast<f16x8*>(p + 32) = ql;
    }
  }
  range_flag_raise(range_flag, bad);
  if constexpr (SUM) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      cs[j] += __shfl_xor(cs[j], 16, 64);
      cs[j] += __shfl_xor(cs[j], 32, 64);
    }
    if (lane < 16) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[wave][8 * lane + j] = cs[j];
    }
    __syncthreads();
    if (threadIdx.x < 128) partial[(size_t)blockIdx.x * C + blockIdx.y * 128 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}

// The same for a TABLE of fp32 matrices in one launch: the pair operands (row pairs for the forward / weight-gradient products, transposed
// pairs for the data-gradient product) of every weight the optimizer just rewrote - two dozen small launches per step otherwise.
struct PairTable {
  enum { MAXN = 32 };
  const float* src[MAXN];
  _Float16* row[MAXN];   // may be null
  _Float16* t[MAXN];     // may be null
  int R[MAXN], C[MAXN], Rpad[MAXN];
  int tile0[MAXN + 1];   // first workgroup of each entry
  int n;
  int* range_flag;
};
__global__ __launch_bounds__(256) void split_pairs_dual_multi_kernel(PairTable tb) {
  int e = 0;
  while (e + 1 < tb.n && (int)blockIdx.x >= tb.tile0[e + 1]) ++e;
  const int tile = blockIdx.x - tb.tile0[e], tr = (tb.Rpad[e] + 63) / 64;
  if (tb.row[e]) transpose_pairs_tile<false, true, false>(tb.src[e], tb.t[e], tb.row[e], tb.R[e], tb.C[e], tb.Rpad[e], nullptr, tile % tr, tile / tr, nullptr, 0, nullptr, tb.range_flag);
  else transpose_pairs_tile<false, false, false>(tb.src[e], tb.t[e], nullptr, tb.R[e], tb.C[e], tb.Rpad[e], nullptr, tile % tr, tile / tr, nullptr, 0, nullptr, tb.range_flag);
}

int launch_splitk_reduce(const float* partial, float* out, long long n, int splits, long long stride, hipStream_t s);  // gemm_f32.hip
int planes8_try(const void* x_planes, long long x_plane_stride, const void* w_planes, long long w_plane_stride, int planes, const float* bias,
                const float* residual, float* y, void* y_planes, long long y_plane_stride, int y_nplanes, int M, int N, int K, int act,
                void* ksplit_ws, size_t ksplit_ws_bytes_, hipStream_t s);                                             // gemm_planes8.hip
int planes8_would_run(int planes, int M, int N, int K, int act, int has_bias, int has_residual, int has_y, int y_nplanes);


// PatchEmbed rows for the bf16-plane path (dino_vision_transformer.py:166-171,236-247): one workgroup per token row m = (frame f, token t).
//   a [F (n + 1)][C P P] bf16: the im2col row of patch t - 1 (k = (c P + y) P + x, the conv weight's own order), ZERO for the class token
//   tokens [F][n + 1][D] fp32: what the GEMM's residual epilogue adds to - pos[t], and cls + pos[0] - bias for the class token (its zero
//   row meets the bias in the epilogue)
// so that ONE plane GEMM over all F (n + 1) rows (tt_linear_fwd_planes, residual = y = tokens) leaves prepare_tokens' result: no row
// gather, no separate class-token pass.
__global__ __launch_bounds__(256) void patch_rows_planes_kernel(const float* __restrict__ img, const int* __restrict__ frame_map,
                                                                  const float* __restrict__ bias, const float* __restrict__ cls,
                                                                  const float* __restrict__ pos, __bf16* __restrict__ a,
                                                                  float* __restrict__ tokens, int C, int H, int W, int P, int D, int gw, int n) {
  const int m = blockIdx.x, f = m / (n + 1), t = m - f * (n + 1);
  const int K = C * P * P;
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  bf16x4* arow = reinterpret_cast<bf16x4*>(a + (long long)m * K);
  float4* trow = reinterpret_cast<float4*>(tokens + (long long)m * D);
  const float4* prow = reinterpret_cast<const float4*>(pos + (long long)t * D);
  if (t == 0) {
    for (int i = threadIdx.x; i < K / 4; i += 256) arow[i] = (bf16x4){(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    for (int i = threadIdx.x; i < D / 4; i += 256) {
      const float4 c = reinterpret_cast<const float4*>(cls)[i], b = reinterpret_cast<const float4*>(bias)[i], q = prow[i];
      trow[i] = make_float4(c.x + q.x - b.x, c.y + q.y - b.y, c.z + q.z - b.z, c.w + q.w - b.w);
    }
    return;
  }
  const int src = frame_map ? frame_map[f] : f;
  const int gy = (t - 1) / gw, gx = (t - 1) - gy * gw;
  const float* base = img + ((long long)src * C * H + gy * P) * W + gx * P;
  const int P4 = P / 4;
  for (int i = threadIdx.x; i < K / 4; i += 256) {
    const int x4 = i % P4, cy = i / P4, y = cy % P, c = cy / P;
    const float4 v = *reinterpret_cast<const float4*>(base + ((long long)c * H + y) * W + 4 * x4);
    arow[i] = (bf16x4){(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  }
  for (int i = threadIdx.x; i < D / 4; i += 256) trow[i] = prow[i];
}

// The same rows for the fp16-PAIR path ("f16x3"): a [F (n + 1)][2 C P P] fp16 in pairs (groups of 32 k as [hi x 32][lo x 32]), the class
// token's row zero; tokens prefilled as above.  ONE pair GEMM over all rows (residual = y = tokens) then leaves prepare_tokens' result.
__global__ __launch_bounds__(256) void patch_rows_pairs_kernel(const float* __restrict__ img, const int* __restrict__ frame_map,
                                                                 const float* __restrict__ bias, const float* __restrict__ cls,
                                                                 const float* __restrict__ pos, _Float16* __restrict__ a,
                                                                 float* __restrict__ tokens, int C, int H, int W, int P, int D, int gw, int n,
                                                                 int* range_flag) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  const int m = blockIdx.x, f = m / (n + 1), t = m - f * (n + 1);
  const int K = C * P * P;
  _Float16* arow = a + (long long)m * 2 * K;
  float4* trow = reinterpret_cast<float4*>(tokens + (long long)m * D);
  const float4* prow = reinterpret_cast<const float4*>(pos + (long long)t * D);
  if (t == 0) {
    for (int i = threadIdx.x; i < K / 2; i += 256) reinterpret_cast<f16x4*>(arow)[i] = (f16x4){(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    for (int i = threadIdx.x; i < D / 4; i += 256) {
      const float4 c = reinterpret_cast<const float4*>(cls)[i], b = reinterpret_cast<const float4*>(bias)[i], q = prow[i];
      trow[i] = make_float4(c.x + q.x - b.x, c.y + q.y - b.y, c.z + q.z - b.z, c.w + q.w - b.w);
    }
    return;
  }
  const int src = frame_map ? frame_map[f] : f;
  const int gy = (t - 1) / gw, gx = (t - 1) - gy * gw;
  const float* base = img + ((long long)src * C * H + gy * P) * W + gx * P;
  const int P4 = P / 4;
  bool bad = false;
  for (int i = threadIdx.x; i < K / 4; i += 256) {
    const int x4 = i % P4, cy = i / P4, y = cy % P, c = cy / P;
    const float4 v = *reinterpret_cast<const float4*>(base + ((long long)c * H + y) * W + 4 * x4);
    _Float16 h0, l0, h1, l1, h2, l2, h3, l3;
    split_pair(v.x, h0, l0); split_pair(v.y, h1, l1); split_pair(v.z, h2, l2); split_pair(v.w, h3, l3);
    bad = bad || pair_hi_bad(h0) || pair_hi_bad(h1) || pair_hi_bad(h2) || pair_hi_bad(h3);
    _Float16* p = arow + pair_index(4 * i);   // k = 4 i .. 4 i + 3 lie in one group of 32
    *reinterpret_cast<f16x4*>(p) = (f16x4){h0, h1, h2, h3};
    *reinterpret_cast<f16x4*>(p + 32) = (f16x4){l0, l1, l2, l3};
  }
  range_flag_raise(range_flag, bad);
  for (int i = threadIdx.x; i < D / 4; i += 256) trow[i] = prow[i];
}

}  // namespace tt

using namespace tt;

extern "C" int tt_transpose_planes(const float* src, void* dst, int R, int C, int Rpad, tt_stream_t stream) {
  TT_REQUIRE(src && dst && R > 0 && C > 0 && Rpad >= R, "transpose_planes: bad arguments");
  hipLaunchKernelGGL(transpose_planes_kernel<false>, dim3((Rpad + 63) / 64, (C + 63) / 64), dim3(256), 0, as_stream(stream), src,
                     static_cast<__bf16*>(dst), R, C, Rpad, nullptr);
  TT_CHECK_LAUNCH("transpose_planes");
  return TT_OK;
}

namespace tt {
int launch_colsum_fold(const float* partial, float* out, int chunks, int N, hipStream_t s);   // rowops.hip
}

extern "C" size_t tt_transpose_planes_colsum_workspace_bytes(int R, int C, int Rpad) {
  if (R <= 0 || C <= 0 || Rpad < R) return 0;
  return (size_t)((Rpad + 63) / 64) * C * sizeof(float);
}

extern "C" int tt_transpose_planes_colsum(const float* src, void* dst, int R, int C, int Rpad, float* colsum, void* workspace,
                                          size_t workspace_bytes, tt_stream_t stream) {
  TT_REQUIRE(src && dst && colsum && workspace && R > 0 && C > 0 && Rpad >= R, "transpose_planes_colsum: bad arguments");
  TT_REQUIRE(workspace_bytes >= tt_transpose_planes_colsum_workspace_bytes(R, C, Rpad), "transpose_planes_colsum: workspace too small");
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(transpose_planes_kernel<true>, dim3((Rpad + 63) / 64, (C + 63) / 64), dim3(256), 0, as_stream(stream), src,
                     static_cast<__bf16*>(dst), R, C, Rpad, partial);
  TT_CHECK_LAUNCH("transpose_planes_colsum");
  return launch_colsum_fold(partial, colsum, (Rpad + 63) / 64, C, as_stream(stream));
}

extern "C" int tt_split_planes(const float* src, void* dst_planes, long long plane_stride, int planes, long long n, tt_stream_t stream) {
  TT_REQUIRE(src && dst_planes && n > 0 && planes >= 1 && planes <= 3, "split_planes: bad arguments");
  TT_REQUIRE(n % 8 == 0 && plane_stride % 8 == 0 && aligned16(src) && aligned16(dst_planes), "split_planes: n and the plane stride must be "
             "multiples of 8 elements and the buffers 16-byte aligned");
  long long blocks = (n / 8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), src, static_cast<__bf16*>(dst_planes),
                     plane_stride, planes, n / 8);
  TT_CHECK_LAUNCH("split_planes");
  return TT_OK;
}

static int linear_planes_impl(const void* x_planes, long long x_plane_stride, const void* w_planes, long long w_plane_stride, int planes,
                              const float* bias, const float* residual, float* y, float* pre_out, void* y_planes,
                              long long y_plane_stride, int y_nplanes, int M, int N, int K, int act, const float* gelu_pre, int splits,
                              long long split_stride, tt_stream_t stream, void* ksplit_ws = nullptr, size_t ksplit_ws_bytes_ = 0);

extern "C" int tt_linear_fwd_planes_route(int planes, int M, int N, int K, int act, int has_bias, int has_residual, int has_y, int y_nplanes,
                                          int has_pre_out) {
  if (tuning_knob(KNOB_PLANES_VARIANT) != 0 || has_pre_out) return 0;
  return planes8_would_run(planes, M, N, K, act, has_bias, has_residual, has_y, y_nplanes) ? 8 : 0;
}

extern "C" int tt_linear_fwd_planes(const void* x_planes, long long x_plane_stride, const void* w_planes, long long w_plane_stride, int planes,
                                    const float* bias, const float* residual, float* y, float* pre_out, void* y_planes,
                                    long long y_plane_stride, int y_nplanes, int M, int N, int K, int act, void* workspace, size_t workspace_bytes,
                                    tt_stream_t stream) {
  return linear_planes_impl(x_planes, x_plane_stride, w_planes, w_plane_stride, planes, bias, residual, y, pre_out, y_planes, y_plane_stride,
                            y_nplanes, M, N, K, act, nullptr, 1, 0, stream, workspace, workspace_bytes);
}

// dx[M,K] = dy[M,N] @ w[N,K] (* gelu'(gelu_pre)): dy in planes [M][N], the weight TRANSPOSED in planes wT [K][N] (tt_transpose_planes)
extern "C" int tt_linear_bwd_data_planes(const void* dy_planes, long long dy_plane_stride, const void* wT_planes, long long wT_plane_stride,
                                         int planes, const float* gelu_pre, float* dx, int M, int N, int K, void* workspace, size_t workspace_bytes,
                                         tt_stream_t stream) {
  TT_REQUIRE(dx, "linear_bwd_data_planes: null output");
  return linear_planes_impl(dy_planes, dy_plane_stride, wT_planes, wT_plane_stride, planes, nullptr, nullptr, dx, nullptr, nullptr, 0, 0, M, K, N, 0,
                            gelu_pre, 1, 0, stream, workspace, workspace_bytes);
}

// dw[N,K] = dy[M,N]^T @ x[M,K]: both operands TRANSPOSED and zero-padded along the reduction, dyT [N][Mpad], xT [K][Mpad].
// Split-K over Mpad (few output tiles, long reduction): partials in the workspace, folded in fixed order.
static int wgrad_splits(int N, int K, int Mpad) {
  const long long tiles = (long long)((N + 63) / 64) * (K / 64);
  int s = (int)((768 + tiles - 1) / tiles);
  const int smax = Mpad / 256;   // >= 4 slabs of 64 per slice
  if (s > smax) s = smax;
  if (s > 16) s = 16;
  return s < 1 ? 1 : s;
}
extern "C" size_t tt_linear_bwd_weight_planes_workspace_bytes(int N, int K, int Mpad) {
  const int s = wgrad_splits(N, K, Mpad);
  return s > 1 ? (size_t)s * N * K * sizeof(float) : 16;
}
extern "C" int tt_linear_bwd_weight_planes(const void* dyT_planes, long long dyT_plane_stride, const void* xT_planes, long long xT_plane_stride,
                                           int planes, float* dw, int N, int K, int Mpad, void* workspace, size_t workspace_bytes,
                                           tt_stream_t stream) {
  TT_REQUIRE(dw && workspace, "linear_bwd_weight_planes: null pointer");
  TT_REQUIRE(workspace_bytes >= tt_linear_bwd_weight_planes_workspace_bytes(N, K, Mpad), "linear_bwd_weight_planes: workspace too small");
  const int s = wgrad_splits(N, K, Mpad);
  if (s == 1)
    return linear_planes_impl(dyT_planes, dyT_plane_stride, xT_planes, xT_plane_stride, planes, nullptr, nullptr, dw, nullptr, nullptr, 0, 0, N, K,
                              Mpad, 0, nullptr, 1, 0, stream);
  float* part = static_cast<float*>(workspace);
  const int rc = linear_planes_impl(dyT_planes, dyT_plane_stride, xT_planes, xT_plane_stride, planes, nullptr, nullptr, part, nullptr, nullptr, 0, 0,
                                    N, K, Mpad, 0, nullptr, s, (long long)N * K, stream);
  if (rc != TT_OK) return rc;
  return launch_splitk_reduce(part, dw, (long long)N * K, s, (long long)N * K, as_stream(stream));
}

static int linear_planes_impl(const void* x_planes, long long x_plane_stride, const void* w_planes, long long w_plane_stride, int planes,
                              const float* bias, const float* residual, float* y, float* pre_out, void* y_planes,
                              long long y_plane_stride, int y_nplanes, int M, int N, int K, int act, const float* gelu_pre, int splits,
                              long long split_stride, tt_stream_t stream, void* ksplit_ws, size_t ksplit_ws_bytes_) {
  TT_REQUIRE(x_planes && w_planes && (y || y_planes), "linear_fwd_planes: null operand / no output");
  TT_REQUIRE(planes >= 1 && planes <= 3 && y_nplanes >= 0 && y_nplanes <= 3 && (y_nplanes == 0) == (y_planes == nullptr),
             "linear_fwd_planes: planes must be 1..3 and y_nplanes 0..3 (0 iff y_planes is null)");
  TT_REQUIRE(M > 0 && N > 0 && K > 0 && N % 64 == 0 && K % 64 == 0, "linear_fwd_planes: need N %% 64 == 0 and K %% 64 == 0 (got N=%d K=%d)", N, K);
  auto ok16 = [](const void* p) { return p == nullptr || aligned16(p); };
  TT_REQUIRE(aligned16(x_planes) && aligned16(w_planes) && ok16(y) && ok16(pre_out) && ok16(y_planes) && ok16(residual) && ok16(bias),
             "linear_fwd_planes: buffers must be 16-byte aligned");
  TT_REQUIRE(x_plane_stride % 8 == 0 && w_plane_stride % 8 == 0 && y_plane_stride % 8 == 0, "linear_fwd_planes: plane strides must be multiples of 8");
  TT_REQUIRE(splits >= 1 && (splits == 1 || (y && !bias && !residual && !pre_out && !y_planes && !act && !gelu_pre)),
             "linear_planes: a split-K launch writes plain fp32 partials only");
  TT_REQUIRE(gelu_pre == nullptr || aligned16(gelu_pre), "linear_planes: gelu_pre must be 16-byte aligned");
  PlaneArgs g{static_cast<const __bf16*>(x_planes), static_cast<const __bf16*>(w_planes), x_plane_stride, w_plane_stride, M, N, K, bias, residual, y,
              pre_out, static_cast<__bf16*>(y_planes), y_plane_stride, y_nplanes, act, gelu_pre, splits, split_stride, nullptr};
  hipStream_t s = as_stream(stream);
  const int variant = tuning_knob(KNOB_PLANES_VARIANT);   // tuning aid (tt_set_tuning_knob: A/B in one process)
  // whole-tile forward products on a grid that fills the chip: the persistent 8-phase kernel (gemm_planes8.hip)
  if (variant == 0 && !pre_out && !gelu_pre && splits == 1) {
    const int rc = planes8_try(x_planes, x_plane_stride, w_planes, w_plane_stride, planes, bias, residual, y, y_planes, y_plane_stride, y_nplanes, M, N,
                               K, act, ksplit_ws, ksplit_ws_bytes_, s);
    if (rc <= 0) return rc;
  }
  // tile: 128 x 128 when the grid still fills the chip more than twice over, else 64-row / 64-column tiles (ViT-S/16's N = 384
  // products: 1182 instead of 591 workgroups)
  const long long t128 = (long long)((M + 127) / 128) * (N / 128);
  const bool big = (N % 128 == 0) && t128 * splits >= 3 * 256;
  const bool wide = (N % 128 == 0) && (long long)((M + 63) / 64) * (N / 128) * splits >= 2 * 256;
  switch (planes) {
    case 1:
      if (big) {
        if (variant == 1) return launch_planes<1, 64, 2, 2, 2>(g, s);
        if (variant == 2) return launch_planes<1, 32, 2, 2, 3>(g, s);
        if (variant == 3) return launch_planes<1, 64, 2, 2, 3>(g, s);
        if (variant == 4 && N % 256 == 0) return launch_planes<1, 32, 4, 2, 3, 2, 4>(g, s);   // 256 x 256, 8 waves of 128 x 64
        if (variant == 5 && N % 256 == 0) return launch_planes<1, 64, 4, 2, 2, 2, 4>(g, s);   // same, 128-byte rows, 2 slabs
        if (variant == 6) return launch_planes<1, 32, 2, 2, 4, 4, 2>(g, s);                    // 256 x 128, 8 waves of 64 x 64
        if (variant == 7) return launch_planes<1, 64, 2, 2, 3, 4, 2>(g, s);
        if (variant == 8) return launch_planes<1, 32, 2, 2, 4, 2, 2, 1>(g, s);                 // 128 x 128, 4 compute + 4 loader waves
        if (variant == 9) return launch_planes<1, 64, 2, 2, 2, 2, 2, 1>(g, s);
        // default: 256 x 256 (8 waves) when that still makes >= 3 tiles per CU (ViT-B/16 qkv / fc1: 131 / 208 us against 150 / 213
        // for 128 x 128, profiles/r02_gemm_planes_variants.txt), else 128 x 128 with a 4-slab ring
        if (variant == 0 && N % 256 == 0 && (long long)((M + 255) / 256) * (N / 256) >= 3 * 256) return launch_planes<1, 64, 4, 2, 2, 2, 4>(g, s);
        return launch_planes<1, 32, 2, 2, 4>(g, s);
      }
      if (wide) return launch_planes<1, 64, 1, 2>(g, s);
      return launch_planes<1, 64, 1, 1>(g, s);
    case 2:
      if (big) return launch_planes<2, 32, 2, 2>(g, s);
      if (wide) return launch_planes<2, 32, 1, 2>(g, s);
      return launch_planes<2, 32, 1, 1>(g, s);
    default:
      if (big) {
        if (variant == 1) return launch_planes<3, 16, 2, 2, 2>(g, s);
        if (variant == 2) return launch_planes<3, 16, 2, 2, 4>(g, s);
        if (variant == 3) return launch_planes<3, 32, 2, 2, 2>(g, s);
        if (variant == 6) return launch_planes<3, 32, 2, 2, 2, 4, 2>(g, s);                    // 256 x 128, 8 waves, 64-byte rows
        if (variant == 7) return launch_planes<3, 16, 2, 2, 2, 4, 2>(g, s);
        if (variant == 8) return launch_planes<3, 16, 2, 2, 3, 2, 2, 1>(g, s);
        if (variant == 9) return launch_planes<3, 32, 2, 2, 2, 2, 2, 1>(g, s);
        // default: 256 x 128 (8 waves, 64-byte rows) when that still makes >= 3 tiles per CU (ViT-S/16 qkv / fc1: 160 / 230 us against
        // 185 / 247), else 128 x 128
        if (variant == 0 && (long long)((M + 255) / 256) * (N / 128) >= 3 * 256) return launch_planes<3, 32, 2, 2, 2, 4, 2>(g, s);
        return launch_planes<3, 16, 2, 2, 3>(g, s);
      }
      if (wide) {
        if (variant == 1) return launch_planes<3, 16, 1, 2, 2>(g, s);
        return launch_planes<3, 32, 1, 2, 2>(g, s);
      }
      return launch_planes<3, 16, 1, 1>(g, s);
  }
}


// ---- fp16-pair operands (the "f16x3" mode) ------------------------------------------------------------------------------------
extern "C" int tt_split_pairs(const float* src, void* dst_pairs, long long n, int* range_flag, tt_stream_t stream) {
  TT_REQUIRE(src && dst_pairs && n > 0, "split_pairs: bad arguments");
  TT_REQUIRE(n % 32 == 0 && aligned16(src) && aligned16(dst_pairs), "split_pairs: n must be a multiple of 32 and the buffers 16-byte aligned");
  long long blocks = (n / 8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), src, static_cast<_Float16*>(dst_pairs), n / 8, range_flag);
  TT_CHECK_LAUNCH("split_pairs");
  return TT_OK;
}

extern "C" int tt_join_pairs(const void* src_pairs, float* dst, long long n, tt_stream_t stream) {
  TT_REQUIRE(src_pairs && dst && n > 0, "join_pairs: bad arguments");
  TT_REQUIRE(n % 32 == 0 && aligned16(src_pairs) && aligned16(dst), "join_pairs: n must be a multiple of 32 and the buffers 16-byte aligned");
  long long blocks = (n / 8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(join_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), static_cast<const _Float16*>(src_pairs), dst, n / 8);
  TT_CHECK_LAUNCH("join_pairs");
  return TT_OK;
}

// y[M,N] = act(x[M,K] @ w[N,K]^T + bias) (+ residual) (* gelu'(gelu_pre)) on pair operands; outputs any of y fp32, pre_out fp32,
// y_pairs.  splits > 1: plain fp32 partials of a K split (weight gradients).
static int linear_pairs_impl(const void* x_pairs, const void* w_pairs, const float* bias, const float* residual, float* y, float* pre_out,
                             void* y_pairs, int M, int N, int K, int act, const float* gelu_pre, int splits, long long split_stride,
                             tt_stream_t stream, const float* out_scale = nullptr, void* ksplit_ws = nullptr, size_t ksplit_ws_bytes_ = 0,
                             int* range_flag = nullptr, float* amax_out = nullptr) {
  TT_REQUIRE(x_pairs && w_pairs && (y || y_pairs), "linear_fwd_pairs: null operand / no output");
  TT_REQUIRE(M > 0 && N > 0 && K > 0 && N % 64 == 0 && K % 32 == 0, "linear_fwd_pairs: need N %% 64 == 0 and K %% 32 == 0 (got N=%d K=%d)", N, K);
  auto ok16 = [](const void* p) { return p == nullptr || aligned16(p); };
  TT_REQUIRE(aligned16(x_pairs) && aligned16(w_pairs) && ok16(y) && ok16(pre_out) && ok16(y_pairs) && ok16(residual) && ok16(bias) && ok16(gelu_pre),
             "linear_fwd_pairs: buffers must be 16-byte aligned");
  TT_REQUIRE(splits >= 1 && (splits == 1 || (y && !bias && !residual && !pre_out && !y_pairs && !act && !gelu_pre)),
             "linear_pairs: a split-K launch writes plain fp32 partials only");
  hipStream_t s = as_stream(stream);
  const bool no8 = tuning_knob(KNOB_PAIRS_NO8) != 0;   // tuning aid: the general kernel everywhere
  if (!no8 && splits == 1) {
    const int rc = pairs8_try(x_pairs, w_pairs, bias, residual, y, pre_out, y_pairs, gelu_pre, out_scale, M, N, K, act, ksplit_ws, ksplit_ws_bytes_,
                              range_flag, amax_out, s);
    if (rc <= 0) return rc;
  }
  // the general kernel sees rows of 2 K 16-bit elements in K-tiles of 64 (= one pair group)
  PlaneArgs g{static_cast<const __bf16*>(x_pairs), static_cast<const __bf16*>(w_pairs), 0, 0, M, N, 2 * K, bias, residual, y,
              pre_out, static_cast<__bf16*>(y_pairs), 0, y_pairs ? 1 : 0, act, gelu_pre, splits, split_stride, out_scale, range_flag, amax_out};
  const long long t128 = (long long)((M + 127) / 128) * (N / 128);
  const bool big = (N % 128 == 0) && t128 * splits >= 3 * 256;
  const bool wide = (N % 128 == 0) && (long long)((M + 63) / 64) * (N / 128) * splits >= 2 * 256;
  // Ring depth (round 5; knob TT_PAIRS_NBUF: 0 = this rule, 2 .. 6 = forced).  These are the launches that do not fill the chip, and in a
  // step their operands are cold (HBM / MALL, not L2): a workgroup's K loop is a chain of LDS-DMA latencies - one per 32-deep K-tile with a
  // double buffer, a half / a third of one with two / three K-tiles in flight.  Three slabs everywhere (C2 -0.8 %); four on the 64 x 64
  // tile while the grid is under two workgroups per CU, where the 64 KB cost no residency (C1 2.39 -> 2.09 ms; at 594 tiles - the
  // 6304-row launches of C2 - four measured slower than two).  With hot operands (tools/ab_pairs.py) the depth changes nothing.
  const int knob = tuning_knob(KNOB_PAIRS_NBUF);
  const long long t64 = (long long)((M + 63) / 64) * (N / 64) * splits;
  const int nbuf = knob ? knob : (!big && !wide && t64 <= 2LL * device_cu_count() ? 4 : 3);
  if (big) return nbuf >= 3 ? launch_planes<1, 64, 2, 2, 3, 2, 2, 0, true>(g, s) : launch_planes<1, 64, 2, 2, 2, 2, 2, 0, true>(g, s);
  if (wide) return nbuf >= 3 ? launch_planes<1, 64, 1, 2, 3, 2, 2, 0, true>(g, s) : launch_planes<1, 64, 1, 2, 2, 2, 2, 0, true>(g, s);
  if (nbuf >= 6) return launch_planes<1, 64, 1, 1, 6, 2, 2, 0, true>(g, s);
  if (nbuf == 5) return launch_planes<1, 64, 1, 1, 5, 2, 2, 0, true>(g, s);
  if (nbuf == 4) return launch_planes<1, 64, 1, 1, 4, 2, 2, 0, true>(g, s);
  if (nbuf == 3) return launch_planes<1, 64, 1, 1, 3, 2, 2, 0, true>(g, s);
  return launch_planes<1, 64, 1, 1, 2, 2, 2, 0, true>(g, s);
}

extern "C" int tt_linear_fwd_pairs_route(int M, int N, int K, int act, int has_bias, int has_residual, int has_y, int has_y_pairs,
                                         int has_pre_out) {
  (void)has_bias;
  if (tuning_knob(KNOB_PAIRS_NO8) != 0) return 0;
  return pairs8_would_run(M, N, K, act, has_residual, has_y, has_y_pairs, has_pre_out, 0) ? 8 : 0;
}

extern "C" int tt_linear_fwd_pairs(const void* x_pairs, const void* w_pairs, const float* bias, const float* residual, float* y, float* pre_out,
                                   void* y_pairs, int M, int N, int K, int act, void* workspace, size_t workspace_bytes, int* range_flag,
                                   tt_stream_t stream) {
  return linear_pairs_impl(x_pairs, w_pairs, bias, residual, y, pre_out, y_pairs, M, N, K, act, nullptr, 1, 0, stream, nullptr, workspace, workspace_bytes,
                           range_flag);
}

// fp32 [R][C] -> transposed pairs [C][2 Rpad] (+ row-major pairs [R][2 C], + fp32 column sums); see transpose_pairs_kernel.
// scale_out (a gradient: see transpose_pairs_tile): the pairs hold src * S, S = the power of two written to *scale_out.
constexpr int kAmaxParts = 256;
extern "C" size_t tt_split_pairs_dual_workspace_bytes(int R, int C, int Rpad) {
  if (R <= 0 || C <= 0 || Rpad < R) return 0;
  return ((size_t)((Rpad + 63) / 64) * C + kAmaxParts) * sizeof(float);
}

// colsum_parts (tt_split_pairs_dual_parts): the column partial sums [ceil(Rpad / 64)][C] are LEFT there, unfolded, for the launch that
// folds the weight gradient's split partials to fold them too (tt_linear_bwd_weight_pairs_tn_bias: one launch less per dy)
static int split_pairs_dual_impl(const float* src, void* dst_t_pairs, void* dst_row_pairs, float* colsum, float* colsum_parts, float* scale_out, int R,
                                 int C, int Rpad, void* workspace, size_t workspace_bytes, int* range_flag, tt_stream_t stream,
                                 const float* amax_in = nullptr) {
  TT_REQUIRE(src && (dst_t_pairs || dst_row_pairs) && R > 0 && C > 0 && Rpad >= R && Rpad % 32 == 0,
             "split_pairs_dual: bad arguments (an output, Rpad a multiple of 32)");
  TT_REQUIRE(!dst_row_pairs || C % 32 == 0, "split_pairs_dual: row-major pairs need C %% 32 == 0 (got %d)", C);
  TT_REQUIRE(!((colsum && !colsum_parts) || (scale_out && !amax_in)) || (workspace && workspace_bytes >= tt_split_pairs_dual_workspace_bytes(R, C, Rpad)),
             "split_pairs_dual: workspace too small");   // (colsum_parts + scale: the max pass's partials still live behind the colsum region)
  TT_REQUIRE((reinterpret_cast<uintptr_t>(dst_t_pairs) & 7u) == 0, "split_pairs_dual: the transposed output must be 8-byte aligned");
  TT_REQUIRE(!scale_out || aligned16(src), "split_pairs_dual: a scaled split needs a 16-byte aligned source");
  const dim3 grid((Rpad + 63) / 64, (C + 63) / 64), block(256);
  hipStream_t s = as_stream(stream);
  _Float16* dt = static_cast<_Float16*>(dst_t_pairs);
  _Float16* dr = static_cast<_Float16*>(dst_row_pairs);
  float* partial = colsum_parts ? colsum_parts : static_cast<float*>(workspace);
  if (colsum_parts) colsum = colsum_parts;   // (non-null: the SUM instantiations below; never folded here)
  float* amax_part = nullptr;
  int n_part = 0;
  if (scale_out && amax_in) {   // the producer of src left max |src| there (amax_publish): no max pass
    amax_part = const_cast<float*>(amax_in);
    n_part = -kAmaxWays;
  } else if (scale_out) {
    amax_part = static_cast<float*>(workspace) + (size_t)((Rpad + 63) / 64) * C;
    const long long n = (long long)R * C;
    n_part = (int)((n + 4095) / 4096 < kAmaxParts ? (n + 4095) / 4096 : kAmaxParts);
    hipLaunchKernelGGL(amax_partial_kernel, dim3(n_part), dim3(256), 0, s, src, n, amax_part);
  }
  if (dr && !dt && C % 128 == 0 && aligned16(src) && aligned16(dr) && tuning_knob(KNOB_SPLIT_ROWS) != 0) {   // row pairs only: the streaming kernel
    const dim3 g2((Rpad + 63) / 64, C / 128);
    if (colsum) hipLaunchKernelGGL((split_rows_kernel<true>), g2, block, 0, s, src, dr, R, C, partial, amax_part, n_part, scale_out, range_flag);
    else hipLaunchKernelGGL((split_rows_kernel<false>), g2, block, 0, s, src, dr, R, C, nullptr, amax_part, n_part, scale_out, range_flag);
    TT_CHECK_LAUNCH("split_rows");
    if (colsum && !colsum_parts) return launch_colsum_fold(partial, colsum, (Rpad + 63) / 64, C, s);
    return TT_OK;
  }
  if (dr && colsum) hipLaunchKernelGGL((transpose_pairs_kernel<false, true, true>), grid, block, 0, s, src, dt, dr, R, C, Rpad, partial, amax_part, n_part, scale_out, range_flag);
  else if (dr) hipLaunchKernelGGL((transpose_pairs_kernel<false, true, false>), grid, block, 0, s, src, dt, dr, R, C, Rpad, nullptr, amax_part, n_part, scale_out, range_flag);
  else if (colsum) hipLaunchKernelGGL((transpose_pairs_kernel<false, false, true>), grid, block, 0, s, src, dt, dr, R, C, Rpad, partial, amax_part, n_part, scale_out, range_flag);
  else hipLaunchKernelGGL((transpose_pairs_kernel<false, false, false>), grid, block, 0, s, src, dt, dr, R, C, Rpad, nullptr, amax_part, n_part, scale_out, range_flag);
  TT_CHECK_LAUNCH("split_pairs_dual");
  if (colsum && !colsum_parts) return launch_colsum_fold(partial, colsum, (Rpad + 63) / 64, C, s);
  return TT_OK;
}
extern "C" int tt_split_pairs_dual(const float* src, void* dst_t_pairs, void* dst_row_pairs, float* colsum, float* scale_out, int R, int C, int Rpad,
                                   void* workspace, size_t workspace_bytes, int* range_flag, tt_stream_t stream) {
  return split_pairs_dual_impl(src, dst_t_pairs, dst_row_pairs, colsum, nullptr, scale_out, R, C, Rpad, workspace, workspace_bytes, range_flag, stream);
}
extern "C" int tt_split_pairs_dual_parts(const float* src, void* dst_t_pairs, void* dst_row_pairs, float* colsum_parts, float* scale_out,
                                         const float* amax_in, int R, int C, int Rpad, void* workspace, size_t workspace_bytes, int* range_flag,
                                         tt_stream_t stream) {
  TT_REQUIRE(colsum_parts && aligned16(colsum_parts), "split_pairs_dual_parts: colsum_parts [ceil(Rpad / 64)][C] is required (16-byte aligned)");
  TT_REQUIRE(!amax_in || scale_out, "split_pairs_dual_parts: amax_in is the scaled split's maximum (scale_out required)");
  return split_pairs_dual_impl(src, dst_t_pairs, dst_row_pairs, nullptr, colsum_parts, scale_out, R, C, Rpad, workspace, workspace_bytes, range_flag, stream,
                               amax_in);
}

// n fp32 matrices [R_i][C_i] -> row pairs and / or transposed pairs [C_i][2 Rpad_i] each, ONE launch per 32 of them
extern "C" int tt_split_pairs_dual_multi(const float* const* src, void* const* dst_t_pairs, void* const* dst_row_pairs, const int* R, const int* C,
                                         const int* Rpad, int n, int* range_flag, tt_stream_t stream) {
  TT_REQUIRE(src && dst_t_pairs && dst_row_pairs && R && C && Rpad && n >= 0, "split_pairs_dual_multi: bad arguments");
  for (int i0 = 0; i0 < n; i0 += PairTable::MAXN) {
    PairTable tb;
    tb.n = n - i0 < PairTable::MAXN ? n - i0 : PairTable::MAXN;
    tb.range_flag = range_flag;
    int tiles = 0;
    for (int i = 0; i < tb.n; ++i) {
      const int j = i0 + i;
      TT_REQUIRE(src[j] && (dst_t_pairs[j] || dst_row_pairs[j]) && R[j] > 0 && C[j] > 0 && Rpad[j] >= R[j] && Rpad[j] % 32 == 0,
                 "split_pairs_dual_multi: entry %d: bad arguments (an output, Rpad a multiple of 32)", j);
      TT_REQUIRE(!dst_row_pairs[j] || C[j] % 32 == 0, "split_pairs_dual_multi: entry %d: row-major pairs need C %% 32 == 0 (got %d)", j, C[j]);
      TT_REQUIRE((reinterpret_cast<uintptr_t>(dst_t_pairs[j]) & 7u) == 0, "split_pairs_dual_multi: entry %d: the transposed output must be 8-byte aligned", j);
      tb.src[i] = src[j];
      tb.t[i] = static_cast<_Float16*>(dst_t_pairs[j]);
      tb.row[i] = static_cast<_Float16*>(dst_row_pairs[j]);
      tb.R[i] = R[j]; tb.C[i] = C[j]; tb.Rpad[i] = Rpad[j];
      tb.tile0[i] = tiles;
      tiles += ((Rpad[j] + 63) / 64) * ((C[j] + 63) / 64);
    }
    tb.tile0[tb.n] = tiles;
    if (tiles == 0) continue;
    hipLaunchKernelGGL(split_pairs_dual_multi_kernel, dim3((unsigned)tiles), dim3(256), 0, as_stream(stream), tb);
    TT_CHECK_LAUNCH("split_pairs_dual_multi");
  }
  return TT_OK;
}

// pairs [R][2 C] -> transposed pairs [C][2 Rpad]
extern "C" int tt_transpose_pairs(const void* src_pairs, void* dst_t_pairs, int R, int C, int Rpad, tt_stream_t stream) {
  TT_REQUIRE(src_pairs && dst_t_pairs && R > 0 && C > 0 && C % 32 == 0 && Rpad >= R && Rpad % 32 == 0,
             "transpose_pairs: bad arguments (C and Rpad must be multiples of 32)");
  TT_REQUIRE((reinterpret_cast<uintptr_t>(dst_t_pairs) & 7u) == 0, "transpose_pairs: the output must be 8-byte aligned");
  hipLaunchKernelGGL((transpose_pairs_kernel<true, false, false>), dim3((Rpad + 63) / 64, (C + 63) / 64), dim3(256), 0, as_stream(stream), src_pairs,
                     static_cast<_Float16*>(dst_t_pairs), static_cast<_Float16*>(nullptr), R, C, Rpad, static_cast<float*>(nullptr),
                     static_cast<const float*>(nullptr), 0, static_cast<float*>(nullptr), static_cast<int*>(nullptr));
  TT_CHECK_LAUNCH("transpose_pairs");
  return TT_OK;
}

// dx[M,K] = dy[M,N] @ w[N,K] (* gelu'(gelu_pre)): dy in pairs [M][2 N], the weight TRANSPOSED in pairs wT [K][2 N].
// dy_scale (device scalar or null): the pairs hold dy * S (tt_split_pairs_dual's scale_out) - the product is divided by S.
extern "C" int tt_linear_bwd_data_pairs(const void* dy_pairs, const void* wT_pairs, const float* gelu_pre, float* dx, const float* dy_scale, int M,
                                        int N, int K, void* workspace, size_t workspace_bytes, float* amax_out, tt_stream_t stream) {
  TT_REQUIRE(dx, "linear_bwd_data_pairs: null output");
  return linear_pairs_impl(dy_pairs, wT_pairs, nullptr, nullptr, dx, nullptr, nullptr, M, K, N, 0, gelu_pre, 1, 0, stream, dy_scale, workspace, workspace_bytes,
                           nullptr, amax_out);
}

// dw[N,K] = dy[M,N]^T @ x[M,K]: both operands transposed in pairs, dyT [N][2 Mpad], xT [K][2 Mpad] (zero beyond M).  Split-K over Mpad
// (few output tiles, long reduction): partials in the workspace, folded in fixed order.
static int wgrad_pairs_splits(int N, int K, int Mpad) {
  const long long tiles = (long long)((N + 127) / 128) * ((K + 127) / 128);
  int s = (int)((640 + tiles - 1) / tiles);
  const int smax = Mpad / 128;   // >= 4 K-tiles of 32 per slice
  if (s > smax) s = smax;
  if (s > 32) s = 32;
  return s < 1 ? 1 : s;
}
extern "C" size_t tt_linear_bwd_weight_pairs_workspace_bytes(int N, int K, int Mpad) {
  const int s = wgrad_pairs_splits(N, K, Mpad);
  return s > 1 ? (size_t)s * N * K * sizeof(float) : 16;
}
extern "C" int tt_linear_bwd_weight_pairs(const void* dyT_pairs, const void* xT_pairs, float* dw, const float* dy_scale, int N, int K, int Mpad,
                                          void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  TT_REQUIRE(dw && workspace, "linear_bwd_weight_pairs: null pointer");
  TT_REQUIRE(workspace_bytes >= tt_linear_bwd_weight_pairs_workspace_bytes(N, K, Mpad), "linear_bwd_weight_pairs: workspace too small");
  const int s = wgrad_pairs_splits(N, K, Mpad);
  if (s == 1) return linear_pairs_impl(dyT_pairs, xT_pairs, nullptr, nullptr, dw, nullptr, nullptr, N, K, Mpad, 0, nullptr, 1, 0, stream, dy_scale);
  float* part = static_cast<float*>(workspace);
  const int rc = linear_pairs_impl(dyT_pairs, xT_pairs, nullptr, nullptr, part, nullptr, nullptr, N, K, Mpad, 0, nullptr, s, (long long)N * K, stream,
                                   dy_scale);
  if (rc != TT_OK) return rc;
  return launch_splitk_reduce(part, dw, (long long)N * K, s, (long long)N * K, as_stream(stream));
}

// workspace: the im2col rows (256-byte rounded), then the GEMM's K-split block (common.hpp KsplitWs; its counters are zeroed by the call)
static size_t patch_rows_bytes(int F, int C, int H, int W, int P, int elem) {
  return ((size_t)F * (1 + (size_t)(H / P) * (W / P)) * C * P * P * elem + 255) / 256 * 256;
}
static int patch_ksplit_init(void* ws, tt_stream_t stream) {
  if (hipMemsetAsync(ws, 0, ksplit_ws_counter_bytes(), as_stream(stream)) != hipSuccess) {
    set_error("patch_embed: hipMemsetAsync of the K-split counters failed");
    return TT_ELAUNCH;
  }
  return TT_OK;
}
extern "C" size_t tt_patch_embed_planes_workspace_bytes(int F, int C, int H, int W, int P) {
  if (F <= 0 || C <= 0 || P <= 0 || H < P || W < P) return 0;
  return patch_rows_bytes(F, C, H, W, P, 2) + ksplit_ws_bytes();
}

extern "C" int tt_patch_embed_fwd_planes(const float* img, const int32_t* frame_map, const void* w_planes, const float* bias, const float* cls,
                                         const float* pos, float* tokens, int F, int C, int H, int W, int P, int D, void* workspace,
                                         size_t workspace_bytes, tt_stream_t stream) {
  TT_REQUIRE(img && w_planes && bias && cls && pos && tokens && workspace, "patch_embed_planes: null pointer");
  TT_REQUIRE(F > 0 && C > 0 && P > 0 && H % P == 0 && W % P == 0, "patch_embed_planes: H, W must be multiples of the patch size");
  const int n = (H / P) * (W / P), K = C * P * P;
  const long long M = (long long)F * (n + 1);
  TT_REQUIRE(P % 4 == 0 && W % 4 == 0 && K % 64 == 0 && D % 64 == 0, "patch_embed_planes: need P %% 4 == 0, W %% 4 == 0, C P P %% 64 == 0, D %% 64 == 0");
  TT_REQUIRE(aligned16(img) && aligned16(bias) && aligned16(cls) && aligned16(pos) && aligned16(tokens) && aligned16(workspace),
             "patch_embed_planes: buffers must be 16-byte aligned");
  TT_REQUIRE(M * (long long)(K > D ? K : D) < (1ll << 31), "patch_embed_planes: F (n + 1) max(C P P, D) exceeds the int range");
  TT_REQUIRE(workspace_bytes >= tt_patch_embed_planes_workspace_bytes(F, C, H, W, P), "patch_embed_planes: workspace too small");
  hipLaunchKernelGGL(patch_rows_planes_kernel, dim3((unsigned)M), dim3(256), 0, as_stream(stream), img, frame_map, bias, cls, pos,
                     static_cast<__bf16*>(workspace), tokens, C, H, W, P, D, W / P, n);
  TT_CHECK_LAUNCH("patch_embed_planes.rows");
  void* kws = static_cast<unsigned char*>(workspace) + patch_rows_bytes(F, C, H, W, P, 2);
  const int rc = patch_ksplit_init(kws, stream);
  if (rc != TT_OK) return rc;
  return tt_linear_fwd_planes(workspace, M * K, w_planes, (long long)D * K, 1, bias, tokens, tokens, nullptr, nullptr, 0, 0, (int)M, D, K, 0, kws,
                              ksplit_ws_bytes(), stream);
}

// prepare_tokens on fp16-pair operands (the "f16x3" mode): w_pairs [D][2 C P P]; workspace: the im2col rows in pairs, F (n + 1) x C P P x 4 bytes,
// then the GEMM's K-split block
extern "C" size_t tt_patch_embed_pairs_workspace_bytes(int F, int C, int H, int W, int P) {
  if (F <= 0 || C <= 0 || P <= 0 || H < P || W < P) return 0;
  return patch_rows_bytes(F, C, H, W, P, 4) + ksplit_ws_bytes();
}

extern "C" int tt_patch_embed_fwd_pairs(const float* img, const int32_t* frame_map, const void* w_pairs, const float* bias, const float* cls,
                                        const float* pos, float* tokens, int F, int C, int H, int W, int P, int D, void* workspace,
                                        size_t workspace_bytes, int* range_flag, tt_stream_t stream) {
  TT_REQUIRE(img && w_pairs && bias && cls && pos && tokens && workspace, "patch_embed_pairs: null pointer");
  TT_REQUIRE(F > 0 && C > 0 && P > 0 && H % P == 0 && W % P == 0, "patch_embed_pairs: H, W must be multiples of the patch size");
  const int n = (H / P) * (W / P), K = C * P * P;
  const long long M = (long long)F * (n + 1);
  TT_REQUIRE(P % 4 == 0 && W % 4 == 0 && K % 32 == 0 && D % 64 == 0, "patch_embed_pairs: need P %% 4 == 0, W %% 4 == 0, C P P %% 32 == 0, D %% 64 == 0");
  TT_REQUIRE(aligned16(img) && aligned16(bias) && aligned16(cls) && aligned16(pos) && aligned16(tokens) && aligned16(workspace) && aligned16(w_pairs),
             "patch_embed_pairs: buffers must be 16-byte aligned");
  TT_REQUIRE(M * (long long)(K > D ? K : D) < (1ll << 31), "patch_embed_pairs: F (n + 1) max(C P P, D) exceeds the int range");
  TT_REQUIRE(workspace_bytes >= tt_patch_embed_pairs_workspace_bytes(F, C, H, W, P), "patch_embed_pairs: workspace too small");
  hipLaunchKernelGGL(patch_rows_pairs_kernel, dim3((unsigned)M), dim3(256), 0, as_stream(stream), img, frame_map, bias, cls, pos,
                     static_cast<_Float16*>(workspace), tokens, C, H, W, P, D, W / P, n, range_flag);
  TT_CHECK_LAUNCH("patch_embed_pairs.rows");
  void* kws = static_cast<unsigned char*>(workspace) + patch_rows_bytes(F, C, H, W, P, 4);
  const int rc = patch_ksplit_init(kws, stream);
  if (rc != TT_OK) return rc;
  return linear_pairs_impl(workspace, w_pairs, bias, tokens, tokens, nullptr, nullptr, (int)M, D, K, 0, nullptr, 1, 0, stream, nullptr, kws, ksplit_ws_bytes());
}
