// Weight gradient of an nn.Linear on fp16-PAIR operands WITHOUT transposes (round 4):
//     dW [N][K] = dy^T x = sum_m dy[m][n] x[m][k]
// (the grad of `F.linear`, dino_vision_transformer.py:94-103,115-130 under autograd) with BOTH operands read as they exist - dy in ROW pairs
// [M][2 N] (what the data-gradient product reads as well) and the layer's input x in ROW pairs [M][2 K] (what the forward kept) - where
// tt_linear_bwd_weight_pairs wants both transposed ([N][2 M], [K][2 M]): three transpose launches per Linear and their HBM round trips
// go away.  The reduction index m is the ROW index of both operands: a 32-row chunk of each goes HBM -> LDS by LDS-DMA as it lies, and
// the MFMA fragments (8 consecutive reduction elements of one output row / column per lane) are gathered by TRANSPOSING LDS reads,
// ds_read_b64_tr_b16 - the recipe of the pair attention kernel's V^T operand (attention_pairs.hip), here for both operands.
//
//   * workgroup = 4 waves, output tile 128 (n) x 128 (k), wave tile 64 x 64 (two accumulator sets of 4 MFMA tiles: 128 registers);
//     two workgroups per CU (64 KB of LDS each: two stages of [32 m][128 n] + [32 m][128 k] pairs = 2 x 32 KB);
//   * a stage = 32 rows of m: row image 512 B = 8 quarters of 64 B ([hi x 32][lo x 32] of four 32-column groups), quarter q stored at
//     q ^ (m & 3) -> conflict-free transposed reads (the 4 rows a 16-lane group gathers differ in their quarter);
//   * per stage and wave 32 transposed reads (16 fragments: 2 column groups x 2 k-steps x (hi, lo) per operand) feed 24 MFMAs
//     (2 x 2 MFMA tiles x 2 k-steps x 3 products);
//   * rows >= M read as zeros (buffer range check of the LDS-DMA), so any M;
//   * the M range is split over workgroups (few output tiles: 9 .. 36 at ViT-S/16); partials [split][N][K] are folded in a fixed order by
//     splitk_reduce_kernel.
#include "common.hpp"

namespace tt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* tn_lds_ptr_t;

struct TnArgs {
  const _Float16* A;   // dy, row pairs [M][2 N]
  const _Float16* B;   // x, row pairs [M][2 K]
  float* C;            // [splits][N][K] (splits == 1: dW itself)
  const float* a_scale;   // device scalar S the dy pairs were scaled by (tt_split_pairs_dual) or null: the product is divided by it
  int M, N, K;
  int ntk, ntiles, splits, nchunks;   // tiles along K, tiles, splits of the m range, 32-row chunks
  int xcd_groups;                     // splits % 8 == 0: the workgroups of a split share an XCD
  // splits == 1 with a bias gradient (tt_linear_bwd_weight_pairs_tn_bias: small row counts, BASELINE C1): there is no fold launch for the
  // column sums to ride on - the workgroups behind the last tile fold them, 64 columns each (colsum_stage2's order: the same bits)
  const float* colpart;
  float* db;
  int colparts;
};

__global__ __launch_bounds__(256, 2) void gemm_pairs_tn_kernel(TnArgs g) {
  constexpr int CH_B = 32 * 512;        // one operand's chunk
  constexpr int STAGE_B = 2 * CH_B;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63;
  if ((int)blockIdx.x >= g.ntiles * g.splits) {   // (uniform) a column-sum workgroup
    colsum_fold_block(g.colpart, g.db, g.colparts, g.N, g.N, 0, (int)blockIdx.x - g.ntiles * g.splits, reinterpret_cast<float (*)[64]>(smem));
    return;
  }
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  // workgroup -> (tile, split).  The tiles of one split read the SAME 32-row chunks of dy and x: they are placed on ONE XCD (workgroup
  // ids go round-robin over the 8 XCDs; splits is a multiple of 8 then), so that a chunk is fetched into one L2 once instead of into all
  // eight (PMC before: L2 hit rate 0.23, 194 MB fetched per launch for 48 MB of operands).
  int tile, split;
  if (g.xcd_groups) {
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    split = xcd + 8 * (loc / g.ntiles);
    tile = loc % g.ntiles;
  } else {
    tile = blockIdx.x % g.ntiles;
    split = blockIdx.x / g.ntiles;
  }
  const int tn = tile / g.ntk, tk = tile - tn * g.ntk;
  const int n0 = tn * 128, k0 = tk * 128;
  const int c0 = (int)((long long)split * g.nchunks / g.splits), c1 = (int)((long long)(split + 1) * g.nchunks / g.splits);

  // ---- LDS-DMA: per stage and wave 4 + 4 instructions of 2 rows x 512 B; lane -> (row, 16-byte slot), source slot = slot ^ ((m & 3) << 2)
  const unsigned a_row_b = (unsigned)g.N * 4u, b_row_b = (unsigned)g.K * 4u;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(g.A), 0, (unsigned)g.M * a_row_b, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(g.B), 0, (unsigned)g.M * b_row_b, 0x00020000);
  const int l_row = lane >> 5, l_slot = lane & 31;
  unsigned a_voff[4], b_voff[4];   // of chunk 0
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 8 + 2 * i + l_row;
    const unsigned src_slot = (unsigned)(l_slot ^ ((row & 3) << 2));
    a_voff[i] = (unsigned)row * a_row_b + (unsigned)n0 * 4u + src_slot * 16u;
    b_voff[i] = (unsigned)row * b_row_b + (unsigned)k0 * 4u + src_slot * 16u;
  }
  auto issue = [&](int chunk, int buf) {
    unsigned char* dst = smem + buf * STAGE_B + wave * 8 * 512;
    const unsigned ao = (unsigned)chunk * 32u * a_row_b, bo = (unsigned)chunk * 32u * b_row_b;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (tn_lds_ptr_t)(dst + i * 1024), 16, a_voff[i] + ao, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (tn_lds_ptr_t)(dst + CH_B + i * 1024), 16, b_voff[i] + bo, 0, 0, 0);
    }
  };

  // ---- transposed fragment reads (attention_pairs.hip, V^T): a 16-lane group gathers rows 4 h + q4 (+ 8) of a k-step, 16 columns of one
  // quarter; lane -> output index 16 g16 + (lane & 15) of the 32-column group, 8 reduction elements {4 h + (j & 3) + 8 (j >> 2)}
  const int h = lane >> 5, g16 = (lane >> 4) & 1, q4 = (lane >> 2) & 3, p4 = lane & 3;
  int fofs[4];   // [2 (column group of the wave) + plane]
#pragma unroll
  for (int q = 0; q < 4; ++q) fofs[q] = (4 * h + q4) * 512 + ((q ^ q4) << 6) + (16 * g16 + 4 * p4) * 2;
  const int a_base = wn * 256, b_base = CH_B + wk * 256;   // the wave's two column groups = quarters 4 w .. 4 w + 3 of the row (256 B)

  f32x16 a1[2][2], a2[2][2];   // [n group][k group]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { a1[i][j][e] = 0.f; a2[i][j][e] = 0.f; }

  if (c0 < c1) issue(c0, 0);
  for (int c = c0; c < c1; ++c) {
    const int buf = (c - c0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // chunk c has landed for every wave; every wave is done with the other buffer
    if (c + 1 < c1) issue(c + 1, buf ^ 1);
    const unsigned char* st = smem + buf * STAGE_B;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f16x8 af[2][2], bf[2][2];   // [group][plane]
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const unsigned char* pa = st + a_base + s * 16 * 512 + fofs[gq];
        const unsigned char* pb = st + b_base + s * 16 * 512 + fofs[gq];
        union { s16x4 s2[2]; f16x8 v; } ua, ub;
        ua.s2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa));
        ua.s2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa + 8 * 512));
        ub.s2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pb));
        ub.s2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pb + 8 * 512));
        af[gq >> 1][gq & 1] = ua.v;
        bf[gq >> 1][gq & 1] = ub.v;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          a1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], bf[j][0], a1[i][j], 0, 0, 0);
          a2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], bf[j][1], a2[i][j], 0, 0, 0);
          a2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][1], bf[j][0], a2[i][j], 0, 0, 0);
        }
    }
  }
  // ---- C[n][k]: lane -> column k = lane & 31 of the group, register e -> row n = (e & 3) + 8 (e >> 2) + 4 h: 128-byte row pieces
  float* out = g.C + (size_t)split * g.N * g.K;
  const int r = lane & 31;
  const float inv_s = g.a_scale ? 1.0f / *g.a_scale : 1.0f;   // exact: S is a power of two
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + wn * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int k = k0 + wk * 64 + j * 32 + r;
        out[(size_t)n * g.K + k] = fmaf(a2[i][j][e], kPairInvScale, a1[i][j][e]) * inv_s;
      }
}

int launch_splitk_reduce(const float* partial, float* out, long long n, int splits, long long stride, hipStream_t s);  // gemm_f32.hip
int launch_splitk_reduce_colfold(const float* partial, float* out, long long n, int splits, long long stride, const float* colpart, float* db,
                                 int colparts, int N, hipStream_t s);                                                   // gemm_f32.hip
int launch_colsum_fold(const float* partial, float* out, int chunks, int N, hipStream_t s);                            // rowops.hip

// splits of the m range: ~ 1.5 workgroups per CU (measured best of 0.25 .. 3, tools/tn_sweep.py: two per CU fit, but the partials'
// traffic - workgroups x 64 KB, written and read once - grows with the count), at least 8 chunks each, at most 64 partials
static int tn_splits(int N, int K, int M) {
  const int tiles = (N / 128) * (K / 128), nchunks = (M + 31) / 32;
  const int wgs = tuning_knob(KNOB_TN_WGS) > 0 ? tuning_knob(KNOB_TN_WGS) : 3 * device_cu_count() / 2;
  int s = (wgs + tiles - 1) / tiles;
  if (s > nchunks / 8) s = nchunks / 8;
  if (s > 64) s = 64;
  if (s >= 8 && tuning_knob(KNOB_TN_XCD) != 0) s = (s + 4) / 8 * 8 > nchunks / 4 ? s / 8 * 8 : (s + 4) / 8 * 8;   // a multiple of 8: whole splits per XCD
  return s < 1 ? 1 : s;
}

}  // namespace tt

using namespace tt;

extern "C" int tt_linear_bwd_weight_pairs_tn_ok(int N, int K, int M) {
  // 32-bit byte offsets into both operands, the last 32-row chunk included (rows M .. M + 31 are formed before the buffer's range check
  // zeroes them: ADVICE r4 - within 32 rows of 4 GiB the offset wrapped back INTO the buffer); the bound of the other pair kernels
  return N > 0 && K > 0 && M > 0 && N % 128 == 0 && K % 128 == 0 && ((long long)M + 32) * N * 4 < 0x7fffffffLL && ((long long)M + 32) * K * 4 < 0x7fffffffLL;
}
extern "C" size_t tt_linear_bwd_weight_pairs_tn_workspace_bytes(int N, int K, int M) {
  if (!tt_linear_bwd_weight_pairs_tn_ok(N, K, M)) return 0;
  const int s = tn_splits(N, K, M);
  return s > 1 ? (size_t)s * N * K * sizeof(float) : 16;
}
static int linear_bwd_weight_pairs_tn_impl(const void* dy_pairs, const void* x_pairs, float* dw, const float* dy_scale, int N, int K, int M,
                                           void* workspace, size_t workspace_bytes, const float* colsum_parts, int colsum_count, float* db,
                                           tt_stream_t stream) {
  TT_REQUIRE(dy_pairs && x_pairs && dw && workspace, "linear_bwd_weight_pairs_tn: null pointer");
  TT_REQUIRE(tt_linear_bwd_weight_pairs_tn_ok(N, K, M), "linear_bwd_weight_pairs_tn: need N %% 128 == 0, K %% 128 == 0, operands under 2 GB (N %d K %d M %d)", N, K, M);
  TT_REQUIRE(workspace_bytes >= tt_linear_bwd_weight_pairs_tn_workspace_bytes(N, K, M), "linear_bwd_weight_pairs_tn: workspace too small");
  TT_REQUIRE(aligned16(dy_pairs) && aligned16(x_pairs) && aligned16(dw) && aligned16(workspace), "linear_bwd_weight_pairs_tn: buffers must be 16-byte aligned");
  const int s = tn_splits(N, K, M);
  TnArgs g{static_cast<const _Float16*>(dy_pairs), static_cast<const _Float16*>(x_pairs), s > 1 ? static_cast<float*>(workspace) : dw, dy_scale, M, N, K,
           K / 128, (N / 128) * (K / 128), s, (M + 31) / 32, (s % 8 == 0 && tuning_knob(KNOB_TN_XCD) != 0) ? 1 : 0,
           colsum_parts, db, colsum_count};
  const int extra = (s == 1 && db) ? (N + 63) / 64 : 0;
  hipLaunchKernelGGL(gemm_pairs_tn_kernel, dim3((unsigned)(g.ntiles * s + extra)), dim3(256), 0, as_stream(stream), g);
  TT_CHECK_LAUNCH("gemm_pairs_tn");
  if (s == 1) return TT_OK;
  if (db)   // ONE launch folds the split partials of dw and the column partials of db (in colsum_stage2's order: the same bits)
    return launch_splitk_reduce_colfold(static_cast<const float*>(workspace), dw, (long long)N * K, s, (long long)N * K, colsum_parts, db, colsum_count, N,
                                        as_stream(stream));
  return launch_splitk_reduce(static_cast<const float*>(workspace), dw, (long long)N * K, s, (long long)N * K, as_stream(stream));
}
extern "C" int tt_linear_bwd_weight_pairs_tn(const void* dy_pairs, const void* x_pairs, float* dw, const float* dy_scale, int N, int K, int M,
                                             void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  return linear_bwd_weight_pairs_tn_impl(dy_pairs, x_pairs, dw, dy_scale, N, K, M, workspace, workspace_bytes, nullptr, 0, nullptr, stream);
}
extern "C" int tt_linear_bwd_weight_pairs_tn_bias(const void* dy_pairs, const void* x_pairs, float* dw, const float* dy_scale, int N, int K, int M,
                                                  void* workspace, size_t workspace_bytes, const float* colsum_parts, int colsum_count, float* db,
                                                  tt_stream_t stream) {
  TT_REQUIRE(colsum_parts && db && colsum_count > 0, "linear_bwd_weight_pairs_tn_bias: the column partials of dy (tt_split_pairs_dual_parts) and db are required");
  return linear_bwd_weight_pairs_tn_impl(dy_pairs, x_pairs, dw, dy_scale, N, K, M, workspace, workspace_bytes, colsum_parts, colsum_count, db, stream);
}
