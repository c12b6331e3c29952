// Lean instance of the f32-MFMA GEMM for the shapes that carry >90 % of a training step: every forward nn.Linear
// (y = act(x @ w^T + b) (+ residual)) with whole column tiles (N % BN == 0, K % 16 == 0, 16-byte aligned operands; any M -
// the last row tile clamps its loads and masks its stores).  Same algorithm and LDS image as gemm_f32.hip (k-major double buffer, ds_read_b32 fragments,
// v_mfma_f32_32x32x2_f32) but with everything the general kernel pays for stripped: no bounds checks, no layout
// switches, row pointers advanced instead of recomputed, and - the part that matters - the tile leaves through the
// idle LDS as 16-byte, fully coalesced row-major stores.  tools/mfma_peak.hip measures the inner loop at
// 133 TFLOP/s, 105 with the natural 64-scalar-stores-per-lane epilogue (store-issue bound) and 121-125 with this one.
// GATHER = 1 is the patch-embedding instance (the conv as a GEMM whose A rows are gathered from the image batch).
//
// Measured dead ends, kept out of the source (DESIGN.md section 5 has the numbers, git history the code):
//   * fragment reads hand-pipelined with inline-asm ds_read_b32 into a register double buffer and counted lgkmcnt waits
//     (hipcc folds any source-level double buffer back into "reads -> wait -> MFMAs"): +8 % on ZERO-filled operands,
//     nothing on random operands - the loop is not latency-limited;
//   * operands swapped (D^T = W X^T) so that a lane owns 4 consecutive output columns and stores 16 bytes straight from
//     the accumulators, no LDS pass: bit-identical, 1-2 % slower (32 rows x 32 B per store instruction);
//   * [row][k] LDS image with ds_write_b128 / ds_read_b128 (k-permuted fragments): bit-identical, -9 ... +2 % by shape;
//   * capping the workgroups per CU: flat from 6 down to 3, -10 % at 2, -26 % at 1 - the loop is not latency-bound;
//   * (round 2) LDS-DMA staging: global_load_lds_dwordx4 into a [row][16 k] image XOR-swizzled through the source address,
//     ds_read_b128 fragments with the k order permuted (k = 4 (2 j + h) + e), no staging VGPRs, no ds_write, ring of 2 / 3 / 4
//     slabs behind counted vmcnt + raw s_barrier (the structure of gemm_planes.hip): 886 / 895 / 948 us per ViT-S/16 block of
//     128 frames (qkv + proj + fc1 + fc2) against 834 us for this kernel on the same box (tools/bench_linear.py), same 5e-7
//     error.  Three very different loop structures (this one, the DMA ring, hipBLASLt's stream-K) land within +-8 % of
//     110 TFLOP/s at K = 384: the ceiling is not in the staging path;
//   * (round 2) s_setprio 1 / 2 for the main loop (epilogue at 0): no difference (interleaved A/B, all four block shapes);
//   * (round 2) BK = 32 slabs on the large grids (half the barriers, twice the staging registers and LDS): 899 us per block
//     against 810 us at BK = 16 - the extra LDS drops a workgroup per CU.  BK = 64 is kept only for grids <= 320 tiles, where
//     a CU holds one workgroup anyway;
//   * (round 2) non-temporal epilogue (nt stores of C, nt loads of the residual) to keep the streamed output out of L2, where
//     the weights and the A row blocks live: qkv / fc1 -0.7 %, fc2 +1.0 %, proj +5.6 % (interleaved A/B) - nothing, and in the
//     step the next kernel WANTS the output in the caches;
//   * (round 2) the compiler's other scheduling strategies (-mllvm -amdgpu-sched-strategy=max-ilp / max-memory-clause /
//     iterative-ilp, -amdgpu-schedule-metric-bias=100): all four block shapes within 1 % of the default;
//   * (round 2) a PERSISTENT form: 256 x occupancy workgroups that walk tiles handed out dynamically (per-XCD atomic counters, so
//     that the tile quantisation of a static round-robin does not eat the gain and the tiles sharing an A row block stay on one
//     L2) and load the first slab of their next tile into registers during the last slab of the current one, so that the epilogue
//     overlaps that latency (the structure tools/mfma_peak.hip measures at 123-125 against 116-119 TFLOP/s for one workgroup per
//     tile): correct on the full matrices, and 7 % SLOWER on the shapes it applies to (qkv 199 -> 214 us, fc1 279 -> 298 us; with the
//     atomic's round trip hidden under the first slabs as well).  68 VGPRs + the accumulators allow 4 resident workgroups per CU
//     against 6, and with 5-6 short-lived workgroups per CU the hardware's own dispatch already overlaps one workgroup's prologue
//     and epilogue with the others' main loops.
#include "common.hpp"
#include <cstdlib>

namespace tt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FastArgs {
  const float* A;   // [M][K]
  const float* B;   // [N][K]
  float* C;         // [M][N]
  int M, N, K;
  const float* bias;      // [N] or null
  const float* residual;  // [M][N] or null (may alias C)
  float* pre_out;         // [M][N] or null
  int act;                // 1 = GELU
  // GATHER instance (patch embedding, dino_vision_transformer.py:156-171 + 236-247): A row m = patch (m % n_patch) of frame
  // frame_map[m / n_patch] of the image batch A = img [F_src][Cin][H][W], k = (channel, row in patch, pixel in row); P = 16, so a
  // 16-k slab is one contiguous pixel row.  The epilogue adds the position rows and writes token row f (n_patch + 1) + 1 + patch.
  const int* frame_map;   // [M / n_patch] or null (identity)
  const float* pos;       // [n_patch + 1][N]
  int Cin, H, W, gw, n_patch;
};

#ifdef TT_CLOCK_STAMP
// Diagnostic build only (tools/gemm_clock.py): every workgroup stamps the shader clock (s_memtime) and the constant
// 100 MHz clock (s_memrealtime) around its main loop; their ratio is the clock the chip sustained while this kernel ran
// (MI355X_MICROARCH.md "DVFS give-back" item 6).  The shipped library has no stamp.
__device__ unsigned long long tt_clock_stamps[6 * 8192];
extern "C" int tt_debug_read_clock_stamps(unsigned long long* host, int count) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(tt_clock_stamps), sizeof(unsigned long long) * count);
}
#endif

// (Round 3, measured and not kept: the 64 x 64 small-grid instance with FOUR 64-deep slabs in flight in registers - 236 VGPRs, every
// slab of a K = 384 product requested up front - made BASELINE C1's step SLOWER, 3.17 against 2.91 ms in two interleaved pairs of runs:
// these launches are not bound by the slab round trips alone.)
// BK = 16 is the throughput instance (5-6 workgroups per CU hide every latency).  BK = 64 is for grids of at most about one
// workgroup per CU (BASELINE C1: 2 x 2 frames = 788 rows, 13 row tiles): there nothing hides the global-load latency of the
// one slab in flight, a launch is (K / BK) dependent round trips long, and four times deeper slabs cut the trips four-fold.
template <int WM, int WN, int BK = 16, int GATHER = 0>
__global__ __launch_bounds__(256) void gemm_nt_fast_kernel(FastArgs g) {
  static_assert(!GATHER || BK == 16, "the patch gather stages one pixel row (16 k) per slab");
  constexpr int BM = 64 * WM, BN = 64 * WN, KQ = BK / 16;
  // k-row stride = tile extent + 2: the transposing staging writes (lane -> k-rows 4 (tid & 3) + e, column tid >> 2) then
  // spread over all 32 banks of a ds_write_b32 lane group (4 * stride = 8 mod 32); with + 4 (= 16 mod 32) they collide
  // two-way on every write (SQ_LDS_BANK_CONFLICT was 24 % of the LDS-active cycles).  Fragment reads walk consecutive
  // columns of one k-row and are conflict-free for any stride.
  constexpr int LDA = BM + 2, LDB = BN + 2;
  constexpr int ASZ = BK * LDA, BSZ = BK * LDB;
  constexpr int PIPE_FLOATS = 2 * (ASZ + BSZ), EPI_FLOATS = 32 * WM * (BN + 4);
  __shared__ __attribute__((aligned(16))) float lds[PIPE_FLOATS > EPI_FLOATS ? PIPE_FLOATS : EPI_FLOATS];

#ifdef TT_CLOCK_STAMP
  const unsigned long long st_entry = __builtin_amdgcn_s_memrealtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int ntn = g.N / BN, ntm = (g.M + BM - 1) / BM;   // the last row tile may be partial: loads clamp, stores mask
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  const int K = g.K;

  // staging: thread owns row (tid >> 2) + 64 i of each operand tile and the 4 k's at (tid & 3) * 4
  const int srow = tid >> 2, skc = (tid & 3) * 4;
  const float* pa[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    int row = m0 + srow + 64 * i;
    row = row < g.M ? row : g.M - 1;
    if constexpr (GATHER) {
      const int f = row / g.n_patch, pi = row - f * g.n_patch;
      const int fs = g.frame_map ? g.frame_map[f] : f;
      const int pr = pi / g.gw, pc = pi - pr * g.gw;
      pa[i] = g.A + ((size_t)fs * g.Cin * g.H + (size_t)pr * 16) * g.W + pc * 16 + skc;
    } else {
      pa[i] = g.A + (size_t)row * K + skc;
    }
  }
  [[maybe_unused]] int gather_py = 0;   // pixel row inside the patch of the NEXT slab to load
  const float* pb = g.B + (size_t)(n0 + srow) * K + skc;
  const size_t step64 = (size_t)64 * K;
  // a thread stages, per 16 k, the 4 k's at skc of its row(s); a BK = 64 slab is four such quarters (q)
  float4 ra[KQ][WM], rb[KQ][WN];
  auto gload = [&]() {
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
#pragma unroll
      for (int i = 0; i < WM; ++i) ra[q][i] = *reinterpret_cast<const float4*>(pa[i] + 16 * q);
#pragma unroll
      for (int i = 0; i < WN; ++i) rb[q][i] = *reinterpret_cast<const float4*>(pb + i * step64 + 16 * q);
    }
    if constexpr (GATHER) {   // next pixel row of the patch; after the 16th, the first row of the next channel
      const size_t adv = (++gather_py == 16) ? (size_t)g.H * g.W - (size_t)15 * g.W : (size_t)g.W;
      if (gather_py == 16) gather_py = 0;
#pragma unroll
      for (int i = 0; i < WM; ++i) pa[i] += adv;
    } else {
#pragma unroll
      for (int i = 0; i < WM; ++i) pa[i] += BK;
    }
    pb += BK;
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      float* da = lds + buf * ASZ + (16 * q + skc) * LDA + srow;
      float* db = lds + 2 * ASZ + buf * BSZ + (16 * q + skc) * LDB + srow;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        da[0 * LDA + 64 * i] = ra[q][i].x;
        da[1 * LDA + 64 * i] = ra[q][i].y;
        da[2 * LDA + 64 * i] = ra[q][i].z;
        da[3 * LDA + 64 * i] = ra[q][i].w;
      }
#pragma unroll
      for (int i = 0; i < WN; ++i) {
        db[0 * LDB + 64 * i] = rb[q][i].x;
        db[1 * LDB + 64 * i] = rb[q][i].y;
        db[2 * LDB + 64 * i] = rb[q][i].z;
        db[3 * LDB + 64 * i] = rb[q][i].w;
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

  const int nk = K / BK;
  gload();
  sstore(0);
  __syncthreads();
#ifdef TT_CLOCK_STAMP
  const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload();
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

#ifdef TT_CLOCK_STAMP
  if (tid == 0 && blockIdx.x < 8192) {
    unsigned long long* st = tt_clock_stamps + 6 * blockIdx.x;
    st[0] = st_c0; st[1] = st_r0; st[2] = __builtin_amdgcn_s_memtime(); st[3] = __builtin_amdgcn_s_memrealtime();
    st[4] = st_entry;
  }
#endif
  // ---- epilogue through LDS: one wave-row (32 * WM tile rows) at a time
  constexpr int CH = 32 * WM, LDCS = BN + 4, TPR = BN / 4, RPP = 256 / TPR;
  static_assert(CH * LDCS == EPI_FLOATS, "epilogue staging is sized with the pipeline buffers");
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
            lds[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * LDCS + wn * (32 * WN) + j * 32 + r] = acc[i][j][e];
    }
    __syncthreads();
    for (int rr = tid / TPR; rr < CH; rr += RPP) {
      if (m0 + wmi * CH + rr >= g.M) break;
      size_t off = (size_t)(m0 + wmi * CH + rr) * g.N + n;
      float4 v = *reinterpret_cast<const float4*>(lds + rr * LDCS + c4);
      v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
      if constexpr (GATHER) {
        const int m = m0 + wmi * CH + rr, f = m / g.n_patch, pi = m - f * g.n_patch;
        off = ((size_t)f * (g.n_patch + 1) + 1 + pi) * g.N + n;
        const float4 ps = *reinterpret_cast<const float4*>(g.pos + (size_t)(1 + pi) * g.N + n);
        v.x += ps.x; v.y += ps.y; v.z += ps.z; v.w += ps.w;
      }
      if (g.pre_out) *reinterpret_cast<float4*>(g.pre_out + off) = v;
      if (g.act == 1) {
        // (exact-erf GELU as nn.GELU; an Abramowitz-Stegun erf is NOT faster here: 265.4 vs 266.8 us on the fc1 shape in an
        // interleaved A/B, tools/ab_linear.py - the epilogue's VALU work hides under the other workgroups' MFMAs)
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
#ifdef TT_CLOCK_STAMP
  if (tid == 0 && blockIdx.x < 8192) tt_clock_stamps[6 * blockIdx.x + 5] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <int WM, int WN>
static int launch_fast(const FastArgs& g, hipStream_t s) {
  const int tiles = ((g.M + 64 * WM - 1) / (64 * WM)) * (g.N / (64 * WN));
  // tuning aid: TT_GEMM_DYNLDS=<bytes> adds unused dynamic LDS to every launch, which caps the workgroups per CU
  static const int dyn_lds = [] { const char* e = getenv("TT_GEMM_DYNLDS"); return e ? atoi(e) : 0; }();
  static const int small_grid = [] { const char* e = getenv("TT_GEMM_SMALL_GRID"); return e ? atoi(e) : 320; }();  // tuning aid
  if (tiles <= small_grid && g.K % 64 == 0) {   // latency-bound grid: deep slabs
    hipLaunchKernelGGL((gemm_nt_fast_kernel<WM, WN, 64>), dim3(tiles), dim3(256), dyn_lds, s, g);
    TT_CHECK_LAUNCH("gemm_nt_fast");
    return TT_OK;
  }
  hipLaunchKernelGGL((gemm_nt_fast_kernel<WM, WN>), dim3(tiles), dim3(256), dyn_lds, s, g);
  TT_CHECK_LAUNCH("gemm_nt_fast");
  return TT_OK;
}

int gemm_tile_choice(int M, int N, int batch);

// Patch embedding on the lean kernel (P = 16, D % 64 == 0, 16-byte aligned rows): TT_OK if launched, 1 if not eligible.
int try_launch_patch_embed_fast(const float* img, const int* frame_map, const float* w, const float* bias, const float* pos, float* tokens,
                                int F, int C, int H, int W, int P, int D, hipStream_t s) {
  if (P != 16 || D % 64 != 0 || W % 4 != 0 || !aligned16(img) || !aligned16(w) || !aligned16(bias) || !aligned16(pos) || !aligned16(tokens))
    return 1;
  const int gw = W / P, n = gw * (H / P);
  FastArgs g{img, w, tokens, F * n, D, C * P * P, bias, nullptr, nullptr, 0, frame_map, pos, C, H, W, gw, n};
  const int tile = gemm_tile_choice(g.M, D, 1);
  const bool bm128 = tile == 0 || tile == 2;
  const int tiles = ((g.M + (bm128 ? 127 : 63)) / (bm128 ? 128 : 64)) * (D / 64);
  if (bm128) hipLaunchKernelGGL((gemm_nt_fast_kernel<2, 1, 16, 1>), dim3(tiles), dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_nt_fast_kernel<1, 1, 16, 1>), dim3(tiles), dim3(256), 0, s, g);
  TT_CHECK_LAUNCH("patch_embed_fast");
  return TT_OK;
}

// Returns TT_OK if launched, 1 if the shape is not eligible (caller falls back to the general kernel).
int try_launch_gemm_nt_fast(const float* A, const float* B, float* C, int M, int N, int K, const float* bias,
                            const float* residual, float* pre_out, int act, hipStream_t s) {
  auto ok16 = [](const void* p) { return p == nullptr || aligned16(p); };
  if (K % 16 != 0 || K < 16 || !aligned16(A) || !aligned16(B) || !aligned16(C) || !ok16(bias) || !ok16(residual) || !ok16(pre_out)) return 1;
  FastArgs g{A, B, C, M, N, K, bias, residual, pre_out, act, nullptr, nullptr, 0, 0, 0, 0, 0};
  const int tile = gemm_tile_choice(M, N, 1);
  const int bn = (tile == 0 || tile == 1) ? 128 : 64;
  if (N % bn != 0) {   // (any M: a partial last row tile is clamped / masked)
    if (N % 64 == 0) return launch_fast<1, 1>(g, s);
    return 1;
  }
  switch (tile) {
    case 0: return launch_fast<2, 2>(g, s);
    case 1: return launch_fast<1, 2>(g, s);
    case 2: return launch_fast<2, 1>(g, s);
    default: return launch_fast<1, 1>(g, s);
  }
}

}  // namespace tt
