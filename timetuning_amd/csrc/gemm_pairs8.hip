// fp16-PAIR GEMM, persistent form - the forward nn.Linear sites (dino_vision_transformer.py:94-103,115-130; models.py:915-926):
//     y = act(x @ w^T + bias) (+ residual)
// in the fp32-accurate split mode "f16x3" (round 4): an fp32 operand x is resident in HBM as a PAIR of fp16 numbers
//     hi = fp16(x),   lo = fp16((x - hi) * 2^11)        x = hi + lo * 2^-11 to <= 2^-23 relative (11 + 1 + 11 significant bits)
// and a product term x w costs THREE v_mfma_f32_32x32x16_f16 - hi hi into one accumulator, hi lo + lo hi into a SECOND one that is
// folded in with the exact factor 2^-11 in the epilogue (the dropped lo lo term is <= 2^-22 |x w|, ~4e-8 of it in the mean) - where the
// three-bf16-plane split (gemm_planes8.hip, P = 3) needs six.  Scaling lo by 2^11 keeps it a NORMAL fp16 number wherever hi is one,
// so the split needs no per-tensor scale and no subnormal arithmetic; operands must lie in fp16's range (|x| <= 65504; beyond it hi is
// an infinity and the result says so).  The MFMA keeps fp16 subnormals and has the bf16 instruction's lane maps
// (tools/probes/mfma_f16_probe.hip, run on the box).
//
// Memory format ("pairs"): groups of 32 consecutive elements of a row, [hi x 32][lo x 32] fp16 = 128 bytes per group - a row of C
// elements is 4 C bytes, as in fp32, and the 32-deep K-tile of a row is ONE 128-byte line.
//
// Kernel structure: that of gemm_planes8.hip (one 8-wave workgroup per CU, two wave groups a barrier interval apart alternating a
// load part and an MFMA part, LDS-DMA in flight across raw barriers behind counted vmcnt, persistent over work items with the DMA
// cursor running ahead across item boundaries, half tiles for the remainder round, swapped operands, per-wave epilogue through private
// scratch with stores that are not waited for) on a geometry that fits TWO accumulator sets:
//   * tile 256 (x rows) x 128 (w rows) x 32: a K-tile is three 16 KB chunks - W, X0, X1 (128 rows of 128 bytes each) - and the LDS
//     holds a ring of THREE K-tiles (144 KB) + 2 KB of scratch per wave;
//   * TWO phases per K-tile (x half 0, x half 1) of 12 MFMAs each (2 MFMA tiles x 2 k-steps x 3 products = 384 cycles of matrix pipe
//     against a load part of 12 / 8 fragment reads and 2 / 4 LDS-DMA instructions per wave): the W fragments of a K-tile are read in
//     phase 0 and stay in registers for phase 1;
//   * schedule: phase (t, 0) issues chunk X1 of K-tile t + 1, phase (t, 1) chunks W and X0 of K-tile t + 2; every wait leaves the
//     youngest three chunks (6 wave-instructions) in flight.  RAW: a chunk is needed three phases after its issue and is retired by the
//     wait of the phase before.  WAR: slot (t + 2) mod 3 was last read in phase (t - 1, 0), slot part X1 of (t + 1) mod 3 in phase
//     (t - 2, 1): three phases before the DMA that overwrites them (two are required with the groups staggered).
//
// Why not the 256 x 256 tile of the P = 1 kernel: two accumulator sets of it are 256 registers per lane, the whole budget of a wave at
// two waves per SIMD.
#include "common.hpp"
#include <cstdlib>

namespace tt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* q8_lds_ptr_t;

struct Q8Args {
  const _Float16* X;   // pairs [M][2 K]
  const _Float16* W;   // pairs [N][2 K]
  int M, N, K;         // K: reduction length in elements (K % 96 == 0)
  const float* bias;      // [N] or null
  const float* residual;  // [M][N] (Q8_F32_RES: added, may alias C; Q8_F32_GELUGRAD: the pre-activation whose gelu' multiplies the result)
  float* C;               // [M][N] fp32 (Q8_F32*, Q8_BOTH: y; Q8_BOTH_GELU: the pre-activation)
  _Float16* Cp;           // pairs [M][2 N] (Q8_PAIR*, Q8_BOTH*)
  int ntn, ntiles, ncu;   // column tiles, whole tiles, workgroups launched
  int n_full, n_half;     // as gemm_planes8.hip: n_full whole tiles per workgroup, then n_half half tiles; both 0: round-robin
  int order_mode;         // order of the load part (see `reads_first`)
  // K-split of the tiles beyond the last whole round (ks_S >= 2; then n_half == 0): tile n_full * ncu + j, j < ks_R, is computed by the
  // ks_S workgroups cu = j * ks_S + s, each over a contiguous range of K-tile triples; every WAVE leaves the fp32 partial of its 64 x 64
  // sub-tile in ks_ws, and the wave that arrives last at the (tile, wave) counter adds the partials in slice order and runs the epilogue
  int ks_S, ks_R;
  const float* out_scale; // device scalar S (a power of two) one operand was scaled by before its split (a gradient) - the product is
                          // divided by it; null: none
  float* ks_ws;           // [ks_R][ks_S][8 waves][4096]
  int* ks_cnt;            // [ks_R][8], zero between launches (the finishing wave resets its counter)
  int* range_flag;        // device word or null: set to 1 when a pair output's hi leaves fp16's range (common.hpp: pair_hi_bad)
  float* amax_out;        // device float or null (Q8_F32_GELUGRAD, the symmetric kernel): max |C| is published into it (common.hpp: amax_publish)
};

__device__ __forceinline__ void q8_dma16(const void* base, unsigned char* lds_dst, int voffset, int soffset) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (q8_lds_ptr_t)lds_dst, 16, voffset, soffset, 0, 0);
}
__device__ __forceinline__ f32x4 q8_ld128(const void* base, unsigned nbytes, unsigned voff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, nbytes, 0x00020000);
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ void q8_st128(void* base, unsigned nbytes, unsigned voff, u32x4 v) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, nbytes, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 0);
}
// a store with a cache-policy operand (aux: bit 0 sc0, bit 1 nt, bit 4 sc1)
template <int AUX>
__device__ __forceinline__ void q8_st128_aux(void* base, unsigned nbytes, unsigned voff, u32x4 v) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, nbytes, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, AUX);
}
// sc0 sc1 accesses (aux bits 0 and 4): write-through / read-around of the XCD's L2 - how the K-split partials travel between workgroups
// that may sit on different XCDs without an L2 write-back + invalidate per wave (measured: agent-scope fences cost ~75 us per launch)
__device__ __forceinline__ f32x4 q8_ld128_sys(const void* base, unsigned nbytes, unsigned voff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, nbytes, 0x00020000);
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 17));
}
__device__ __forceinline__ void q8_st128_sys(void* base, unsigned nbytes, unsigned voff, u32x4 v) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, nbytes, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 17);
}
template <int N>
__device__ __forceinline__ void q8_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// TT_Q8_TAIL (timing study, round 4): 1 = the LDS-DMA instructions of phase f + 1 are issued behind the MFMAs of phase f (the MFMA
// part's idle tail: the kernel is bound by its load part, see profiles/r04_q8_ablation.txt) instead of at the head of phase f + 1's
// load part - half a phase earlier, same order, same counted waits; 2 = in the middle of the MFMAs.
#ifndef TT_Q8_TAIL
#define TT_Q8_TAIL 0
#endif
#if TT_Q8_TAIL != 0
#error "TT_Q8_TAIL was a round-4 timing study (profiles/r04_q8_dma_placement_ab.txt: no gain / 2x slower); the half-item schedule no longer supports it"
#endif
// TT_Q8_SQ: the wave tile.  0 = 128 (x) x 32 (w) - waves 2 x 4, a K-tile costs a wave 4 + 8 + 8 = 20 fragment reads; 1 = 64 x 64 - waves
// 4 x 2, 8 (W, kept for both phases) + 4 + 4 = 16 reads for the same 24 MFMAs: the load part is the long part of a phase and the LDS
// the busiest unit of the CU (160 KB of fragment reads + 48 KB of DMA writes per K-tile at 128 B / clk against 1536 cycles of MFMAs).
#ifndef TT_Q8_SQ
#define TT_Q8_SQ 1
#endif
// DBG (timing studies only; the shipped instantiations are DBG = 0), a bit mask: 1 no MFMAs, 2 no LDS-DMA, 8 no epilogue
template <int EPI, int DBG = 0>
__global__ __launch_bounds__(512) void gemm_pairs8_kernel(Q8Args g) {
  constexpr int ROWB = 128;                 // bytes per LDS row = one pair group
  constexpr int CPR = 8;                    // 16-byte chunks per row
  constexpr int WIN = 2;                    // rows per 256-byte bank window
  constexpr int RPI = 8;                    // rows one LDS-DMA wave-instruction fills
  constexpr int JPW = 2;                    // DMA wave-instructions per wave and chunk (128 rows / 8 / 8 waves)
  constexpr int GCH = JPW;                  // the vmcnt unit: wave-instructions per wave and chunk
  constexpr int CHUNK_B = 128 * ROWB;       // 16 KB
  constexpr int SLOT_B = 3 * CHUNK_B;       // W, X0, X1
  constexpr int RING_B = 3 * SLOT_B;        // three K-tiles
  constexpr int SCR_B = (160 * 1024 - RING_B) / 8;   // per-wave epilogue scratch (2 KB)
  constexpr int CW = SCR_B / 128;                    // columns of a 32-row MFMA tile staged per pass (16)
  constexpr int NPASS = 32 / CW;
  constexpr int BN = 128;
  constexpr bool F32OUT = EPI == Q8_F32 || EPI == Q8_F32_RES || EPI == Q8_F32_GELUGRAD;   // the 4-columns-per-lane read-back
  constexpr bool RES = EPI == Q8_F32_RES || EPI == Q8_F32_GELUGRAD;                       // an [M][N] fp32 operand, prefetched two tiles ahead
  constexpr bool GG = EPI == Q8_F32_GELUGRAD;
  constexpr bool BOTH = EPI == Q8_BOTH || EPI == Q8_BOTH_GELU;                            // pairs AND fp32
  constexpr bool ACT = EPI == Q8_PAIR_GELU || EPI == Q8_BOTH_GELU;
  constexpr int L = 3;                                // chunks a wait leaves in flight
  constexpr int WFULL = L * GCH;                      // 6
  constexpr int WGUARD = 2 * GCH;                     // a window that may hold a half tile's K-tile (no X1 chunk): W, X0 only
  constexpr int ST_TILE = BOTH ? 8 : 4;               // stores a wave issues per 32 x 32 MFMA tile (fp32: 2 passes x 2; pairs: 2 passes x (hi, lo))
  constexpr int ST_FULL = 4 * ST_TILE, ST_HALF = ST_FULL / 2;
  constexpr int S_FULL = ST_FULL < 63 - WFULL ? ST_FULL : 63 - WFULL, S_HALF = ST_HALF < 63 - WFULL ? ST_HALF : 63 - WFULL;
  constexpr int POST = TT_Q8_TAIL ? 3 : 2;            // phases whose window still reaches back across an epilogue
  static_assert(RING_B + 8 * SCR_B <= 160 * 1024 && CW == 16, "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char smem[160 * 1024];

#ifdef TT_Q8_CLOCK   // diagnostic build only: the clock the chip holds under this kernel (s_memtime ticks per 100 MHz s_memrealtime tick)
  const unsigned long long clk0_t = __builtin_amdgcn_s_memtime(), clk0_r = __builtin_amdgcn_s_memrealtime();
#endif
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool grp1 = wave >= 4;
  constexpr bool SQ = TT_Q8_SQ != 0;
  const int wr = SQ ? (wave & 3) : (wave >> 2), wc = SQ ? (wave >> 2) : (wave & 3);   // SQ: 4 x 2 waves of 64 x 64; else 2 x 4 of 128 x 32
  const int r = lane & 31, h = lane >> 5;
  const int K4 = g.K * 4, nk = g.K / 32;   // bytes per operand row; K-tiles

  // ---- this workgroup's work items (gemm_planes8.hip: whole tiles dealt round-robin so that the 32 workgroups of an XCD work on
  // 32 consecutive tiles; the tiles beyond the last whole round cut into halves)
  int cu = blockIdx.x;
  if ((g.ncu & 7) == 0) cu = (blockIdx.x & 7) * (g.ncu >> 3) + (blockIdx.x >> 3);
  int n_whole;
  bool has_half = false;
  const bool ksplit = g.ks_S >= 2;
  if (g.n_full > 0 || g.n_half > 0 || ksplit) {
    n_whole = g.n_full;
    has_half = !ksplit && cu < g.n_half;
  } else {
    n_whole = cu < g.ntiles ? (g.ntiles - cu + g.ncu - 1) / g.ncu : 0;
  }
  const bool has_slice = ksplit && cu < g.ks_R * g.ks_S;   // the last item: a K range of one of the left-over tiles
  const int n_items = n_whole + (has_half || has_slice ? 1 : 0);
  if (n_items == 0) return;   // whole workgroup
  const bool half_first = has_half && (cu & 1) && n_whole > 0;
  // item -> output tile origin, half flag, K-tile range [kt0, kend) (a multiple of 3 K-tiles long)
  auto item = [&](int it, int& row0, int& n0, bool& half, int& kt0, int& kend) {
    int tile;
    kt0 = 0;
    kend = nk;
    half = has_half && (half_first ? it == 0 : it == n_whole);
    int hsel = 0;
    if (half) {
      tile = g.n_full * g.ncu + (cu >> 1);
      hsel = cu & 1;
    } else if (has_slice && it == n_whole) {
      const int j = cu / g.ks_S, sl = cu - j * g.ks_S, U = nk / 3;
      tile = g.n_full * g.ncu + j;
      kt0 = 3 * (sl * U / g.ks_S);
      kend = 3 * ((sl + 1) * U / g.ks_S);
    } else {
      tile = (half_first ? it - 1 : it) * g.ncu + cu;
    }
    const int mb = tile / g.ntn, ns = tile - mb * g.ntn;
    row0 = mb * 256 + hsel * 128;
    n0 = ns * BN;
  };

  // ---- LDS-DMA lane map: lane -> (row, slot) of a 1 KiB piece (8 rows), source chunk = slot ^ f(row), f(row) = (row / 2) & 7
  const int l_row = lane / CPR, l_slot = lane % CPR;
  const int d_row0 = wave * RPI + l_row;                                  // image row of piece `wave`; piece wave + 8 adds 64 rows (same f)
  const int d_chunk = l_slot ^ ((d_row0 / WIN) & (CPR - 1));
  const int w_voff = d_row0 * K4 + d_chunk * 16;

  // DMA cursor: the K-tile whose chunks are issued next (scalar state + the X voffsets of its item; rows beyond M are clamped)
  int d_item = 0, d_kt = 0, d_kend = 0;
  bool d_done = false, d_half = false;
  int half_guard = 0;   // > 0: the DMA window may hold a K-tile of a half item: the counted waits take the smaller count
  int d_kofs = 0;       // d_kt * ROWB
  int d_wbase = 0;      // first W row of the item * K4
  int x_voff[2][JPW];
  auto cursor_item = [&]() {
    int row0, n0;
    item(d_item, row0, n0, d_half, d_kt, d_kend);
    d_kofs = d_kt * ROWB;
    d_wbase = n0 * K4;
#pragma unroll
    for (int ha = 0; ha < 2; ++ha)
#pragma unroll
      for (int i = 0; i < JPW; ++i) {
        int row = row0 + ha * 128 + 8 * i * RPI + d_row0;
        row = row < g.M ? row : g.M - 1;
        x_voff[ha][i] = row * K4 + d_chunk * 16;
      }
  };
  cursor_item();
  auto cursor_next_ktile = [&]() {
    ++d_kt;
    d_kofs += ROWB;
    if (d_kt == d_kend) {
      ++d_item;
      if (d_item >= n_items) {
        d_done = true;
      } else {
        cursor_item();
      }
    }
  };
  bool steady = false;
  // issue chunk C (0 W, 1 X0, 2 X1) of the cursor's K-tile into ring slot S
  auto issue = [&](auto c_c, auto s_c) {
    constexpr int CI = decltype(c_c)::value, S = decltype(s_c)::value;
    if constexpr (!(DBG & 2)) {
      if (!steady) {
        if (d_done) return;
        if (CI == 2 && d_half) return;
      }
      const int lds_base = S * SLOT_B + CI * CHUNK_B + wave * 1024;
#pragma unroll
      for (int i = 0; i < JPW; ++i) {
        unsigned char* dst = smem + lds_base + i * 8 * 1024;
        if constexpr (CI == 0)
          q8_dma16(g.W, dst, w_voff, d_wbase + (8 * i * RPI) * K4 + d_kofs);
        else
          q8_dma16(g.X, dst, x_voff[CI - 1][i], d_kofs);
      }
    }
  };

  // ---- fragment addressing: image row = (slice of the chunk) + r; chunk of the row: hi of k-step ks = 2 ks + h, lo = 4 + 2 ks + h;
  // swizzled by f(r) (slices are multiples of 32 rows)
  const int f_sw = (r / WIN) & (CPR - 1);
  int fo[4];   // 0, 1: hi of k-steps 0, 1; 2, 3: lo
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) fo[c4] = r * ROWB + (((2 * c4 + h) ^ f_sw) << 4);
  const int x_slice = wr * (SQ ? 32 : 64) * ROWB, w_slice = wc * (SQ ? 64 : 32) * ROWB;   // within a chunk (X0 / X1: per x half)

  f32x16 a1[2][2], a2[2][2];   // [x half][MFMA tile: SQ the w 32-row block, else the x 32-row block]: hi hi | hi lo + lo hi (x 2^11)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int e = 0; e < 16; ++e) { a1[a][m][e] = 0.f; a2[a][m][e] = 0.f; }
  f16x8 One[4], Two[2][4];   // fragments [hi k-step 0, 1, lo k-step 0, 1]: One = the operand with ONE 32-row block per phase (W; SQ: X), Two = two blocks

  // order of the load part (gemm_planes8.hip `reads_first`): 0 every wave DMA first, 1 every wave reads first, 2 odd waves read first,
  // 3 waves 2, 3 (6, 7) of a group read first
  const int omode = g.order_mode % 100;   // (+100: diagnostic builds print their stamps)
  const bool reads_first = omode == 1 || (omode == 2 && (wave & 1)) || (omode == 3 && (wave & 2));
  int post_epi = 0;          // phases left in which the stores of the last epilogue may still be outstanding
  bool post_half = false;
  bool c_half = false;       // the item being computed is a half tile

#ifdef TT_Q8_STAMP   // diagnostic build only (tools/q8_stamp.py): where a wave's steady-state phase goes, s_memtime stamps
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0, ts6 = 0, tsl = 0;
  unsigned st_a[2] = {0, 0}, st_b[2] = {0, 0}, st_wait[2] = {0, 0}, st_bar1[2] = {0, 0}, st_lgkm[2] = {0, 0}, st_mfma[2] = {0, 0}, st_bar2[2] = {0, 0}, st_n[2] = {0, 0};
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#define Q8_STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
#define Q8_STAMP_NOWAIT(t) asm volatile("s_memtime %0" : "=s"(t)::"memory")
#else
#define Q8_STAMP(t)
#define Q8_STAMP_NOWAIT(t)
#endif
  // ---- one phase: [DMA issue | fragment reads | counted wait] barrier [MFMAs] barrier
  auto phase = [&](auto s_c, auto ha_c) {
    constexpr int S = decltype(s_c)::value, HA = decltype(ha_c)::value;
    constexpr int base = S * SLOT_B;
    // a half item has no second x half: its (t, 1) phases only issue DMA and synchronise.  (Dropping them altogether - one phase per
    // K-tile - was tried in round 4 and is a WAR race on the 3-slot ring: consecutive K-tiles would be ONE phase apart, and the DMA into
    // slot (t + 2) mod 3 must trail the last read of K-tile t - 1 by two.)
    const bool work = !(HA == 1 && c_half);
    auto frag_reads = [&]() {
      if (work) {
        if constexpr (SQ) {   // W: two blocks, read in phase 0 and kept; X: one block per phase
          if constexpr (HA == 0) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int c4 = 0; c4 < 4; ++c4) Two[mt][c4] = *reinterpret_cast<const f16x8*>(smem + base + w_slice + mt * 32 * ROWB + fo[c4]);
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) One[c4] = *reinterpret_cast<const f16x8*>(smem + base + (1 + HA) * CHUNK_B + x_slice + fo[c4]);
        } else {
          if constexpr (HA == 0) {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) One[c4] = *reinterpret_cast<const f16x8*>(smem + base + w_slice + fo[c4]);
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4)
              Two[mt][c4] = *reinterpret_cast<const f16x8*>(smem + base + (1 + HA) * CHUNK_B + x_slice + mt * 32 * ROWB + fo[c4]);
        }
      }
    };
    auto dma_issue = [&]() {
      if constexpr (HA == 0) {
        issue(std::integral_constant<int, 2>{}, std::integral_constant<int, (S + 1) % 3>{});   // X1 of K-tile t + 1 (the cursor's)
      } else {
        cursor_next_ktile();                                                                     // -> K-tile t + 2
        issue(std::integral_constant<int, 0>{}, std::integral_constant<int, (S + 2) % 3>{});
        issue(std::integral_constant<int, 1>{}, std::integral_constant<int, (S + 2) % 3>{});
      }
    };
    auto dma_wait = [&]() {
      if (steady && !(DBG & 2)) {
        q8_wait_vmcnt<WFULL>();
      } else {
        half_guard = (d_half && !d_done) ? 4 : (half_guard > 0 ? half_guard - 1 : 0);
        if (d_done || (DBG & 2)) {
          q8_wait_vmcnt<0>();
        } else if (post_epi > 0) {
          --post_epi;
          if (half_guard > 0 || post_half) q8_wait_vmcnt<WGUARD + S_HALF>();
          else q8_wait_vmcnt<WFULL + S_FULL>();
        } else if (half_guard > 0) {
          q8_wait_vmcnt<WGUARD>();
        } else {
          q8_wait_vmcnt<WFULL>();
        }
      }
    };
    // what the NEXT phase's load part would issue (TT_Q8_TAIL: issued behind this phase's MFMAs instead)
    auto dma_issue_next = [&]() {
      if constexpr (HA == 0) {
        cursor_next_ktile();                                                                     // -> K-tile t + 2
        issue(std::integral_constant<int, 0>{}, std::integral_constant<int, (S + 2) % 3>{});
        issue(std::integral_constant<int, 1>{}, std::integral_constant<int, (S + 2) % 3>{});
      } else {
        issue(std::integral_constant<int, 2>{}, std::integral_constant<int, (S + 2) % 3>{});   // X1 of K-tile t + 2 (the cursor's)
      }
    };
    Q8_STAMP(ts0);
    if constexpr (TT_Q8_TAIL == 0) {
      if (!reads_first) dma_issue();
      __builtin_amdgcn_sched_barrier(0);
#ifdef TT_Q8_STAMP
      if (!reads_first) Q8_STAMP(ts1);
#endif
      frag_reads();
      __builtin_amdgcn_sched_barrier(0);
#ifdef TT_Q8_STAMP
      if (reads_first) Q8_STAMP_NOWAIT(ts1);
#endif
      if (reads_first) dma_issue();
    } else {
      frag_reads();
      __builtin_amdgcn_sched_barrier(0);
    }
    Q8_STAMP_NOWAIT(ts2);
    dma_wait();
    Q8_STAMP_NOWAIT(ts3);
    __builtin_amdgcn_s_barrier();
    Q8_STAMP_NOWAIT(ts4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    Q8_STAMP_NOWAIT(ts5);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DBG & 16) {
      // timing study (wrong numbers): every 32x32x16 MFMA replaced by TWO 16x16x32 MFMAs on the same operand registers - the same issue
      // cycles and flops - to see what the MFMA shape alone does to the clock the chip holds under this kernel ('DVFS give-back' item 7)
      if (work) {
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        auto two16 = [&](const f16x8& a, const f16x8& b, f32x16& c) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            f32x4_ c4 = {c[8 * q], c[8 * q + 1], c[8 * q + 2], c[8 * q + 3]};
            c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4, 0, 0, 0);
            c[8 * q] = c4[0]; c[8 * q + 1] = c4[1]; c[8 * q + 2] = c4[2]; c[8 * q + 3] = c4[3];
          }
        };
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            two16(One[ks], Two[mt][ks], a1[HA][mt]);
            two16(One[ks], Two[mt][2 + ks], a2[HA][mt]);
          }
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) two16(One[2 + ks], Two[mt][ks], a2[HA][mt]);
        }
      }
    } else
    if (work && !(DBG & 1)) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {   // the A operand is always the W fragment (swapped operands: an output row on a lane)
          if constexpr (SQ) {
            a1[HA][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Two[mt][ks], One[ks], a1[HA][mt], 0, 0, 0);
            a2[HA][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Two[mt][ks], One[2 + ks], a2[HA][mt], 0, 0, 0);
          } else {
            a1[HA][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(One[ks], Two[mt][ks], a1[HA][mt], 0, 0, 0);
            a2[HA][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(One[ks], Two[mt][2 + ks], a2[HA][mt], 0, 0, 0);
          }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          if constexpr (SQ) a2[HA][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Two[mt][2 + ks], One[ks], a2[HA][mt], 0, 0, 0);
          else a2[HA][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(One[2 + ks], Two[mt][ks], a2[HA][mt], 0, 0, 0);
        }
        if constexpr (TT_Q8_TAIL == 2) {
          if (ks == 0) {
            __builtin_amdgcn_sched_barrier(0);
            dma_issue_next();
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    if constexpr (TT_Q8_TAIL == 2) {
      if (!work || (DBG & 1)) dma_issue_next();
    }
    if constexpr (TT_Q8_TAIL == 1) {
      __builtin_amdgcn_sched_barrier(0);
      dma_issue_next();
    }
    if constexpr (DBG & 1) {
      if (work) {
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) keep += (float)One[i][0] + (float)Two[0][i][0] + (float)Two[1][i][0];
        a1[0][0][0] += keep;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    Q8_STAMP_NOWAIT(ts6);   // (all MFMAs issued)
    __builtin_amdgcn_s_barrier();
#ifdef TT_Q8_STAMP
    {
      unsigned long long te;
      Q8_STAMP(te);
      if (steady) {
        st_a[HA] += (unsigned)(ts1 - ts0); st_b[HA] += (unsigned)(ts2 - ts1); st_wait[HA] += (unsigned)(ts3 - ts2); st_bar1[HA] += (unsigned)(ts4 - ts3);
        st_lgkm[HA] += (unsigned)(ts5 - ts4); st_mfma[HA] += (unsigned)(ts6 - ts5); st_bar2[HA] += (unsigned)(te - ts6); ++st_n[HA];
      }
    }
#endif
  };

  // ---- epilogue of one item: per wave, through its private scratch, no workgroup barrier (gemm_planes8.hip).
  // Lane (m = r) holds columns 8 g + 4 h + {0..3} of a 32 x 32 MFMA tile in registers 4 g .. 4 g + 3.  A pass stages 16 columns
  // ([32][16] fp32, 16-byte chunks XOR-swizzled by the row) and reads them back row-major: fp32 outputs 4 columns per lane, pair
  // outputs 8 columns per lane (16 bytes of hi + 16 bytes of lo: a wave's 32 columns are exactly one pair group of the output row).
  unsigned char* scr = smem + RING_B + wave * SCR_B;
  constexpr int CPRW = CW / 4;                    // 16-byte chunks per staged row (4)
  constexpr int LPR = F32OUT ? CPRW : CPRW / 2;   // lanes per staged row on the way back
  constexpr int RPW = 64 / LPR;                   // rows per read-back instruction
  constexpr int NRB = 32 / RPW;                   // read-back instructions per pass
  constexpr int NLD = NPASS * NRB;                // ... per MFMA tile (fp32: 4)
  const int rr = lane / LPR, cc = lane % LPR;
  // swizzle of the staged rows (64 bytes = 4 chunks each; an LDS pass serves 16 lanes x 16 bytes = one 256-byte bank window = 4 rows):
  // rows r, r + 4, r + 8, r + 12 fall into the same quarter of the window and must differ in their chunk -> the row's bits 2..3
  // (bits 0..1 - round 3's choice - left the 16 lanes of a write pass 4-way conflicted: 14 % of the kernel's LDS cycles by PMC)
#ifndef TT_Q8_ESW
#define TT_Q8_ESW 1
#endif
  auto esw = [](int row) { return TT_Q8_ESW ? ((row >> 2) & (CPRW - 1)) : (row & (CPRW - 1)); };
  const unsigned out_bytes = (unsigned)g.M * (unsigned)g.N * 4u;   // fp32 [M][N] and pairs [M][2 N] fp16 alike
  auto epilogue = [&](int row0, int n0, bool half) {
    if constexpr (DBG & 8) {
      float sres = 0.f;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int e = 0; e < 16; ++e) { sres += a1[a][m][e] + a2[a][m][e]; a1[a][m][e] = 0.f; a2[a][m][e] = 0.f; }
      if (g.C && sres == 12345.678f) g.C[threadIdx.x] = sres;
      return;
    }
    constexpr int NT = 4;   // MFMA tiles in the order (ha, mt): a half item ends after the first two
    const int nt = half ? NT / 2 : NT;
    bool range_bad = false;
    const float inv_s = g.out_scale ? 1.0f / *g.out_scale : 1.0f;   // exact: S is a power of two
    // MFMA tile j = (ha, mt): output rows mrow(j) .., columns ncol(j) .. (+ 32 each)
    auto mrow = [&](int j) { return row0 + (j >> 1) * 128 + (SQ ? wr * 32 : wr * 64 + (j & 1) * 32); };
    auto ncol = [&](int j) { return n0 + (SQ ? wc * 64 + (j & 1) * 32 : wc * 32); };
    constexpr int NB = SQ ? 2 : 1;   // distinct column blocks of a wave
    f32x4 bias_lo[NB][NPASS], bias_hi[NB][NPASS];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int q = 0; q < NPASS; ++q) {
        const int n = ncol(b) + q * CW + (F32OUT ? 4 : 8) * cc;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        bias_lo[b][q] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + n) : zero;
        if constexpr (!F32OUT) bias_hi[b][q] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + n + 4) : zero;
      }
    f32x4 rres[3][RES ? NLD : 1];
    auto prefetch = [&](int j, int slot) {
      if constexpr (RES) {
        const int mbase = mrow(j), nbase = ncol(j);
#pragma unroll
        for (int q = 0; q < NPASS; ++q)
#pragma unroll
          for (int i = 0; i < NRB; ++i) {
            const unsigned off = ((unsigned)(mbase + i * RPW + rr) * (unsigned)g.N + (unsigned)(nbase + q * CW + 4 * cc)) * 4u;
            rres[slot][q * NRB + i] = q8_ld128(g.residual, out_bytes, off);
          }
      }
    };
    if constexpr (RES) {
      prefetch(0, 0);
      prefetch(1, 1);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j < nt) {
        const int ha = j >> 1, mt = j & 1;
        const int mbase = mrow(j), nbase = ncol(j);
        constexpr int BI = 0;
        const int bi = SQ ? mt : BI;
        if constexpr (RES) {
          if (j + 2 < nt) prefetch(j + 2, (j + 2) % 3);
        }
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
#pragma unroll
          for (int gg = 0; gg < CW / 8; ++gg) {
            const int gi = q * (CW / 8) + gg;
            const int phys = (2 * gg + h) ^ esw(r);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(a2[ha][mt][4 * gi + e], 0.00048828125f, a1[ha][mt][4 * gi + e]) * inv_s;   // exact 2^-11
            *reinterpret_cast<f32x4*>(scr + r * (CW * 4) + phys * 16) = v;
          }
#pragma unroll
          for (int i = 0; i < NRB; ++i) {
            const int row = i * RPW + rr;
            const int m = mbase + row;
            if constexpr (F32OUT) {
              f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * (CW * 4) + ((cc ^ esw(row)) << 4));
              v += bias_lo[bi][q];
              if constexpr (GG) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_fast_f(rres[j % 3][q * NRB + i][e]);
              } else if constexpr (RES) {
                v += rres[j % 3][q * NRB + i];
              }
              const unsigned off = ((unsigned)m * (unsigned)g.N + (unsigned)(nbase + q * CW + 4 * cc)) * 4u;
              q8_st128(g.C, out_bytes, off, __builtin_bit_cast(u32x4, v));
            } else {
              f32x4 v0 = *reinterpret_cast<const f32x4*>(scr + row * (CW * 4) + (((2 * cc) ^ esw(row)) << 4));
              f32x4 v1 = *reinterpret_cast<const f32x4*>(scr + row * (CW * 4) + (((2 * cc + 1) ^ esw(row)) << 4));
              v0 += bias_lo[bi][q];
              v1 += bias_hi[bi][q];
              float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
              if constexpr (BOTH) {   // the fp32 value (before the activation): 8 columns = two 16-byte stores
                const unsigned offc = ((unsigned)m * (unsigned)g.N + (unsigned)(nbase + q * CW + 8 * cc)) * 4u;
                q8_st128(g.C, out_bytes, offc, __builtin_bit_cast(u32x4, v0));
                q8_st128(g.C, out_bytes, offc + 16u, __builtin_bit_cast(u32x4, v1));
              }
              if constexpr (ACT) {
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
#ifdef TT_Q8_GELU_SCALAR   // (A/B: the round-4 form)
                  v[e] = gelu_fast_f(v[e]);
                  v[e + 1] = gelu_fast_f(v[e + 1]);
#else
                  const tt_f32x2 gq = gelu_fast_f2(tt_f32x2{v[e], v[e + 1]});
                  v[e] = gq[0];
                  v[e + 1] = gq[1];
#endif
                }
              }
              f16x8 qh, ql;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                _Float16 hi_, lo_;
                split_pair(v[e], hi_, lo_);
                qh[e] = hi_;
                ql[e] = lo_;
                if (m < g.M) range_bad |= pair_hi_bad(hi_);   // (rows beyond M are dropped by the store's range check)
              }
              // pair group of this wave's 32 columns: byte offset of (m, nbase) = (m * 2 N + 2 nbase) * 2; hi then lo (64 bytes on)
              const unsigned off = ((unsigned)m * (unsigned)g.N + (unsigned)nbase) * 4u + (unsigned)(q * CW + 8 * cc) * 2u;
              q8_st128(g.Cp, out_bytes, off, __builtin_bit_cast(u32x4, qh));
              q8_st128(g.Cp, out_bytes, off + 64u, __builtin_bit_cast(u32x4, ql));
            }
          }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) { a1[ha][mt][e] = 0.f; a2[ha][mt][e] = 0.f; }
      }
    }
    // The stores are NOT waited for: they share the vmcnt queue with the LDS-DMA in issue order, and the waits of the next POST
    // phases - whose windows still reach back across this epilogue - allow for S_FULL / S_HALF more outstanding operations, a LOWER
    // bound of what was really issued (allowing fewer only waits for a few of the oldest epilogue operations as well).
    post_epi = POST;
    post_half = half;
    if constexpr (!F32OUT) range_flag_raise(g.range_flag, range_bad);
  };

  // ---- K-split item: this wave's partial -> workspace; the wave arriving last at the (tile, wave) counter sums the slices' partials in
  // slice order (its own from registers, in its place: the result does not depend on who finishes) into a1 and goes on to the epilogue.
  // The slices of a tile may run on different XCDs (separate L2s): the partials are stored write-through and loaded around the L2
  // (sc0 sc1), the stores are complete (vmcnt 0) before the wave's agent-scope increment of the counter.
  auto slice_reduce = [&]() -> bool {
    const int j = cu / g.ks_S, sl = cu - j * g.ks_S;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) { a1[a][m][e] = fmaf(a2[a][m][e], 0.00048828125f, a1[a][m][e]); a2[a][m][e] = 0.f; }
    // this (tile, wave)'s partials: slice t at byte t * 128 KB, 64 values per lane as 16 lane-contiguous 16-byte pieces
    float* wbase = g.ks_ws + ((size_t)j * g.ks_S * 8 + wave) * 4096;
    const unsigned wbytes = (unsigned)g.ks_S * 8u * 4096u * 4u;
    const unsigned sstride = 8u * 4096u * 4u;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {a1[a][m][4 * q], a1[a][m][4 * q + 1], a1[a][m][4 * q + 2], a1[a][m][4 * q + 3]};
          q8_st128_sys(wbase, wbytes, (unsigned)sl * sstride + (unsigned)(((a * 2 + m) * 4 + q) * 1024 + lane * 16), __builtin_bit_cast(u32x4, v));
        }
    q8_wait_vmcnt<0>();
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(g.ks_cnt + j * 8 + wave, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != g.ks_S - 1) return false;
    if (lane == 0) __hip_atomic_store(g.ks_cnt + j * 8 + wave, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // every slice's partial (its own too: the sum is in slice order whoever finishes) back from memory, 16 loads in flight per slice;
    // a2 is the landing buffer, a1 the running sum.  (Two slices in flight - a second 64-register landing buffer - spills.)
    for (int t = 0; t < g.ks_S; ++t) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = q8_ld128_sys(wbase, wbytes, (unsigned)t * sstride + (unsigned)(((a * 2 + m) * 4 + q) * 1024 + lane * 16));
#pragma unroll
            for (int e = 0; e < 4; ++e) a2[a][m][4 * q + e] = v[e];
          }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int e = 0; e < 16; ++e) a1[a][m][e] = t == 0 ? a2[a][m][e] : a1[a][m][e] + a2[a][m][e];
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) a2[a][m][e] = 0.f;
    return true;
  };

  // ---- prologue: K-tile 0 whole, W and X0 of K-tile 1 (what phases -3 .. -1 of the steady schedule would have issued, in its order)
  {
    using I0_ = std::integral_constant<int, 0>; using I1_ = std::integral_constant<int, 1>; using I2_ = std::integral_constant<int, 2>;
    issue(I0_{}, I0_{}); issue(I1_{}, I0_{}); issue(I2_{}, I0_{});
    cursor_next_ktile();
    issue(I0_{}, I1_{}); issue(I1_{}, I1_{});
    half_guard = d_half ? 4 : 0;
    if (DBG & 2) q8_wait_vmcnt<0>();
    else if (d_half || d_done) q8_wait_vmcnt<WGUARD>();   // (d_done: a single K-tile cannot happen, nk >= 3)
    else q8_wait_vmcnt<WFULL>();
    if constexpr (TT_Q8_TAIL != 0) issue(I2_{}, I1_{});   // ... and X1 of K-tile 1, which the tail of "phase (-1, 1)" would have issued
    __builtin_amdgcn_s_barrier();
  }

  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  for (int it = 0; it < n_items; ++it) {
    int row0, n0, c_kt0, c_kend;
    item(it, row0, n0, c_half, c_kt0, c_kend);
    if (grp1) __builtin_amdgcn_s_barrier();   // the second group runs one barrier interval behind
    for (int kk = c_kt0; kk < c_kend; kk += 3) {
      // steady for these 6 phases?  The cursor advances three K-tiles in them: it must stay in whole tiles and short of the end.
      steady = post_epi == 0 && half_guard == 0 && !c_half && !d_half && !d_done;
      if (steady && d_kt + 3 >= d_kend) {   // it crosses into the next item
        bool nhalf = false;
        if (d_item + 1 < n_items) { int r0_, n0_, k0_, k1_; item(d_item + 1, r0_, n0_, nhalf, k0_, k1_); }
        steady = d_item + 1 < n_items && !nhalf;
      }
      phase(I0{}, I0{}); phase(I0{}, I1{});
      phase(I1{}, I0{}); phase(I1{}, I1{});
      phase(I2{}, I0{}); phase(I2{}, I1{});
    }
    steady = false;
    if (!grp1) __builtin_amdgcn_s_barrier();  // realign: both groups run their epilogues at the same time
    if (has_slice && it == n_whole) {
      if (!slice_reduce()) continue;   // (the last item: nothing follows)
    }
    epilogue(row0, n0, c_half);
  }
#ifdef TT_Q8_CLOCK
  if (g.order_mode >= 100 && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 101 || blockIdx.x == 202)) {
    const unsigned long long dt = __builtin_amdgcn_s_memtime() - clk0_t, dr = __builtin_amdgcn_s_memrealtime() - clk0_r;
    printf("q8 clock: block %d  %llu cycles in %llu x 10 ns = %.3f GHz\n", (int)blockIdx.x, dt, dr, (double)dt / (double)dr * 0.1);
  }
#endif
#ifdef TT_Q8_STAMP
  if (g.order_mode >= 100 && lane == 0 && (wave == 0 || wave == 2 || wave == 5 || wave == 7) && blockIdx.x == 3) {
    const unsigned long long dt = __builtin_amdgcn_s_memtime() - clk_t0, dr = __builtin_amdgcn_s_memrealtime() - clk_r0;
    for (int ha = 0; ha < 2; ++ha)
      if (st_n[ha] > 0)
        printf("q8 stamps: wave %d phase ha=%d  %u steady phases, cycles: first part (DMA issue, or reads when reads-first) %.0f | second part %.0f | counted wait %.0f | "
               "barrier 1 %.0f | lgkmcnt(0) %.0f | MFMAs issued %.0f | barrier 2 %.0f   [kernel %llu cycles, %.3f GHz]\n", wave, ha, st_n[ha],
               (double)st_a[ha] / st_n[ha], (double)st_b[ha] / st_n[ha], (double)st_wait[ha] / st_n[ha], (double)st_bar1[ha] / st_n[ha],
               (double)st_lgkm[ha] / st_n[ha], (double)st_mfma[ha] / st_n[ha], (double)st_bar2[ha] / st_n[ha], dt, (double)dt / (double)dr * 0.1);
  }
#endif
}


// TT_Q8S_ST_AUX: cache policy of the epilogue's output stores (timing study).  The kernel is bound by what the memory side delivers per CU
// (round 4's ablation: with no MFMA at all it still takes 88 % of its time; PMC: L2 hit 0.63 - 0.78, 3.5 x the operand bytes fetched from
// beyond L2), and an output tile is 128 KB that nobody reads again in this launch - 4 MB per round and XCD through a 4 MB L2 that has to hold
// the X row blocks the sibling column tiles re-read.  Measured (tools/ab_pairs.py, one box, us: qkv / proj / fc1 / fc2): plain stores (0)
// 73.0 / 36.6 / 111.5 / 95.4; sc1 (16: written through and dropped from L2) 74.2 / 35.6 / 120.8 / 96.4; nt (2) 73.0 / 37.5 / 186.1 / 95.6 -
// the 16-byte hi / lo pieces of a pair output are partial lines, which only a write-back L2 merges.  Plain stays.
#ifndef TT_Q8S_ST_AUX
#define TT_Q8S_ST_AUX 0
#endif
// =====================================================================================================================================
// gemm_pairs8s_kernel (round 5): the same tile, operands, work items and epilogues on a SYMMETRIC, register-prefetched main loop.
//
// What round 4's profile said of the kernel above (profiles/r04_gemm_pairs_pmc.json, r04_q8_ablation.txt): matrix pipe busy 0.45 of a
// workgroup's life on the K = 384 shapes; its two wave groups alternate a "load part" (LDS-DMA issue, then 12 / 8 fragment reads, then a
// counted wait) with an "MFMA part" of 384 cycles, four barrier intervals per K-tile - and the load part is the LONG one (457 / 527
// cycles: the four waves of a group hit the texture path with 8 - 16 KB at once, 64 B / clk, then the LDS with 48 KB of reads at once);
// fragment reads + barriers alone, no MFMAs, no DMA, take 68 k cycles against the MFMAs' 64.5 k.  Here:
//   * every wave runs the SAME stream, one barrier per K-tile: [wait for the ring | barrier | 24 MFMAs with the 16 fragment reads of the
//     NEXT phases and the 6 LDS-DMA pieces of later K-tiles issued between them].  The two waves of a SIMD interleave their MFMAs on the
//     shared matrix pipe (each sees a 64-cycle cadence): a read (16 cycles of the SIMD's LDS return path) or a DMA piece fits in the gap,
//     and what one wave waits for at the texture path the other covers with MFMAs - the DMA sites of waves 4 - 7 are shifted by two MFMAs
//     against those of waves 0 - 3 (their SIMD partners), so that at most four 1 KB pieces arrive at the texture path together;
//   * fragments are prefetched into registers and refreshed IN PLACE, each as soon as the last MFMA of the K-tile that reads it has been
//     issued: W (both k-steps, hi and lo, of the wave's two 32-row blocks: 32 registers) and X half 0 (16) during phase B (x half 1) of the
//     K-tile before, X half 1 (16) during phase A.  128 accumulator + 64 fragment registers (a second W set - 96 - was built first:
//     the allocator spilled accumulators around every item boundary);
//   * the ring is the same three K-tiles of 48 KB.  Registers free LDS one K-tile earlier, so the DMA runs one K-tile further ahead:
//     iteration t issues X1 of K-tile t + 2 and W, X0 of K-tile t + 3 (into the slot of K-tile t, whose W / X0 went to registers in
//     iteration t - 1; its X1 chunk, read in phase A of t, is refilled in iteration t + 1).
//     RAW: phase A of t reads X1(t), both phases read W / X0(t + 1): issued in iteration t - 2, retired by `vmcnt(6)` (the six pieces of
//     iteration t - 1 stay in flight) + the barrier at the top of t.  WAR: a chunk is overwritten by a DMA issued after the barrier that
//     follows the `lgkmcnt(0)` of its last readers: W / X0 of slot(t) were read in iteration t - 1 (complete before barrier t), X1 of
//     slot(t + 2) = slot(t - 1) in phase A of t - 1 (complete before barrier t).
//   * the ring slot is a run-time offset and nothing alternates at compile time: any K % 32 == 0 (the kernel above: K % 96 == 0) - the
//     projection head (K = 1024 / 512 / 256) is taken.
// Items, half tiles, the K-split exchange and the epilogue are those of the kernel above.
// TT_Q8S_LOADER: who issues the LDS-DMA.  0: every wave its own six pieces per K-tile.  1: waves 4 - 7 issue all twelve of their SIMD (their own
// and those of wave - 4), waves 0 - 3 none.  2: waves 0 - 3 do.  Why (stamps, tools/q8s_stamp.py, qkv shape): the two waves of a SIMD do NOT share the matrix pipe
// evenly - the older wave (0 - 3) wins the arbitration (priority, then age: MI355X_MICROARCH "Two waves per SIMD" item 2), is through its 24
// MFMAs after ~ 1050 cycles and parks ~ 900 cycles at the next barrier, while its partner needs ~ 1790: once alone on the SIMD, every one of
// its DMA-issue stalls and read waits idles the pipe.  With the loader role on the YOUNGER wave its stalls fall into the time in which it
// would be waiting for pipe slots anyway, and its lone second half is MFMAs and reads only (that was the idea; the measurement says the
// opposite assignment, 2, is the one that pays - see below).
// Measured (tools/ab_pairs.py, one box, us: qkv / proj / fc1 / fc2 / ViT-B qkv): 0: 74.6 / 37.0 / 111.7 / 95.3 / 246.1; 1: 75.2 / 37.5 / 111.5 /
// 95.5 / 249.5; 2: 72.5 / 36.5 / 108.8 / 94.3 / 238.5 - the OLDER wave runs first whatever it does (it wins the matrix pipe), so ITS stalls are
// the ones the younger wave's MFMAs fill; the younger wave's lone second half should be free of them.  (s_setprio for either half: +- 1 %.)
#ifndef TT_Q8S_LOADER
#define TT_Q8S_LOADER 2
#endif
// DBG (timing studies only, tools/q8_ablate.py; the shipped instantiations are DBG = 0), a bit mask: 1 no MFMAs, 2 no LDS-DMA, 8 no epilogue,
// 4 hot operands (every item streams operand tile (0, 0): what the traffic from beyond the L2 costs)
template <int EPI, int DBG = 0>
__global__ __launch_bounds__(512) void gemm_pairs8s_kernel(Q8Args g) {
  constexpr int ROWB = 128, CPR = 8, WIN = 2, RPI = 8;
  constexpr int CHUNK_B = 128 * ROWB;       // 16 KB
  constexpr int SLOT_B = 3 * CHUNK_B;       // W, X0, X1
  constexpr int RING_B = 3 * SLOT_B;
  constexpr int SCR_B = (160 * 1024 - RING_B) / 8;
  constexpr int CW = SCR_B / 128;
  constexpr int NPASS = 32 / CW;
  constexpr int BN = 128;
  constexpr bool F32OUT = EPI == Q8_F32 || EPI == Q8_F32_RES || EPI == Q8_F32_GELUGRAD;
  constexpr bool RES = EPI == Q8_F32_RES || EPI == Q8_F32_GELUGRAD;
  constexpr bool GG = EPI == Q8_F32_GELUGRAD;
  constexpr bool BOTH = EPI == Q8_BOTH || EPI == Q8_BOTH_GELU;
  constexpr bool ACT = EPI == Q8_PAIR_GELU || EPI == Q8_BOTH_GELU;
  constexpr int ST_TILE = BOTH ? 8 : 4;               // stores a wave issues per 32 x 32 MFMA tile
  constexpr int ST_FULL = 4 * ST_TILE, ST_HALF = ST_FULL / 2;
  constexpr int NPW = TT_Q8S_LOADER ? 12 : 6;
  constexpr int S_FULL = ST_FULL < 63 - NPW ? ST_FULL : 63 - NPW, S_HALF = ST_HALF < 63 - NPW ? ST_HALF : 63 - NPW;
  static_assert(RING_B + 8 * SCR_B <= 160 * 1024 && CW == 16, "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char smem[160 * 1024];

#ifdef TT_Q8S_STAMP   // diagnostic build only (tools/q8s_stamp.py): where a wave's K-tile iteration goes - s_memtime stamps around the top waits
  unsigned long long sa = 0, sb = 0, sc_ = 0, sd = 0, se = 0;
  unsigned st_lgkm = 0, st_vm = 0, st_bar = 0, st_body = 0, st_n = 0, st_epi = 0, st_nepi = 0, st_pro = 0;
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#define Q8S_STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
#else
#define Q8S_STAMP(t)
#endif
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool late = wave >= 4;                      // the SIMD partners of waves 0 - 3: their DMA sites are two MFMAs later
  const int wr = wave & 3, wc = wave >> 2;          // 4 x 2 waves of 64 x 64
  const int r = lane & 31, h = lane >> 5;
  const int K4 = g.K * 4, nk = g.K / 32;

  // ---- work items (as gemm_pairs8_kernel)
  int cu = blockIdx.x;
  if ((g.ncu & 7) == 0) cu = (blockIdx.x & 7) * (g.ncu >> 3) + (blockIdx.x >> 3);
  int n_whole;
  bool has_half = false;
  const bool ksplit = g.ks_S >= 2;
  if (g.n_full > 0 || g.n_half > 0 || ksplit) {
    n_whole = g.n_full;
    has_half = !ksplit && cu < g.n_half;
  } else {
    n_whole = cu < g.ntiles ? (g.ntiles - cu + g.ncu - 1) / g.ncu : 0;
  }
  const bool has_slice = ksplit && cu < g.ks_R * g.ks_S;
  const int n_items = n_whole + (has_half || has_slice ? 1 : 0);
  if (n_items == 0) return;
  const bool half_first = has_half && (cu & 1) && n_whole > 0;
  auto item = [&](int it, int& row0, int& n0, bool& half, int& kt0, int& kend) {
    int tile;
    kt0 = 0;
    kend = nk;
    half = has_half && (half_first ? it == 0 : it == n_whole);
    int hsel = 0;
    if (half) {
      tile = g.n_full * g.ncu + (cu >> 1);
      hsel = cu & 1;
    } else if (has_slice && it == n_whole) {
      const int j = cu / g.ks_S, sl = cu - j * g.ks_S;   // slices: whole K-tiles, at least one each (pairs8_plan, kgroup = 1: ks_S <= nk)
      tile = g.n_full * g.ncu + j;
      kt0 = sl * nk / g.ks_S;
      kend = (sl + 1) * nk / g.ks_S;
    } else {
      tile = (half_first ? it - 1 : it) * g.ncu + cu;
    }
    const int mb = tile / g.ntn, ns = tile - mb * g.ntn;
    row0 = mb * 256 + hsel * 128;
    n0 = ns * BN;
  };

  // ---- LDS-DMA lane map (as above): lane -> (row, 16-byte slot) of a 1 KB piece, source chunk = slot ^ f(row).  ONE per-lane offset serves
  // every piece (W and X alike: rows of K4 bytes): the piece's first row and the K-tile are a SCALAR added to it, and the descriptors
  // carry the operands' true sizes, so that rows beyond M (a ragged last tile) and the pieces of K-tiles that do not exist (the cursor
  // past its last item: scalar 2^31) are zero-filled by the range check instead of being clamped or branched around - every iteration
  // issues exactly six pieces, and every counted wait is `vmcnt(6)`.
  const int l_row = lane / CPR, l_slot = lane % CPR;
  const int d_row0 = wave * RPI + l_row;
  const int d_chunk = l_slot ^ ((d_row0 / WIN) & (CPR - 1));
  const unsigned lane_voff = (unsigned)(d_row0 * K4 + d_chunk * 16);
  // (TT_Q8S_LOADER: a loader wave also issues the pieces of its SIMD partner, wave ^ 4: rows 32 further up / down the chunk)
  const int d_row0b = (wave ^ 4) * RPI + l_row;
  const unsigned lane_voff_b = (unsigned)(d_row0b * K4 + (l_slot ^ ((d_row0b / WIN) & (CPR - 1))) * 16);
  constexpr int NP = TT_Q8S_LOADER ? 12 : 6;   // pieces a DMA-issuing wave has in flight per K-tile: the unit of its counted waits
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(g.X), 0, (unsigned)g.M * (unsigned)K4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(g.W), 0, (unsigned)g.N * (unsigned)K4, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const unsigned piece_step = 8u * RPI * (unsigned)K4;   // 64 rows: the second piece of a chunk

  // ---- DMA cursor: the K-tile whose W / X0 chunks are issued next (three K-tiles ahead of the MFMAs), as scalar byte offsets of its first
  // W row / first X row + its K-tile; `lag` = the K-tile before it, whose X1 chunk is issued next
  int d_item = 0, d_kt = 0, d_kend = 0;
  bool d_done = false, d_half = false;
  unsigned cur_w = OOB, cur_x = OOB, lag_x1 = OOB;
  auto cursor_item = [&]() {
    int row0, n0;
    item(d_item, row0, n0, d_half, d_kt, d_kend);
    if constexpr (DBG & 4) { row0 = 0; n0 = 0; }   // (timing study: every item streams the SAME operand tiles - hot in every XCD's L2)
    cur_w = (unsigned)n0 * (unsigned)K4 + (unsigned)d_kt * ROWB;
    cur_x = (unsigned)row0 * (unsigned)K4 + (unsigned)d_kt * ROWB;
  };
  cursor_item();
#ifdef TT_Q8S_PRIO   // timing study: 1 = the younger waves (4 - 7) at priority 1 for the whole kernel (MI355X_MICROARCH "Two waves per SIMD" item 4)
  if (TT_Q8S_PRIO == 1 && late) __builtin_amdgcn_s_setprio(1);
  if (TT_Q8S_PRIO == 2 && !late) __builtin_amdgcn_s_setprio(1);
#endif
  // ring slots (byte offsets) of K-tile t (s0), t + 1 (s1), t + 2 (s2); W / X0 of t + 3 go to s0
  int s0 = 0, s1 = SLOT_B, s2 = 2 * SLOT_B;
  // which: 0 this wave's piece, 1 (TT_Q8S_LOADER, loader waves only) the piece of its partner wave ^ 4, 2 both
  auto dma = [&](const __amdgpu_buffer_rsrc_t& rs, int lds_off, unsigned soff, int which) {
    if constexpr (!(DBG & 2)) {
      if (which != 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (q8_lds_ptr_t)(smem + lds_off + wave * 1024), 16, lane_voff + soff, 0, 0, 0);
      if (which != 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (q8_lds_ptr_t)(smem + lds_off + (wave ^ 4) * 1024), 16, lane_voff_b + soff, 0, 0, 0);
    }
  };
  // the six DMA sites of an iteration: X1 0 / 1 of the lag (-> slot_x1), W 0 / 1 and X0 0 / 1 of the cursor (-> slot_wx)
  auto site = [&](int k, int slot_x1, int slot_wx, int which = 0) {
    if (k == 0) dma(rs_x, slot_x1 + 2 * CHUNK_B, lag_x1, which);
    else if (k == 1) dma(rs_x, slot_x1 + 2 * CHUNK_B + 8 * 1024, lag_x1 + piece_step, which);
    else if (k == 2) dma(rs_w, slot_wx, cur_w, which);
    else if (k == 3) dma(rs_w, slot_wx + 8 * 1024, cur_w + piece_step, which);
    else if (k == 4) dma(rs_x, slot_wx + CHUNK_B, cur_x, which);
    else if (k == 5) dma(rs_x, slot_wx + CHUNK_B + 8 * 1024, cur_x + piece_step, which);
  };
  // after the cursor's W / X0 have been issued: it becomes the lag, and moves on
  auto cursor_advance = [&]() {
    lag_x1 = (d_done || d_half) ? OOB : cur_x + 128u * (unsigned)K4;
    if (d_done) return;
    ++d_kt;
    cur_w += ROWB;
    cur_x += ROWB;
    if (d_kt == d_kend) {
      ++d_item;
      if (d_item >= n_items) { d_done = true; cur_w = OOB; cur_x = OOB; }
      else cursor_item();
    }
  };

  // ---- fragment addressing (as above)
  const int f_sw = (r / WIN) & (CPR - 1);
  int fo[4];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) fo[c4] = r * ROWB + (((2 * c4 + h) ^ f_sw) << 4);
  const int x_slice = wr * 32 * ROWB, w_slice = wc * 64 * ROWB;

  f32x16 a1[2][2], a2[2][2];   // zeroed at the top of every item (not at the end of the epilogue: 128 registers of zeros live across the
                               // item loop's header were spilled)
  f16x8 Wf[2][4];      // [w block][hi k-step 0, 1, lo k-step 0, 1]
  f16x8 Xf[2][4];      // [x half][...]
  auto rd_w = [&](int slot, int mt, int c4) { Wf[mt][c4] = *reinterpret_cast<const f16x8*>(smem + slot + w_slice + mt * 32 * ROWB + fo[c4]); };
  auto rd_x = [&](int ha, int slot, int c4) { Xf[ha][c4] = *reinterpret_cast<const f16x8*>(smem + slot + (1 + ha) * CHUNK_B + x_slice + fo[c4]); };

  // counted wait at the top of an iteration: all but the six pieces of the previous iteration (+ the stores of an epilogue that lies
  // between: a lower bound of their number) are complete
  int post_epi = 0;         // iterations whose window still reaches back across an epilogue
  bool post_half = false;
  auto wait_ring = [&]() {
    if constexpr (DBG & 2) { q8_wait_vmcnt<0>(); return; }
    if (TT_Q8S_LOADER != 0 && late != (TT_Q8S_LOADER == 1)) {   // not a loader: no DMA of its own, nothing to wait for
      if (post_epi > 0) --post_epi;
      return;
    }
    if (post_epi > 0) {
      --post_epi;
      if (post_half) q8_wait_vmcnt<NP + S_HALF>();
      else q8_wait_vmcnt<NP + S_FULL>();
    } else {
      q8_wait_vmcnt<NP>();
    }
  };

  // ---- one K-tile.  LAST: the item's last K-tile (no refresh of the fragments: they would be live across the epilogue); HALF: a half item
  // (x half 0 only); LATE: waves 4 - 7.  A K-tile is 8 (HALF: 4) "slots" of three MFMAs - one (k-step, w block) pair: hi lo | hi hi | lo hi,
  // the two that share the cross accumulator apart - with this wave's reads and DMA sites between them: waves 0 - 3 issue a slot's piece
  // behind its first MFMA, their SIMD partners behind its last one.  Fragments are refreshed IN PLACE, each as soon as the last MFMA of this
  // K-tile that reads it has been issued (a register is read when its MFMA issues; the LDS data arrives > 64 cycles later):
  //   phase A (x half 0): X1(t) -> Xf[1], one fragment per slot (phase B of t - 1 was its last reader);
  //   phase B (x half 1): X0(t + 1) -> Xf[0], one per slot (phase A is over); W(t + 1): the slot's own two fragments behind its last MFMA.
  // A fragment is next read at least four slots (~ 800 cycles) after its refresh was issued.
#define Q8S_FENCE() __builtin_amdgcn_sched_barrier(0)
  auto ktile = [&](auto last_c, auto half_c, auto late_c) {
    constexpr bool LAST = decltype(last_c)::value, HALF = decltype(half_c)::value, LATE = decltype(late_c)::value;
    // top: the reads of the previous iteration are in registers (and nobody still reads what this iteration's DMA overwrites); the ring
    // holds X1(t) and W / X0(t + 1) once every wave has passed its counted wait
#ifdef TT_Q8S_STAMP
    Q8S_STAMP(sa);                       // (its lgkmcnt(0) IS the top's wait for the fragment reads)
    if (sd) st_body += (unsigned)(sa - sd);
    Q8S_STAMP(sb);
    wait_ring();
    Q8S_STAMP(sc_);
    __builtin_amdgcn_s_barrier();
    Q8S_STAMP(sd);
    st_lgkm += (unsigned)(sb - sa); st_vm += (unsigned)(sc_ - sb); st_bar += (unsigned)(sd - sc_); ++st_n;
#else
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_ring();
    __builtin_amdgcn_s_barrier();
#endif
    Q8S_FENCE();
    const int sx1 = s2, swx = s0, sr0 = s0, sr1 = s1;
    // slot (phase PH, q = (k-step, w block)): k, k2 = its DMA site(s), -1 = none
    auto slot = [&](auto ph_c, int q, int k, int k2) {
      constexpr int PH = decltype(ph_c)::value;
      const int ks = q >> 1, mt = q & 1;
      if constexpr (!(DBG & 1)) a2[PH][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[mt][ks], Xf[PH][2 + ks], a2[PH][mt], 0, 0, 0);
      else asm volatile("" ::"v"(Wf[mt][ks]), "v"(Xf[PH][2 + ks]));
      Q8S_FENCE();
      if constexpr (!LATE && TT_Q8S_LOADER == 0) { if (k >= 0) site(k, sx1, swx); if (k2 >= 0) site(k2, sx1, swx); }
      if constexpr (LATE && TT_Q8S_LOADER != 0) { if (k >= 0) site(k, sx1, swx, 0); if (k2 >= 0) site(k2, sx1, swx, 0); }   // (LATE = "is a loader" then) its own piece(s) here, its partner's behind the slot's last MFMA
      Q8S_FENCE();
      if constexpr (!(DBG & 1)) a1[PH][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[mt][ks], Xf[PH][ks], a1[PH][mt], 0, 0, 0);
      else asm volatile("" ::"v"(Xf[PH][ks]));
      Q8S_FENCE();
      if constexpr (!HALF) {
        if constexpr (PH == 0) rd_x(1, sr0, q);                    // X1(t): phase B's operand
        else if constexpr (!LAST) rd_x(0, sr1, q);                 // X0(t + 1)
      }
      Q8S_FENCE();
      if constexpr (!(DBG & 1)) a2[PH][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[mt][2 + ks], Xf[PH][ks], a2[PH][mt], 0, 0, 0);
      else asm volatile("" ::"v"(Wf[mt][2 + ks]));
      Q8S_FENCE();
      if constexpr (!LAST && (HALF || PH == 1)) {                  // the last phase of the K-tile: W(t + 1) in place, this slot's two fragments
        rd_w(sr1, mt, ks);
        rd_w(sr1, mt, 2 + ks);
        if constexpr (HALF) {                                      // ... and X0(t + 1): k-step ks is done once its second w block (mt = 1) is
          if (mt == 1) { rd_x(0, sr1, ks); rd_x(0, sr1, 2 + ks); }
        }
      }
      Q8S_FENCE();
      if constexpr (LATE && TT_Q8S_LOADER == 0) { if (k >= 0) site(k, sx1, swx); if (k2 >= 0) site(k2, sx1, swx); }
      if constexpr (LATE && TT_Q8S_LOADER != 0) { if (k >= 0) site(k, sx1, swx, 1); if (k2 >= 0) site(k2, sx1, swx, 1); }
      Q8S_FENCE();
    };
    using PH0 = std::integral_constant<int, 0>; using PH1 = std::integral_constant<int, 1>;
    if constexpr (HALF) {   // (its X1 pieces are the lag's: out of range for a half item, zero-filled into a chunk nobody reads)
      slot(PH0{}, 0, 0, 2); slot(PH0{}, 1, 3, -1); slot(PH0{}, 2, 1, 4); slot(PH0{}, 3, 5, -1);
    } else {
      slot(PH0{}, 0, 0, -1); slot(PH0{}, 1, 1, -1); slot(PH0{}, 2, 2, -1); slot(PH0{}, 3, -1, -1);
      slot(PH1{}, 0, 3, -1); slot(PH1{}, 1, 4, -1); slot(PH1{}, 2, 5, -1); slot(PH1{}, 3, -1, -1);
    }
    cursor_advance();
    const int t_ = s0; s0 = s1; s1 = s2; s2 = t_;
  };

  // ---- epilogue of one item (gemm_pairs8_kernel's, through the wave's private scratch)
  unsigned char* scr = smem + RING_B + wave * SCR_B;
  constexpr int CPRW = CW / 4;
  constexpr int LPR = F32OUT ? CPRW : CPRW / 2;
  constexpr int RPW = 64 / LPR;
  constexpr int NRB = 32 / RPW;
  constexpr int NLD = NPASS * NRB;
  auto esw = [](int row) { return (row >> 2) & (CPRW - 1); };
  const unsigned out_bytes = (unsigned)g.M * (unsigned)g.N * 4u;
  float amax_run = 0.f;   // (Q8_F32_GELUGRAD with g.amax_out: max |.| of what this wave stored - a dy for the next Linear's backward)
  auto epilogue = [&](int row0, int n0, bool half) {
    if constexpr (DBG & 8) {
      float sres = 0.f;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int e = 0; e < 16; ++e) sres += a1[a][m][e] + a2[a][m][e];
      if (g.C && sres == 12345.678f) g.C[threadIdx.x] = sres;
      return;
    }
    constexpr int NT = 4;
    const int nt = half ? NT / 2 : NT;
    // every lane-dependent address of the epilogue is formed HERE, from a lane id the compiler cannot see through: hoisted out of the item
    // loop they would be live across the main loop, whose 224 accumulator + fragment registers leave no room for them (they spilled)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int r = lane_e & 31, h = lane_e >> 5;
    const int rr = lane_e / LPR, cc = lane_e % LPR;
    // range flag (common.hpp pair_hi_bad) at one VALU op per element: the running maximum of the hi halves' magnitude bits, packed
    unsigned hi_max = 0;
    const float inv_s = g.out_scale ? 1.0f / *g.out_scale : 1.0f;
    auto mrow = [&](int j) { return row0 + (j >> 1) * 128 + wr * 32; };
    auto ncol = [&](int j) { return n0 + wc * 64 + (j & 1) * 32; };
    f32x4 bias_lo[2][NPASS], bias_hi[2][NPASS];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < NPASS; ++q) {
        const int n = ncol(b) + q * CW + (F32OUT ? 4 : 8) * cc;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        bias_lo[b][q] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + n) : zero;
        if constexpr (!F32OUT) bias_hi[b][q] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + n + 4) : zero;
      }
    f32x4 rres[3][RES ? NLD : 1];
    auto prefetch = [&](int j, int slot) {
      if constexpr (RES) {
        const int mbase = mrow(j), nbase = ncol(j);
#pragma unroll
        for (int q = 0; q < NPASS; ++q)
#pragma unroll
          for (int i = 0; i < NRB; ++i) {
            const unsigned off = ((unsigned)(mbase + i * RPW + rr) * (unsigned)g.N + (unsigned)(nbase + q * CW + 4 * cc)) * 4u;
            rres[slot][q * NRB + i] = q8_ld128(g.residual, out_bytes, off);
          }
      }
    };
    if constexpr (RES) {
      prefetch(0, 0);
      prefetch(1, 1);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j < nt) {
        const int ha = j >> 1, mt = j & 1;
        const int mbase = mrow(j), nbase = ncol(j);
        if constexpr (RES) {
          if (j + 2 < nt) prefetch(j + 2, (j + 2) % 3);
        }
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
#pragma unroll
          for (int gg = 0; gg < CW / 8; ++gg) {
            const int gi = q * (CW / 8) + gg;
            const int phys = (2 * gg + h) ^ esw(r);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(a2[ha][mt][4 * gi + e], 0.00048828125f, a1[ha][mt][4 * gi + e]) * inv_s;
            *reinterpret_cast<f32x4*>(scr + r * (CW * 4) + phys * 16) = v;
          }
#pragma unroll
          for (int i = 0; i < NRB; ++i) {
            const int row = i * RPW + rr;
            const int m = mbase + row;
            if constexpr (F32OUT) {
              f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * (CW * 4) + ((cc ^ esw(row)) << 4));
              v += bias_lo[mt][q];
              if constexpr (GG) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_fast_f(rres[j % 3][q * NRB + i][e]);
                if (m < g.M) amax_run = fmaxf(fmaxf(amax_run, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
              } else if constexpr (RES) {
                v += rres[j % 3][q * NRB + i];
              }
              const unsigned off = ((unsigned)m * (unsigned)g.N + (unsigned)(nbase + q * CW + 4 * cc)) * 4u;
              q8_st128_aux<TT_Q8S_ST_AUX>(g.C, out_bytes, off, __builtin_bit_cast(u32x4, v));
            } else {
              f32x4 v0 = *reinterpret_cast<const f32x4*>(scr + row * (CW * 4) + (((2 * cc) ^ esw(row)) << 4));
              f32x4 v1 = *reinterpret_cast<const f32x4*>(scr + row * (CW * 4) + (((2 * cc + 1) ^ esw(row)) << 4));
              v0 += bias_lo[mt][q];
              v1 += bias_hi[mt][q];
              float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
              if constexpr (BOTH) {
                const unsigned offc = ((unsigned)m * (unsigned)g.N + (unsigned)(nbase + q * CW + 8 * cc)) * 4u;
                q8_st128_aux<TT_Q8S_ST_AUX>(g.C, out_bytes, offc, __builtin_bit_cast(u32x4, v0));
                q8_st128_aux<TT_Q8S_ST_AUX>(g.C, out_bytes, offc + 16u, __builtin_bit_cast(u32x4, v1));
              }
              if constexpr (ACT) {
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
#ifdef TT_Q8_GELU_SCALAR   // (A/B: the round-4 form)
                  v[e] = gelu_fast_f(v[e]);
                  v[e + 1] = gelu_fast_f(v[e + 1]);
#else
                  const tt_f32x2 gq = gelu_fast_f2(tt_f32x2{v[e], v[e + 1]});
                  v[e] = gq[0];
                  v[e + 1] = gq[1];
#endif
                }
              }
              f16x8 qh, ql;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                _Float16 hi_, lo_;
                split_pair(v[e], hi_, lo_);
                qh[e] = hi_;
                ql[e] = lo_;
              }
              {
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                const u32x4 hb = __builtin_bit_cast(u32x4, qh);
                unsigned mx = 0;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4)
                  mx = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, mx), __builtin_bit_cast(u16x2, hb[w4] & 0x7fff7fffu)));
                if (m < g.M) hi_max = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, hi_max), __builtin_bit_cast(u16x2, mx)));
              }
              const unsigned off = ((unsigned)m * (unsigned)g.N + (unsigned)nbase) * 4u + (unsigned)(q * CW + 8 * cc) * 2u;
              q8_st128_aux<TT_Q8S_ST_AUX>(g.Cp, out_bytes, off, __builtin_bit_cast(u32x4, qh));
              q8_st128_aux<TT_Q8S_ST_AUX>(g.Cp, out_bytes, off + 64u, __builtin_bit_cast(u32x4, ql));
            }
          }
        }
      }
    }
    post_epi = 2;
    post_half = half;
    if constexpr (!F32OUT) range_flag_raise(g.range_flag, (hi_max & 0xffffu) >= 0x7c00u || (hi_max >> 16) >= 0x7c00u);
  };

  // ---- K-split item (as above)
  auto slice_reduce = [&]() -> bool {
    const int j = cu / g.ks_S, sl = cu - j * g.ks_S;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) { a1[a][m][e] = fmaf(a2[a][m][e], 0.00048828125f, a1[a][m][e]); a2[a][m][e] = 0.f; }
    float* wbase = g.ks_ws + ((size_t)j * g.ks_S * 8 + wave) * 4096;
    const unsigned wbytes = (unsigned)g.ks_S * 8u * 4096u * 4u;
    const unsigned sstride = 8u * 4096u * 4u;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {a1[a][m][4 * q], a1[a][m][4 * q + 1], a1[a][m][4 * q + 2], a1[a][m][4 * q + 3]};
          q8_st128_sys(wbase, wbytes, (unsigned)sl * sstride + (unsigned)(((a * 2 + m) * 4 + q) * 1024 + lane * 16), __builtin_bit_cast(u32x4, v));
        }
    q8_wait_vmcnt<0>();
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(g.ks_cnt + j * 8 + wave, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != g.ks_S - 1) return false;
    if (lane == 0) __hip_atomic_store(g.ks_cnt + j * 8 + wave, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int t = 0; t < g.ks_S; ++t) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = q8_ld128_sys(wbase, wbytes, (unsigned)t * sstride + (unsigned)(((a * 2 + m) * 4 + q) * 1024 + lane * 16));
#pragma unroll
            for (int e = 0; e < 4; ++e) a2[a][m][4 * q + e] = v[e];
          }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int e = 0; e < 16; ++e) a1[a][m][e] = t == 0 ? a2[a][m][e] : a1[a][m][e] + a2[a][m][e];
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) a2[a][m][e] = 0.f;
    return true;
  };

  // ---- prologue: what the iterations -3 .. -1 of the steady schedule would have issued, in its order: [X1(-1): none,] W / X0 of K-tile 0 |
  // X1(0), W / X0 of K-tile 1 | X1(1), W / X0 of K-tile 2 (the pieces of K-tiles that do not exist are out of range: zero-filled)
#pragma unroll
  for (int v = 0; v < 3; ++v) {
    const int sx1 = (v == 2) ? SLOT_B : 0, swx = v * SLOT_B;   // the lag's slot (K-tile v - 1), the cursor's (K-tile v)
    if (TT_Q8S_LOADER == 0 || late == (TT_Q8S_LOADER == 1)) {
#pragma unroll
      for (int k = (v == 0 ? 2 : 0); k < 6; ++k) site(k, sx1, swx, TT_Q8S_LOADER ? 2 : 0);   // (v == 0: there is no lag yet)
    }
    cursor_advance();
  }

  using NO = std::false_type; using YES = std::true_type;
  for (int it = 0; it < n_items; ++it) {
    int row0, n0, c_kt0, c_kend;
    bool c_half;
    item(it, row0, n0, c_half, c_kt0, c_kend);
    // the item's first fragments (W and X0 of its first K-tile) are read here, exposed: no fragment is live across an epilogue.  The ring
    // holds them: the barrier at the top of the previous iteration followed every wave's wait for them (the first item: waited for here).
    if (it == 0) {
      q8_wait_vmcnt<NP>();
      __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      rd_w(s0, 0, c4);
      rd_w(s0, 1, c4);
      rd_x(0, s0, c4);
    }
    const int nkt = c_kend - c_kt0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) { a1[a][m][e] = 0.f; a2[a][m][e] = 0.f; }
    auto run = [&](auto half_c, auto late_c) {
      for (int t = 0; t + 1 < nkt; ++t) ktile(NO{}, half_c, late_c);
      ktile(YES{}, half_c, late_c);
    };
    // (the last template argument: "late sites" without a loader role; "is a loader" with one - 1: waves 4 - 7, 2: waves 0 - 3)
    const bool role = TT_Q8S_LOADER == 2 ? !late : late;
    if (c_half) { if (role) run(YES{}, YES{}); else run(YES{}, NO{}); }
    else { if (role) run(NO{}, YES{}); else run(NO{}, NO{}); }
    if (has_slice && it == n_whole) {
      if (!slice_reduce()) continue;
    }
#ifdef TT_Q8S_STAMP
    Q8S_STAMP(se);
    st_body += (unsigned)(se - sd);
    epilogue(row0, n0, c_half);
    Q8S_STAMP(sa);
    st_epi += (unsigned)(sa - se); ++st_nepi;
    sd = 0;
#else
    epilogue(row0, n0, c_half);
#endif
  }
#ifdef TT_Q8S_STAMP
  if (g.order_mode >= 100 && lane == 0 && (wave == 0 || wave == 3 || wave == 4 || wave == 7) && (blockIdx.x == 3 || blockIdx.x == 200)) {
    const unsigned long long dt = __builtin_amdgcn_s_memtime() - clk_t0, dr = __builtin_amdgcn_s_memrealtime() - clk_r0;
    printf("q8s stamps: block %d wave %d: %u K-tiles, cycles per K-tile: lgkmcnt(0) %.0f | vmcnt wait %.0f | barrier %.0f | body (24 MFMAs + reads + DMA) %.0f ; "
           "%u epilogues of %.0f cycles ; kernel %llu cycles, %.3f GHz\n", (int)blockIdx.x, wave, st_n, (double)st_lgkm / st_n, (double)st_vm / st_n, (double)st_bar / st_n,
           (double)st_body / st_n, st_nepi, st_nepi ? (double)st_epi / st_nepi : 0.0, dt, (double)dt / (double)dr * 0.1);
  }
#endif
  if constexpr (GG) {
    if (g.amax_out) amax_publish(g.amax_out, amax_run);   // (uniform; every lane of the wave is here)
  }
#undef Q8S_FENCE
}

// =====================================================================================================================================
// gemm_pairs4_kernel (round 6, EXPERIMENT behind the knob TT_Q4): the same product, operands, lane maps and epilogues as gemm_pairs8s_kernel
// in FOUR-wave workgroups on 128 x 128 tiles, TWO workgroups per CU.  Why: an item of the 8-wave kernel ends with an epilogue in which no
// wave issues an MFMA (10 - 29 % of a launch: profiles/r05_q8s_ablation.txt), and its eight waves share one barrier per K-tile; two
// independent workgroups per CU interleave - one's epilogue, barrier waits and DMA-issue stalls run under the other's MFMAs.  Price: the W
// tile is re-read per 128 (not 256) x rows (+ 33 % LDS-DMA bytes per flop) and each workgroup's ring is TWO K-tiles (2 x 32 KB + 8 KB of
// scratch = 72 KB; two workgroups = 144 of the 160 KB) with the fragments of K-tile t + 1 prefetched into registers during K-tile t:
//   iteration t:  [lgkmcnt(0) | vmcnt: the pieces of K-tile t + 1, issued in iteration t - 1 | barrier]
//                 24 MFMAs on the registers of K-tile t; between them the 8 LDS-DMA pieces of K-tile t + 2 into the slot of K-tile t
//                 (its fragments went to registers in iteration t - 1: every wave's reads completed before this barrier), and the 16
//                 fragment reads of K-tile t + 1, each into the register its last reader of K-tile t has just been issued from.
// No half tiles (the tile IS the half), no K-split; items dealt round-robin to the 2 x CUs workgroups, consecutive tiles to one XCD.
#ifndef TT_Q4_SITES
#define TT_Q4_SITES 2
#endif
// Half items (the left-over tiles cut in two: 64 x rows, x block 0 of every wave only) are a RUN-TIME property of an item here - uniform branches
// around the second phase and the refresh sites; as a template parameter of the K-tile / epilogue lambdas (four + two copies of the code) the
// kernel spilled 15 - 45 registers.
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_pairs4_kernel(Q8Args g) {
  constexpr int ROWB = 128, CPR = 8, WIN = 2, RPI = 8;
  constexpr int CHUNK_B = 128 * ROWB;       // 16 KB: 128 rows of one operand
  constexpr int SLOT_B = 2 * CHUNK_B;       // W, X
  constexpr int RING_B = 2 * SLOT_B;
  constexpr int SCR_B = 2048;
  constexpr int CW = SCR_B / 128;
  constexpr int NPASS = 32 / CW;
  constexpr int BN = 128;
  constexpr bool F32OUT = EPI == Q8_F32 || EPI == Q8_F32_RES || EPI == Q8_F32_GELUGRAD;
  constexpr bool RES = EPI == Q8_F32_RES || EPI == Q8_F32_GELUGRAD;
  constexpr bool GG = EPI == Q8_F32_GELUGRAD;
  constexpr bool BOTH = EPI == Q8_BOTH || EPI == Q8_BOTH_GELU;
  constexpr bool ACT = EPI == Q8_PAIR_GELU || EPI == Q8_BOTH_GELU;
  constexpr int ST_TILE = BOTH ? 8 : 4;               // stores a wave issues per 32 x 32 MFMA tile
  constexpr int ST_FULL = 4 * ST_TILE;
  static_assert(CW == 16 && ST_FULL < 56, "scratch / counted waits");
  __shared__ __attribute__((aligned(16))) unsigned char smem[RING_B + 4 * SCR_B];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave & 1, wc = wave >> 1;          // 2 x 2 waves of 64 x 64
  const int r = lane & 31, h = lane >> 5;
  const int K4 = g.K * 4, nk = g.K / 32;

  // ---- work items: tile = it * nwg + wg; consecutive wg on one XCD (they share the x rows of a row block / the W column tile)
  const int nwg = g.ncu;
  int wg = blockIdx.x;
  if ((nwg & 7) == 0) wg = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  // n_half > 0: g.n_full whole tiles per workgroup, then the left-over tiles cut into HALVES (64 x rows: x block 0 of every wave only),
  // workgroup h < n_half takes half h & 1 of tile n_full * nwg + h / 2 - a left-over round costs ~ 0.55 of a tile instead of a whole one
  // g.n_half > 0: g.n_full whole tiles per workgroup, then workgroup h < n_half takes half h & 1 of tile n_full * nwg + h / 2; else round-robin
  const bool has_half = wg < g.n_half;
  const int n_whole = g.n_half > 0 ? g.n_full : (wg < g.ntiles ? (g.ntiles - wg + nwg - 1) / nwg : 0);
  const int n_items = n_whole + (has_half ? 1 : 0);
  if (n_items == 0) return;
  auto item = [&](int it, int& row0, int& n0, bool& half) {
    half = has_half && it == n_whole;
    const int tile = half ? g.n_full * nwg + (wg >> 1) : it * nwg + wg;
    const int mb = tile / g.ntn, ns = tile - mb * g.ntn;
    row0 = mb * 128 + (half ? (wg & 1) * 64 : 0);
    n0 = ns * BN;
  };

  // ---- LDS-DMA lane map (as gemm_pairs8s_kernel): lane -> (row, 16-byte slot) of a 1 KB piece (8 rows); four waves cover 32 rows per
  // site, four sites per 128-row chunk.  Rows beyond the operand and K-tiles that do not exist are out of the descriptor's range: zeros.
  const int l_row = lane / CPR, l_slot = lane % CPR;
  const int d_row0 = wave * RPI + l_row;
  const int d_chunk = l_slot ^ ((d_row0 / WIN) & (CPR - 1));
  const unsigned lane_voff = (unsigned)(d_row0 * K4 + d_chunk * 16);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(g.X), 0, (unsigned)g.M * (unsigned)K4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(g.W), 0, (unsigned)g.N * (unsigned)K4, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const unsigned site_step = 4u * RPI * (unsigned)K4;   // 32 rows

  // ---- DMA cursor: the K-tile whose pieces are issued next (two K-tiles ahead of the MFMAs)
  int d_item = 0, d_kt = 0;
  bool d_done = false, d_half = false;
  unsigned cur_w = OOB, cur_x = OOB;
  auto cursor_item = [&]() {
    int row0, n0;
    item(d_item, row0, n0, d_half);
    d_kt = 0;
    cur_w = (unsigned)n0 * (unsigned)K4;
    cur_x = (unsigned)row0 * (unsigned)K4;
  };
  cursor_item();
  auto cursor_advance = [&]() {
    if (d_done) return;
    ++d_kt;
    cur_w += ROWB;
    cur_x += ROWB;
    if (d_kt == nk) {
      ++d_item;
      if (d_item >= n_items) { d_done = true; cur_w = OOB; cur_x = OOB; }
      else cursor_item();
    }
  };
  // site k of the cursor's K-tile into ring slot `slot`: k = 0 .. 3 W rows [32 k, 32 k + 32), 4 .. 7 X rows
  auto site = [&](int k, int slot) {
    if (k < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (q8_lds_ptr_t)(smem + slot + k * 4096 + wave * 1024), 16, lane_voff + cur_w + (unsigned)k * site_step, 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (q8_lds_ptr_t)(smem + slot + CHUNK_B + (k - 4) * 4096 + wave * 1024), 16,
                                                  (k >= 6 && d_half) ? OOB : lane_voff + cur_x + (unsigned)(k - 4) * site_step, 0, 0, 0);   // (a half item has 64 x rows)
  };

  // ---- fragment addressing (as gemm_pairs8s_kernel)
  const int f_sw = (r / WIN) & (CPR - 1);
  int fo[4];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) fo[c4] = r * ROWB + (((2 * c4 + h) ^ f_sw) << 4);
  int x_slice = CHUNK_B + wr * 64 * ROWB;   // (a half item: wr * 32 - set per item)
  const int w_slice = wc * 64 * ROWB;

  f32x16 a1[2][2], a2[2][2];   // [x block][w block]
  f16x8 Wf[2][4];              // [w block][hi k-step 0, 1, lo k-step 0, 1]
  f16x8 Xf[2][4];              // [x block][...]
  auto rd_w = [&](int slot, int mt, int c4) { Wf[mt][c4] = *reinterpret_cast<const f16x8*>(smem + slot + w_slice + mt * 32 * ROWB + fo[c4]); };
  auto rd_x = [&](int slot, int xt, int c4) { Xf[xt][c4] = *reinterpret_cast<const f16x8*>(smem + slot + x_slice + xt * 32 * ROWB + fo[c4]); };

  int s_cur = 0, s_nxt = SLOT_B;   // ring slots of K-tile t (refilled with t + 2 during iteration t) and t + 1 (read during iteration t)
  bool post_epi = false, post_half = false;
#define Q4_FENCE() __builtin_amdgcn_sched_barrier(0)
  auto ktile = [&](auto last_c, const bool HALF) {
    constexpr bool LAST = decltype(last_c)::value;   // the item's last K-tile: no refresh (the fragments would be live across the epilogue)
    // HALF (uniform, run time): x block 0 only - 12 MFMAs, the W refresh in its one phase
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (post_epi) {                                   // (the epilogue's stores are younger than the pieces waited for)
      if (post_half) q8_wait_vmcnt<ST_FULL / 2>(); else q8_wait_vmcnt<ST_FULL>();
      post_epi = false;
    } else q8_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    Q4_FENCE();
    auto slot = [&](auto ph_c, int q) {
      constexpr int PH = decltype(ph_c)::value;
      const int ks = q >> 1, mt = q & 1;
      a2[PH][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[mt][ks], Xf[PH][2 + ks], a2[PH][mt], 0, 0, 0);
      Q4_FENCE();
      // the eight pieces of K-tile t + 2, all in the FIRST phase (TT_Q4_SITES = 2: two per slot; 1: one per slot over both phases): a piece
      // issued in the last slot had the whole memory latency exposed at the next iteration's wait
      if constexpr (PH == 0) site(2 * q, s_cur);
      Q4_FENCE();
      a1[PH][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[mt][ks], Xf[PH][ks], a1[PH][mt], 0, 0, 0);
      Q4_FENCE();
      if constexpr (PH == 0) site(2 * q + 1, s_cur);
      Q4_FENCE();
      a2[PH][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[mt][2 + ks], Xf[PH][ks], a2[PH][mt], 0, 0, 0);
      Q4_FENCE();
      if constexpr (!LAST) {
        // in-place refresh from K-tile t + 1: X block PH's k-step ks is done once its second w block (mt = 1) is; in the last phase every
        // slot's own two W fragments are done behind its last MFMA
        if (mt == 1) { rd_x(s_nxt, PH, ks); rd_x(s_nxt, PH, 2 + ks); }
        if (PH == 1 || HALF) { rd_w(s_nxt, mt, ks); rd_w(s_nxt, mt, 2 + ks); }
      }
      Q4_FENCE();
    };
    using PH0 = std::integral_constant<int, 0>; using PH1 = std::integral_constant<int, 1>;
    slot(PH0{}, 0); slot(PH0{}, 1); slot(PH0{}, 2); slot(PH0{}, 3);
    if (!HALF) { slot(PH1{}, 0); slot(PH1{}, 1); slot(PH1{}, 2); slot(PH1{}, 3); }
    cursor_advance();
    const int t_ = s_cur; s_cur = s_nxt; s_nxt = t_;
  };

  // ---- epilogue of one item (gemm_pairs8s_kernel's, through the wave's private scratch)
  unsigned char* scr = smem + RING_B + wave * SCR_B;
  constexpr int CPRW = CW / 4;
  constexpr int LPR = F32OUT ? CPRW : CPRW / 2;
  constexpr int RPW = 64 / LPR;
  constexpr int NRB = 32 / RPW;
  constexpr int NLD = NPASS * NRB;
  auto esw = [](int row) { return (row >> 2) & (CPRW - 1); };
  const unsigned out_bytes = (unsigned)g.M * (unsigned)g.N * 4u;
  float amax_run = 0.f;
  auto epilogue = [&](int row0, int n0, const bool half) {
    constexpr int NT = 4;
    const int nt = half ? 2 : NT;
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int r = lane_e & 31, h = lane_e >> 5;
    const int rr = lane_e / LPR, cc = lane_e % LPR;
    unsigned hi_max = 0;
    const float inv_s = g.out_scale ? 1.0f / *g.out_scale : 1.0f;
    auto mrow = [&](int j) { return half ? row0 + wr * 32 : row0 + wr * 64 + (j >> 1) * 32; };
    auto ncol = [&](int j) { return n0 + wc * 64 + (j & 1) * 32; };
    f32x4 bias_lo[2][NPASS], bias_hi[2][NPASS];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < NPASS; ++q) {
        const int n = ncol(b) + q * CW + (F32OUT ? 4 : 8) * cc;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        bias_lo[b][q] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + n) : zero;
        if constexpr (!F32OUT) bias_hi[b][q] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + n + 4) : zero;
      }
    f32x4 rres[3][RES ? NLD : 1];
    auto prefetch = [&](int j, int slot) {
      if constexpr (RES) {
        const int mbase = mrow(j), nbase = ncol(j);
#pragma unroll
        for (int q = 0; q < NPASS; ++q)
#pragma unroll
          for (int i = 0; i < NRB; ++i) {
            const unsigned off = ((unsigned)(mbase + i * RPW + rr) * (unsigned)g.N + (unsigned)(nbase + q * CW + 4 * cc)) * 4u;
            rres[slot][q * NRB + i] = q8_ld128(g.residual, out_bytes, off);
          }
      }
    };
    if constexpr (RES) {
      prefetch(0, 0);
      prefetch(1, 1);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
     if (j < nt) {
      const int ha = j >> 1, mt = j & 1;
      const int mbase = mrow(j), nbase = ncol(j);
      if constexpr (RES) {
        if (j + 2 < nt) prefetch(j + 2, (j + 2) % 3);
      }
#pragma unroll
      for (int q = 0; q < NPASS; ++q) {
#pragma unroll
        for (int gg = 0; gg < CW / 8; ++gg) {
          const int gi = q * (CW / 8) + gg;
          const int phys = (2 * gg + h) ^ esw(r);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaf(a2[ha][mt][4 * gi + e], 0.00048828125f, a1[ha][mt][4 * gi + e]) * inv_s;
          *reinterpret_cast<f32x4*>(scr + r * (CW * 4) + phys * 16) = v;
        }
#pragma unroll
        for (int i = 0; i < NRB; ++i) {
          const int row = i * RPW + rr;
          const int m = mbase + row;
          if constexpr (F32OUT) {
            f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * (CW * 4) + ((cc ^ esw(row)) << 4));
            v += bias_lo[mt][q];
            if constexpr (GG) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_fast_f(rres[j % 3][q * NRB + i][e]);
              if (m < g.M) amax_run = fmaxf(fmaxf(amax_run, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
            } else if constexpr (RES) {
              v += rres[j % 3][q * NRB + i];
            }
            const unsigned off = ((unsigned)m * (unsigned)g.N + (unsigned)(nbase + q * CW + 4 * cc)) * 4u;
            q8_st128(g.C, out_bytes, off, __builtin_bit_cast(u32x4, v));
          } else {
            f32x4 v0 = *reinterpret_cast<const f32x4*>(scr + row * (CW * 4) + (((2 * cc) ^ esw(row)) << 4));
            f32x4 v1 = *reinterpret_cast<const f32x4*>(scr + row * (CW * 4) + (((2 * cc + 1) ^ esw(row)) << 4));
            v0 += bias_lo[mt][q];
            v1 += bias_hi[mt][q];
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if constexpr (BOTH) {
              const unsigned offc = ((unsigned)m * (unsigned)g.N + (unsigned)(nbase + q * CW + 8 * cc)) * 4u;
              q8_st128(g.C, out_bytes, offc, __builtin_bit_cast(u32x4, v0));
              q8_st128(g.C, out_bytes, offc + 16u, __builtin_bit_cast(u32x4, v1));
            }
            if constexpr (ACT) {
#pragma unroll
              for (int e = 0; e < 8; e += 2) {
                const tt_f32x2 gq = gelu_fast_f2(tt_f32x2{v[e], v[e + 1]});
                v[e] = gq[0];
                v[e + 1] = gq[1];
              }
            }
            f16x8 qh, ql;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              _Float16 hi_, lo_;
              split_pair(v[e], hi_, lo_);
              qh[e] = hi_;
              ql[e] = lo_;
            }
            {
              typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
              const u32x4 hb = __builtin_bit_cast(u32x4, qh);
              unsigned mx = 0;
#pragma unroll
              for (int w4 = 0; w4 < 4; ++w4)
                mx = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, mx), __builtin_bit_cast(u16x2, hb[w4] & 0x7fff7fffu)));
              if (m < g.M) hi_max = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, hi_max), __builtin_bit_cast(u16x2, mx)));
            }
            const unsigned off = ((unsigned)m * (unsigned)g.N + (unsigned)nbase) * 4u + (unsigned)(q * CW + 8 * cc) * 2u;
            q8_st128(g.Cp, out_bytes, off, __builtin_bit_cast(u32x4, qh));
            q8_st128(g.Cp, out_bytes, off + 64u, __builtin_bit_cast(u32x4, ql));
          }
        }
      }
     }
    }
    post_epi = true;
    post_half = half;
    if constexpr (!F32OUT) range_flag_raise(g.range_flag, (hi_max & 0xffffu) >= 0x7c00u || (hi_max >> 16) >= 0x7c00u);
  };

  // ---- prologue: K-tile 0 -> slot 0, K-tile 1 -> slot 1 (K-tiles that do not exist: zero-filled)
#pragma unroll
  for (int k = 0; k < 8; ++k) site(k, 0);
  cursor_advance();
#pragma unroll
  for (int k = 0; k < 8; ++k) site(k, SLOT_B);
  cursor_advance();
  q8_wait_vmcnt<8>();            // K-tile 0's eight pieces (K-tile 1's may still be in flight)
  __builtin_amdgcn_s_barrier();

  using NO = std::false_type; using YES = std::true_type;
  for (int it = 0; it < n_items; ++it) {
    int row0, n0;
    bool c_half;
    item(it, row0, n0, c_half);
    x_slice = CHUNK_B + wr * (c_half ? 32 : 64) * ROWB;
    // the item's first fragments, exposed: its first K-tile is in s_cur (waited for by every wave before the barrier at the top of the
    // previous iteration - the first item: by the prologue)
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      rd_w(s_cur, 0, c4);
      rd_w(s_cur, 1, c4);
      rd_x(s_cur, 0, c4);
      if (!c_half) rd_x(s_cur, 1, c4);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) { a1[a][m][e] = 0.f; a2[a][m][e] = 0.f; }
    for (int t = 0; t + 1 < nk; ++t) ktile(NO{}, c_half);
    ktile(YES{}, c_half);
    epilogue(row0, n0, c_half);
  }
  if constexpr (GG) {
    if (g.amax_out) amax_publish(g.amax_out, amax_run);
  }
#undef Q4_FENCE
}

template <int EPI>
static int launch_pairs4(const Q8Args& g, hipStream_t s) {
  hipLaunchKernelGGL((gemm_pairs4_kernel<EPI>), dim3(g.ncu), dim3(256), 0, s, g);
  TT_CHECK_LAUNCH("gemm_pairs4");
  return TT_OK;
}

template <int EPI, int DBG = 0>
static int launch_pairs8s(const Q8Args& g, hipStream_t s) {
  hipLaunchKernelGGL((gemm_pairs8s_kernel<EPI, DBG>), dim3(g.ncu), dim3(512), 0, s, g);
  TT_CHECK_LAUNCH("gemm_pairs8s");
  return TT_OK;
}

static int q8_order_mode() { return tuning_knob(KNOB_Q8_ORDER); }   // (+100: the stamp builds print)

template <int EPI, int DBG = 0>
static int launch_pairs8(const Q8Args& g, hipStream_t s) {
  hipLaunchKernelGGL((gemm_pairs8_kernel<EPI, DBG>), dim3(g.ncu), dim3(512), 0, s, g);
  TT_CHECK_LAUNCH("gemm_pairs8");
  return TT_OK;
}


// Shape / epilogue eligibility and the work decomposition.  Returns the epilogue kind or -1.
struct Q8Plan {
  int ntn, ncu, n_full, n_half, ks_S, ks_R;
  long long ntiles;
  bool small;   // fewer tiles than the persistent kernels want: theirs only under the knob TT_Q4_SMALL (the four-wave kernel's 128 x 128 tiles)
};
// kgroup: the K-tiles a K-tile range must be a multiple of - 3 for gemm_pairs8_kernel (its ring advances three K-tiles per loop trip), 1 for
// gemm_pairs8s_kernel (run-time ring slots, fragments refreshed in place)
static int pairs8_plan(bool has_residual, bool has_y, bool has_pairs, bool has_pre, bool has_gelu_pre, int M, int N, int K, int act, Q8Plan* pl,
                       bool allow_ksplit = true, int kgroup = 3) {
  if (N % 128 != 0 || K % (32 * kgroup) != 0 || M < 256) return -1;
  int epi = -1;
  if (has_gelu_pre) {
    if (has_y && !has_pairs && !has_pre && !has_residual && !act) epi = Q8_F32_GELUGRAD;
  } else if (has_pre) {
    if (!has_y && has_pairs && act && !has_residual) epi = Q8_BOTH_GELU;
  } else if (has_y && has_pairs) {
    if (!act && !has_residual) epi = Q8_BOTH;
  } else if (has_y) {
    if (!act) epi = has_residual ? Q8_F32_RES : Q8_F32;
  } else if (has_pairs && !has_residual) {
    epi = act ? Q8_PAIR_GELU : Q8_PAIR;
  }
  if (epi < 0) return -1;
  if (tuning_knob(KNOB_PAIRS8_NO_KEPT) != 0 && epi >= Q8_F32_GELUGRAD) return -1;   // tuning aid: A/B of the kept-frame routes
  // 32-bit buffer offsets
  if ((long long)M * K * 4 >= 0x7fffffffLL || (long long)N * K * 4 >= 0x7fffffffLL || (long long)M * N * 4 >= 0x7fffffffLL) return -1;
  const int ntm = (M + 255) / 256, ntn = N / 128;
  const long long ntiles = (long long)ntm * ntn;
  const int ncu_dev = device_cu_count();
  const long long R = ntiles / ncu_dev, rem = ntiles - R * ncu_dev;
  // K-split of the rem left-over tiles (Q8Args::ks_S): S = the workgroups available per tile, at most one K-tile triple each and at most 6
  // (the finishing wave reads S - 1 partials).  Taken when its estimated last-round time - the longest slice + the partial traffic, in
  // microseconds - beats the half tiles' (0.86 of a tile, tools/q8_nohalf.py) or the round-robin's whole tile by 10 %; knob
  // TT_Q8_KSPLIT = mode + 10 * cap: mode 0 off, 1 (default) only behind whole rounds, 2 also for grids of less than one round (which
  // without it go to the small-tile kernel below half a round; measured: no gain); cap = most slices per tile (0: 6).
  const int ksplit_knob = tuning_knob(KNOB_Q8_KSPLIT) % 10, ksplit_cap = tuning_knob(KNOB_Q8_KSPLIT) / 10;
  int ks_S = 0;
  if (allow_ksplit && ksplit_knob != 0 && rem > 0 && (R > 0 || ksplit_knob >= 2)) {
    const int U = K / (32 * kgroup), nk = K / 32;
    int S = (int)(ncu_dev / rem);
    if (S > U) S = U;
    if (S > (ksplit_cap > 0 ? ksplit_cap : 6)) S = ksplit_cap > 0 ? ksplit_cap : 6;
    if (S >= 2) {
      // microseconds (tools/q8_ksplit.py): 1.3 per K-tile + 3 for the epilogue; the exchange ~ 9 + 1 per slice (store, counter, the partials)
      const double t_tile = 1.3 * nk + 3.0;
      const double t_slice = 1.3 * kgroup * ((U + S - 1) / S) + 3.0 + 9.0 + 1.0 * S;
      const double t_else = (2 * rem <= ncu_dev && tuning_knob(KNOB_P8_NO_HALF) == 0) ? 0.86 * t_tile : t_tile;
      if (t_slice < 0.9 * t_else && (R > 0 || rem * S >= ncu_dev / 2)) ks_S = S;
    }
  }
  // a persistent grid that cannot fill the chip: the small-tile kernel does better.  Round 4: below half the CUs.  The symmetric kernel's
  // half tiles cost ~ 0.55 of a whole one, so a grid of 2 x tiles half items pays a little earlier - knob TT_Q8_MIN_TILES, round 5's default 96; round 6: 128 - with the step's chains on three streams (engine.TWO_STREAMS) the 100-tile launches of the projection
  // head do better on the general kernel's small workgroups, which share a CU with the other stream's kernel, than on one 160 KB workgroup per
  // CU: C2 7.06 -> 7.03 ms, C4 18.25 -> 18.14, C3 / C5 unchanged (profiles/r06_step_knob_sweeps.txt), although alone the persistent kernel wins
  // (measured, us, persistent / general kernel: 100 tiles [6272 x 512 x 1024, the head's third Linear] 28.7 / 41.7; 75 tiles [6304 x 384 x
  // 384 / x 1536] 16.0 / 12.9 and 38.7 / 35.2; 50 tiles [6272 x 256 x 512] 16.3 / 10.8)
  const int min_tiles = kgroup == 1 ? tuning_knob(KNOB_Q8_MIN_TILES) : ncu_dev / 2;
  pl->small = false;
  if (ntiles < min_tiles && ks_S == 0) {
    if (kgroup != 1 || tuning_knob(KNOB_Q4_SMALL) == 0) return -1;
    pl->small = true;
  }
  int ncu = (int)(ntiles < ncu_dev ? ntiles : ncu_dev), n_full = 0, n_half = 0;
  if (ks_S >= 2) {
    ncu = R > 0 ? ncu_dev : (int)rem * ks_S;
    n_full = (int)R;
  } else if (rem > 0 && 2 * rem <= ncu_dev && tuning_knob(KNOB_P8_NO_HALF) == 0) {
    ncu = ncu_dev;
    n_full = (int)R;
    n_half = (int)(2 * rem);
  }
  pl->ntn = ntn; pl->ntiles = ntiles; pl->ncu = ncu; pl->n_full = n_full; pl->n_half = n_half; pl->ks_S = ks_S; pl->ks_R = ks_S ? (int)rem : 0;
  return epi;
}

// which of the two persistent kernels a call goes to: the symmetric one (round 5: any K % 32 == 0) unless the knob TT_Q8_STREAM is 0 (A/B:
// the round-4 kernel, K % 96 == 0 only)
static int q8_kgroup(int K) { (void)K; return tuning_knob(KNOB_Q8_STREAM) != 0 ? 1 : 3; }

int pairs8_would_run(int M, int N, int K, int act, int has_residual, int has_y, int has_pairs, int has_pre, int has_gelu_pre) {
  Q8Plan pl;
  return pairs8_plan(has_residual != 0, has_y != 0, has_pairs != 0, has_pre != 0, has_gelu_pre != 0, M, N, K, act, &pl, true, q8_kgroup(K)) >= 0;
}

// Called by linear_pairs_impl (gemm_planes.hip).  Returns TT_OK after a launch, 1 when the shape / epilogue is not this kernel's (the caller
// then takes the general kernel), < 0 on a launch error.  pre_out: the fp32 pre-activation of a GELU layer (its y must then be pairs only);
// gelu_pre: the pre-activation whose gelu' multiplies a data-gradient product.
int pairs8_try(const void* x_pairs, const void* w_pairs, const float* bias, const float* residual, float* y, float* pre_out, void* y_pairs,
               const float* gelu_pre, const float* out_scale, int M, int N, int K, int act, void* ksplit_ws, size_t ksplit_ws_bytes_, int* range_flag,
               float* amax_out, hipStream_t s) {
  // (max |y| for the caller: the symmetric kernel's x gelu' epilogue publishes it; every other route leaves the call to the general kernel)
  if (amax_out && (!gelu_pre || q8_kgroup(K) != 1)) return 1;
  Q8Plan pl;
  // the K-split needs the caller's workspace (tt_linear_ksplit_workspace_bytes, counters zeroed by tt_linear_ksplit_workspace_init);
  // without one the left-over tiles are cut into halves / dealt round-robin
  KsplitWs kw{nullptr, nullptr};
  const bool have_ws = ksplit_ws_carve(ksplit_ws, ksplit_ws_bytes_, &kw);
  const int kgroup = q8_kgroup(K);
  const int epi = pairs8_plan(residual != nullptr, y != nullptr, y_pairs != nullptr, pre_out != nullptr, gelu_pre != nullptr, M, N, K, act, &pl, have_ws, kgroup);
  if (epi < 0) return 1;
  float* ks_ws = pl.ks_S >= 2 ? kw.partials : nullptr;
  int* ks_cnt = pl.ks_S >= 2 ? kw.counters : nullptr;
  Q8Args g{static_cast<const _Float16*>(x_pairs), static_cast<const _Float16*>(w_pairs), M, N, K, bias, gelu_pre ? gelu_pre : residual,
           pre_out ? pre_out : y, static_cast<_Float16*>(y_pairs), pl.ntn, (int)pl.ntiles, pl.ncu, pl.n_full, pl.n_half, q8_order_mode(),
           pl.ks_S, pl.ks_R, out_scale, ks_ws, ks_cnt, range_flag, amax_out};
#ifdef TT_Q8_ABLATE   // timing-study build only: TT_Q8_DBG selects a crippled instantiation
  if (kgroup == 3) {
    const char* e = getenv("TT_Q8_DBG");
    const int dbg = e ? atoi(e) : 0;
#define Q8_DBG_CASE(EV)                                  \
  if (epi == EV) {                                       \
    if (dbg == 1) return launch_pairs8<EV, 1>(g, s);     \
    if (dbg == 2) return launch_pairs8<EV, 2>(g, s);     \
    if (dbg == 8) return launch_pairs8<EV, 8>(g, s);     \
    if (dbg == 9) return launch_pairs8<EV, 9>(g, s);     \
    if (dbg == 10) return launch_pairs8<EV, 10>(g, s);   \
    if (dbg == 3) return launch_pairs8<EV, 3>(g, s);     \
    if (dbg == 11) return launch_pairs8<EV, 11>(g, s);   \
    if (dbg == 16) return launch_pairs8<EV, 16>(g, s);   \
  }
    Q8_DBG_CASE(Q8_F32) Q8_DBG_CASE(Q8_F32_RES) Q8_DBG_CASE(Q8_PAIR_GELU)
#undef Q8_DBG_CASE
  }
#endif
#ifdef TT_Q8_ABLATE   // timing-study build only: TT_Q8_DBG selects a crippled instantiation of the symmetric kernel too
  if (kgroup == 1) {
    const char* e = getenv("TT_Q8_DBG");
    const int dbg = e ? atoi(e) : 0;
#define Q8S_DBG_CASE(EV)                                   \
  if (epi == EV) {                                         \
    if (dbg == 1) return launch_pairs8s<EV, 1>(g, s);      \
    if (dbg == 2) return launch_pairs8s<EV, 2>(g, s);      \
    if (dbg == 8) return launch_pairs8s<EV, 8>(g, s);      \
    if (dbg == 9) return launch_pairs8s<EV, 9>(g, s);      \
    if (dbg == 10) return launch_pairs8s<EV, 10>(g, s);    \
    if (dbg == 3) return launch_pairs8s<EV, 3>(g, s);      \
    if (dbg == 11) return launch_pairs8s<EV, 11>(g, s);    \
    if (dbg == 4) return launch_pairs8s<EV, 4>(g, s);      \
    if (dbg == 12) return launch_pairs8s<EV, 12>(g, s);    \
    if (dbg == 13) return launch_pairs8s<EV, 13>(g, s);    \
  }
    Q8S_DBG_CASE(Q8_F32) Q8S_DBG_CASE(Q8_F32_RES) Q8S_DBG_CASE(Q8_PAIR_GELU)
#undef Q8S_DBG_CASE
  }
#endif
  // knob TT_Q4: 0 never; 1 every shape with at least one round of 128 x 128 tiles; 2 only the GELU epilogues (the long, VALU-bound ones: where
  // hiding the epilogue under the other workgroup's main loop pays most) at K <= 768
  const int q4 = tuning_knob(KNOB_Q4);
  const bool q4_gelu = epi == Q8_PAIR_GELU || epi == Q8_BOTH_GELU;
  if (kgroup == 1 && (pl.small || ((q4 == 1 || (q4 == 2 && q4_gelu && K <= 768) || (q4 == 3 && K <= 768)) &&
                                   (long long)((M + 127) / 128) * pl.ntn >= 2LL * device_cu_count()))) {
    // (experiment, default off) the four-wave kernel: 128 x 128 tiles, two workgroups per CU, at least one round of them - or (knob
    // TT_Q4_SMALL) the grids of less than 96 256 x 128 tiles that otherwise go to the general kernel's 64 x 64 tiles, one workgroup per tile
    Q8Args g4 = g;
    g4.ntiles = ((M + 127) / 128) * pl.ntn;
    g4.ncu = g4.ntiles < 2 * device_cu_count() ? g4.ntiles : 2 * device_cu_count();
    g4.n_full = g4.n_half = 0; g4.ks_S = g4.ks_R = 0; g4.ks_ws = nullptr; g4.ks_cnt = nullptr;
    {   // the left-over tiles as halves when they fit one round of workgroups
      const int R4 = g4.ntiles / g4.ncu, rem4 = g4.ntiles - R4 * g4.ncu;
      if (rem4 > 0 && 2 * rem4 <= g4.ncu && tuning_knob(KNOB_P8_NO_HALF) == 0) { g4.n_full = R4; g4.n_half = 2 * rem4; }
      // (TT_Q4_SMALL = 2: a grid of at most one workgroup per CU as twice as many half tiles)
      if (pl.small && tuning_knob(KNOB_Q4_SMALL) == 2 && g4.ntiles <= device_cu_count()) { g4.ncu = 2 * g4.ntiles; g4.n_full = 0; g4.n_half = 2 * g4.ntiles; }
    }
    switch (epi) {
      case Q8_F32: return launch_pairs4<Q8_F32>(g4, s);
      case Q8_F32_RES: return launch_pairs4<Q8_F32_RES>(g4, s);
      case Q8_PAIR: return launch_pairs4<Q8_PAIR>(g4, s);
      case Q8_PAIR_GELU: return launch_pairs4<Q8_PAIR_GELU>(g4, s);
      case Q8_F32_GELUGRAD: return launch_pairs4<Q8_F32_GELUGRAD>(g4, s);
      case Q8_BOTH: return launch_pairs4<Q8_BOTH>(g4, s);
      case Q8_BOTH_GELU: return launch_pairs4<Q8_BOTH_GELU>(g4, s);
      default: return 1;
    }
  }
  if (kgroup == 1) {
    switch (epi) {
      case Q8_F32: return launch_pairs8s<Q8_F32>(g, s);
      case Q8_F32_RES: return launch_pairs8s<Q8_F32_RES>(g, s);
      case Q8_PAIR: return launch_pairs8s<Q8_PAIR>(g, s);
      case Q8_PAIR_GELU: return launch_pairs8s<Q8_PAIR_GELU>(g, s);
      case Q8_F32_GELUGRAD: return launch_pairs8s<Q8_F32_GELUGRAD>(g, s);
      case Q8_BOTH: return launch_pairs8s<Q8_BOTH>(g, s);
      case Q8_BOTH_GELU: return launch_pairs8s<Q8_BOTH_GELU>(g, s);
      default: return 1;
    }
  }
  switch (epi) {
    case Q8_F32: return launch_pairs8<Q8_F32>(g, s);
    case Q8_F32_RES: return launch_pairs8<Q8_F32_RES>(g, s);
    case Q8_PAIR: return launch_pairs8<Q8_PAIR>(g, s);
    case Q8_PAIR_GELU: return launch_pairs8<Q8_PAIR_GELU>(g, s);
    case Q8_F32_GELUGRAD: return launch_pairs8<Q8_F32_GELUGRAD>(g, s);
    case Q8_BOTH: return launch_pairs8<Q8_BOTH>(g, s);
    case Q8_BOTH_GELU: return launch_pairs8<Q8_BOTH_GELU>(g, s);
    default: return 1;
  }
}

}  // namespace tt
