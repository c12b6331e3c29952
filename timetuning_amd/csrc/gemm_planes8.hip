// bf16-plane GEMM, persistent "8-phase" form - the forward nn.Linear sites of the frozen blocks
// (dino_vision_transformer.py:94-103,115-130; models.py:915-926):  y = act(x @ w^T + bias) (+ residual)
// on operands that are already resident in HBM as P planes of bf16 (gemm_planes.hip explains the planes: P = 1 is BASELINE
// C4's bf16 path, P = 3 the fp32-accurate split with six bf16 products per term).
//
// Why a second kernel: gemm_planes_kernel drains its LDS-DMA at one barrier per slab with all eight waves in the same phase, so the
// matrix pipe idles while everybody loads (round-2 PMC: matrix pipe 40 % busy, 43 % of wave cycles parked).  This one follows the
// guide's 256 x 256 "8-phase" structure (cdna_hip_programming.md section 5) and adds what the ViT shapes need on top of it:
//
//   * ONE workgroup of 8 waves per CU, the whole 160 KB of LDS in one array: a ring of two K-tiles + 8 x 4 KB (2 KB at P = 3)
//     of per-wave epilogue scratch.  Tile 256 x 256 x 64 at P = 1, 256 x 128 x 32 at P = 3.
//   * the waves run as two GROUPS (0-3 and 4-7 = the two waves of every SIMD) one barrier interval apart: while one group issues
//     its MFMAs the other reads fragments (ds_read_b128) and issues LDS-DMA for a later K-tile.  Two raw s_barrier per phase,
//     four phases per K-tile; a K-tile's operands are split in "chunks" (half tiles of 128 rows) that the phases consume one
//     after the other, so a phase needs at most one new chunk.
//   * LDS-DMA (buffer_load_dwordx4 ... lds) stays in flight ACROSS the barriers behind counted s_waitcnt vmcnt(N): chunk c is
//     issued D phases before the phase that needs it, the wait of phase f retires everything but the youngest L chunks, and
//     a chunk is read one phase after the wait that retired it.  The schedule tables below are checked at compile time
//     against the two hazards (RAW: DMA -> ds_read, WAR: ds_read -> DMA of the K-tile two later into the same slot).
//   * PERSISTENT: a workgroup walks a run of work items; the DMA cursor runs D chunks ahead of the compute cursor ACROSS item
//     boundaries, so an item's first K-tiles land while the previous one finishes - no per-tile prologue bubble.  The tiles that
//     do not make a whole round over the CUs are cut into HALF tiles (128 rows: the X1 chunk and the two phases that use it are
//     skipped) when that fills the chip better; workgroups with a half tile take it first or last by parity, which puts the two
//     populations half a tile apart in time so that their epilogues (the only HBM-heavy part) do not all hit memory at once.
//     (Starting the workgroups that have no half tile - and therefore half a tile of slack - a quarter of that slack apart was
//     measured too: no gain, 153 -> 159 us on ViT-B/16's fc2; not kept.)
//   * swapped operands: D = W_tile . X_tile^T puts an output ROW on a lane and 4 consecutive columns in 4 registers; the epilogue
//     stages one 32 x 32 MFMA tile at a time through the wave's private scratch (no workgroup barrier: both groups run their
//     epilogues concurrently) and leaves as whole row segments.  It is specialised at compile time (EPI), branch-free (buffer
//     loads / stores: rows beyond M are out of range and dropped by the hardware), residual rows are prefetched two MFMA tiles
//     ahead, and the stores are NOT waited for: they drain under the next tile's main loop, whose counted waits allow for them.
//
// LDS images and the fragment / DMA lane maps are those of gemm_planes.hip (rows of BK bf16, 16-byte chunks XOR-swizzled through
// the per-lane SOURCE address; conflict-free ds_read_b128 for v_mfma_f32_32x32x16_bf16).
//
// Measured and dropped (profiles/r03_p8_variants.txt): s_setprio around the MFMA clusters (+8 % time here - the clusters are pinned by
// sched_barrier anyway), a fifth chunk in flight (D = 7, L = 5 with phase 0 retiring its W0 reads before its barrier: +-1 %), the nt
// cache policy on either LDS-DMA stream (3-60 % slower), a start delay for the workgroups with slack (no gain), and the cursor / DMA issue /
// counted wait moved into the MFMA part of the phase so that the load part is fragment reads only (D = 7 / 6 at the same L: 13-16 %
// SLOWER - a DMA instruction that waits for the texture path blocks the in-order wave's next MFMAs).
// Where the time goes now (profiles/r03_p8_clock.txt, r03_p8_ablation.txt): in CYCLES the main loop is at 72 % (P = 1) / 86 % (P = 3) of
// the MFMA-only cycle count; the fragment reads cost no cycles beside the MFMAs but pull the clock from 2.4 to ~1.85 GHz (MFMA + LDS reads
// alone: same cycles, 1.85 GHz), so the wall-clock gap to the nominal peak is mostly the power the chip has for LDS reads + MFMAs.
//
// Round-3 measurements that shaped it (tools/p8_ablate.py, profiles/r03_p8_ablation.txt): first version 841 / 744 TFLOP/s on
// ViT-B/16's qkv / fc2 (gemm_planes_kernel: 627 / 701); its epilogue - a chain of runtime branches with every residual load
// consumed at once, an IEEE division inside the GELU (__frcp_rn) and a vmcnt(0) behind the stores, all CUs in lockstep - was
// 17-43 % of the launches.
#include "common.hpp"
#include <cstdlib>

namespace tt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// epilogue kinds (compile time)
enum { P8_BF16 = 0, P8_BF16_GELU = 1, P8_F32 = 2, P8_F32_RES = 3, P8_PL3_GELU = 4 };

struct P8Args {
  const __bf16* X;   // [P][M][K]
  const __bf16* W;   // [P][N][K]
  long long x_stride, w_stride;   // plane strides in elements
  int M, N, K;
  const float* bias;      // [N]
  const float* residual;  // [M][N] (P8_F32_RES; may alias C)
  float* C;               // [M][N] fp32 (P8_F32*)
  __bf16* Cp;             // [po][M][N] bf16 planes (P8_BF16*, P8_PL3_GELU)
  long long c_stride;
  int ntn, ntiles, ncu;   // column tiles, whole tiles, workgroups launched
  int n_full;             // whole tiles per workgroup in the first part (tiles [0, n_full * ncu)); with n_half = 0 and n_full = 0:
                          // all tiles dealt round-robin
  int n_half;             // half tiles that follow (tiles [n_full * ncu, ntiles) cut in two): workgroup h < n_half takes half h
  int clock_print;        // TT_P8_CLOCK diagnostic builds only
  int order_mode;         // order of the load part, see `reads_first`
  // K-split of the tiles beyond the last whole round (P = 1 only; gemm_pairs8.hip has the description): ks_S >= 2 workgroups per left-over
  // tile over contiguous ranges of K-tile pairs; per-wave fp32 partials through ks_ws, the last wave to arrive at the (tile, wave) counter
  // sums them in slice order and runs the epilogue
  int ks_S, ks_R;
  float* ks_ws;           // [ks_R][ks_S][8 waves][128 x 64 floats]
  int* ks_cnt;            // [ks_R][8], zero between launches
};

// Device helpers at namespace scope: the buffer builtins inside a generic lambda of the kernel template make clang's HOST pass
// drop the kernel's stub (ROCm 7.2: undefined __device_stub__ at load time, no diagnostic).
__device__ __forceinline__ void p8_dma16(const void* base, unsigned char* lds_dst, int voffset, int soffset) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds_dst, 16, voffset, soffset, 0, 0);
}
__device__ __forceinline__ f32x4 p8_ld128(const void* base, unsigned nbytes, unsigned voff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, nbytes, 0x00020000);
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ void p8_st128(void* base, unsigned nbytes, unsigned voff, u32x4 v) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, nbytes, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 0);
}
// sc0 sc1 accesses (write-through / around the XCD's L2): the K-split partials, whose slices may run on different XCDs
__device__ __forceinline__ f32x4 p8_ld128_sys(const void* base, unsigned nbytes, unsigned voff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, nbytes, 0x00020000);
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 17));
}
__device__ __forceinline__ void p8_st128_sys(void* base, unsigned nbytes, unsigned voff, u32x4 v) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, nbytes, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 17);
}

// SHIFT0 (see `phase`): phase 3 issues phase 0's DMA chunk as well.  Measured per plane count (tools/ab_planes.py, interleaved, bit-identical):
// P = 1 1.6-4.4 % faster on the four ViT-B/16 block shapes, P = 3 1.2-3.6 % slower on the ViT-S/16 ones - so it is on for P = 1 only.
// -DTT_P8_SHIFT0=0/1 forces it for both (timing studies).  The P = 3 counterpart - phase 1 (9 reads + the X1 chunk) handing its chunk to
// phase 2 (6 reads, no chunk: the fourth slot does not exist at P = 3), one phase later, phase 1's wait allowing one chunk less - was
// built and measured too: bit-identical, within +-1 % on all four ViT-S/16 shapes (profiles/r03_p8_order.txt), not kept.
#ifndef TT_P8_SHIFT0
#define TT_P8_SHIFT0 -1
#endif
template <int P>
constexpr bool p8_shift0() { return TT_P8_SHIFT0 < 0 ? P == 1 : TT_P8_SHIFT0 != 0; }

template <int N>
__device__ __forceinline__ void p8_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- the schedule, per P.  A K-tile is 4 chunk slots issued one per phase; chunk i of K-tile t has stream index 4 t + i.
template <int P>
struct P8Cfg;
template <>
struct P8Cfg<1> {
  static constexpr int BK = 64, NHW = 2, D = 6, L = 4, X1 = 3;
  // chunks in need order: W0, X0, W1, X1
  static constexpr bool exists(int i) { return true; }
  static constexpr bool is_x(int i) { return i & 1; }
  static constexpr int half(int i) { return i >> 1; }
  static constexpr int need(int i) { return i == 0 ? 0 : i - 1; }        // first phase (0..3) that reads the chunk
  static constexpr int last_read(int i) { return i == 0 ? 0 : i - 1; }   // W0 / W1 fragments stay in registers for the K-tile
};
template <>
struct P8Cfg<3> {
  static constexpr int BK = 32, NHW = 1, D = 5, L = 3, X1 = 2;
  // chunks: W, X0, X1, (none)
  static constexpr bool exists(int i) { return i < 3; }
  static constexpr bool is_x(int i) { return i >= 1; }
  static constexpr int half(int i) { return i == 2 ? 1 : 0; }
  static constexpr int need(int i) { return i == 2 ? 2 : 0; }
  static constexpr int last_read(int i) { return i == 2 ? 3 : 1; }
};

// Compile-time check of the two hazards for "phase f issues chunk f + D and then waits for all but the youngest L chunks":
//   RAW  chunk h (needed at phase N(h)) must have been retired by the wait of phase N(h) - 1:  h <= N(h) - 1 + D - L
//   WAR  chunk h of K-tile t + 2 is issued at phase h - D, which must be >= 2 phases after the last read of the same slot in K-tile t
template <int P>
constexpr bool p8_schedule_ok() {
  using C = P8Cfg<P>;
  for (int i = 0; i < 4; ++i) {
    if (!C::exists(i)) continue;
    if (!(i <= C::need(i) - 1 + C::D - C::L)) return false;
    if (!(8 + i - C::D >= C::last_read(i) + 2)) return false;
  }
  return C::D - C::L >= 1 && C::D <= 8;
}
// wave-instructions of the youngest L chunks after the issue of compute phase ph (chunks ph + D - L + 1 .. ph + D of the stream);
// gch = wave-instructions per wave and chunk.  half: the X1 chunk is not issued (half tiles) - also the safe (smaller) count
// while the window may still hold a slot of a half tile.
template <int P>
constexpr int p8_window(int ph, int gch, bool half) {
  using C = P8Cfg<P>;
  int n = 0;
  for (int k = 0; k < C::L; ++k) {
    const int idx = (ph + C::D - k) & 3;
    n += (C::exists(idx) && !(half && idx == C::X1)) ? gch : 0;
  }
  return n;
}
static_assert(p8_schedule_ok<1>() && p8_schedule_ok<3>(), "LDS-DMA schedule violates a RAW / WAR rule");

// DBG (timing studies only, tools/p8_ablate.py; the shipped instantiations are DBG = 0), a bit mask: 1 no MFMAs, 2 no LDS-DMA, 4 no fragment
// reads, 8 no epilogue (accumulators consumed by a dummy store), 16 epilogue without global loads / stores
template <int P, int EPI, int DBG = 0>
__global__ __launch_bounds__(512) void gemm_planes8_kernel(P8Args g) {
  using CF = P8Cfg<P>;
  constexpr int BK = CF::BK, NHW = CF::NHW, D = CF::D, L = CF::L;
  constexpr bool P8_SHIFT0 = p8_shift0<P>();
  constexpr int ROWB = BK * 2;              // bytes per LDS row
  constexpr int CPR = ROWB / 16;            // 16-byte chunks per row
  constexpr int WIN = 256 / ROWB;           // rows per 256-byte bank window
  constexpr int RPI = 64 / CPR;             // rows one LDS-DMA wave-instruction fills
  constexpr int NKS = BK / 16;              // MFMA k-steps per K-tile
  constexpr int JPW = (128 / RPI) / 8;      // DMA wave-instructions per wave, plane and chunk
  constexpr int GCH = P * JPW;              // ... per wave and chunk (the vmcnt unit)
  constexpr int PLANE_B = 128 * ROWB;       // one plane of one 128-row half
  constexpr int HALF_B = P * PLANE_B;
  constexpr int NSLOT = NHW + 2;
  constexpr int BUF_B = NSLOT * HALF_B;
  constexpr int RING_B = 2 * BUF_B;
  constexpr int SCR_B = (160 * 1024 - RING_B) / 8;   // per-wave epilogue scratch
  constexpr int CW = SCR_B / 128;                    // columns of a 32-row MFMA tile staged per pass (32 or 16)
  constexpr int NPASS = 32 / CW;
  constexpr int BN = 128 * NHW;
  constexpr int NF = P == 1 ? NKS : P;               // fragments per register set
  constexpr bool F32OUT = EPI == P8_F32 || EPI == P8_F32_RES, RES = EPI == P8_F32_RES, ACT = EPI == P8_BF16_GELU || EPI == P8_PL3_GELU;
  constexpr int PO = F32OUT ? 0 : (EPI == P8_PL3_GELU ? 3 : 1);
  // stores a wave issues per whole tile (the epilogue's loads come on top): what the counted waits of the following phases may
  // leave outstanding besides their DMA window.  A LOWER bound is what keeps those waits safe (see `epilogue`).
  constexpr int ST_TILE = F32OUT ? 4 : 2 * PO;                  // per 32 x 32 MFMA tile (fp32: 4 KB in 1 KB pieces; bf16: 2 KB per plane)
  constexpr int ST_FULL = 4 * NHW * ST_TILE, ST_HALF = ST_FULL / 2;
  constexpr int WMAX = GCH * (L + (P8_SHIFT0 ? 1 : 0));
  constexpr int S_FULL = ST_FULL < 63 - WMAX ? ST_FULL : 63 - WMAX, S_HALF = ST_HALF < 63 - WMAX ? ST_HALF : 63 - WMAX;
  static_assert(RING_B + 8 * SCR_B <= 160 * 1024 && (CW == 32 || CW == 16), "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char smem[160 * 1024];

#ifdef TT_P8_CLOCK   // diagnostic build only: the clock the chip holds under this kernel (s_memtime ticks per 100 MHz s_memrealtime tick)
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool grp1 = wave >= 4;
  const int wr = wave >> 2, wc = wave & 3;
  const int r = lane & 31, h = lane >> 5;
  const int K2 = g.K * 2, nk = g.K / BK;

  // ---- this workgroup's work items.  Workgroups that share an XCD (equal id mod 8) get adjacent runs of the row-major tile
  // order, so an activation row block is fetched into one L2 while the weight strips stay hot in all of them.
  int cu = blockIdx.x;
  if ((g.ncu & 7) == 0) cu = (blockIdx.x & 7) * (g.ncu >> 3) + (blockIdx.x >> 3);
  // Whole tiles are dealt INTERLEAVED: workgroup cu takes tiles cu, cu + ncu, cu + 2 ncu, ... of the row-major order (columns fastest),
  // so at any time the 32 workgroups of an XCD work on 32 CONSECUTIVE tiles - every activation row block is being read by all the
  // workgroups of its column strips at once and is fetched into that L2 once (contiguous runs per workgroup put them on different
  // row blocks: round-3 PMC, L2 hit rate 63 %).
  int n_whole;
  bool has_half = false;
  const bool ksplit = P == 1 && g.ks_S >= 2;
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
  // item -> first row, first column, half?, K-tile range [kt0, kend) (an even number of K-tiles)
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
      const int j = cu / g.ks_S, sl = cu - j * g.ks_S, U = nk / 2;
      tile = g.n_full * g.ncu + j;
      kt0 = 2 * (sl * U / g.ks_S);
      kend = 2 * ((sl + 1) * U / g.ks_S);
    } else {
      tile = (half_first ? it - 1 : it) * g.ncu + cu;
    }
    const int mb = tile / g.ntn, ns = tile - mb * g.ntn;
    row0 = mb * 256 + hsel * 128;
    n0 = ns * BN;
  };

  // ---- LDS-DMA lane map (see gemm_planes.hip): lane -> (row, slot) of the 1 KiB piece, source chunk = slot ^ f(row)
  const int l_row = lane / CPR, l_slot = lane % CPR;
  const int d_row0 = wave * RPI + l_row;                                  // image row of piece j = wave; j = wave + 8 i adds 8 i RPI
  const int d_chunk = l_slot ^ ((d_row0 / WIN) & (CPR - 1));              // (8 RPI rows further: same f)
  const int w_voff = d_row0 * K2 + d_chunk * 16;
  const int xps = (int)(g.x_stride * 2), wps = (int)(g.w_stride * 2);     // plane strides in bytes

  // DMA cursor (scalar state + the X voffsets of its item; rows beyond M are clamped to M - 1 and dropped at the store)
  int d_item = 0, d_kt = 0, d_kend = 0;
  bool d_done = false, d_half = false;
  int half_guard = 0;   // > 0: the DMA window may hold a slot of a half tile (no X1 chunk): the counted waits take the smaller count
  int d_kofs = 0;       // d_kt * ROWB
  int d_wbase = 0;      // first W row of the item * K2
  int x_voff[2][JPW];
  auto cursor_item = [&]() {
    int row0, n0;
    item(d_item, row0, n0, d_half, d_kt, d_kend);
    d_kofs = d_kt * ROWB;
    if constexpr (DBG & 32) row0 = 0, n0 = 0;   // (ablation) every workgroup streams the SAME operand tile: the DMA stream from a warm L2
    d_wbase = n0 * K2;
#pragma unroll
    for (int ha = 0; ha < 2; ++ha)
#pragma unroll
      for (int i = 0; i < JPW; ++i) {
        int row = row0 + ha * 128 + 8 * i * RPI + d_row0;
        row = row < g.M ? row : g.M - 1;
        x_voff[ha][i] = row * K2 + d_chunk * 16;
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
  bool steady = false;   // see `phase`
  // issue chunk IDX of the cursor's K-tile into ring buffer B
  auto issue = [&](auto idx_c, auto buf_c) {
    constexpr int IDX = decltype(idx_c)::value, B = decltype(buf_c)::value;
    if constexpr (CF::exists(IDX) && !(DBG & 2)) {
      if (!steady) {   // steady state: the cursor is neither done nor in a half tile
        if (d_done) return;
        if (IDX == CF::X1 && d_half) return;
      }
      constexpr int HA = CF::half(IDX);
      const int lds_base = B * BUF_B + IDX * HALF_B + wave * 1024;
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int i = 0; i < JPW; ++i) {
          unsigned char* dst = smem + lds_base + p * PLANE_B + i * 8 * 1024;
          if constexpr (CF::is_x(IDX))
            p8_dma16(g.X, dst, x_voff[HA][i], d_kofs + p * xps);
          else
            p8_dma16(g.W, dst, w_voff, d_wbase + (HA * 128 + 8 * i * RPI) * K2 + d_kofs + p * wps);
        }
    }
  };

  // ---- fragment addressing: image row = (slice of the half) + r, chunk 2 ks + h, swizzled by f(r) (slices are multiples of 32 rows)
  const int f_sw = (r / WIN) & (CPR - 1);
  int lo[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) lo[ks] = r * ROWB + (((2 * ks + h) ^ f_sw) << 4);
  const int x_slice = wr * 64 * ROWB, w_slice = wc * 32 * ROWB;

  f32x16 acc[2][NHW][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NHW; ++b)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][m][e] = 0.f;
  bf16x8 Wf[2][NF], Xf[2][NF];

  // ablation operands (DBG & 4): four per-lane pseudo-random bf16x8 vectors in [-2, 2), made once and handed out by the (compile-time)
  // fragment tag - NOT constants, because the clock the chip holds under MFMA load depends on the operand data (MI355X_MICROARCH.md
  // "DVFS give-back"), and no VALU work per fragment
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 rnd4[4];
  if constexpr (DBG & 4) {
    unsigned st = (unsigned)threadIdx.x * 2654435761u + (unsigned)blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        st = st * 1664525u + 1013904223u;
        rnd4[v][e] = (st & 0x80ff80ffu) | 0x3f003f00u | ((st >> 5) & 0x00800080u);
      }
  }
  auto ld = [&](int off, int tag) {
    if constexpr (DBG & 4) {
      return __builtin_bit_cast(bf16x8, rnd4[tag & 3]);
    } else {
      return *reinterpret_cast<const bf16x8*>(smem + off);
    }
  };

#ifdef TT_P8_STAMP   // diagnostic build only: where a wave's steady-state phase goes (s_memtime stamps, consumed behind the phase's own lgkmcnt(0))
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0, ts6 = 0;
  unsigned st_issue = 0, st_reads = 0, st_wait = 0, st_bar1 = 0, st_mfma = 0, st_bar2 = 0, st_n = 0;
#define P8_STAMP(t) asm volatile("s_memtime %0" : "=s"(t)::"memory")
#else
#define P8_STAMP(t)
#endif
  // Order of the load part.  The texture path takes one 1 KiB DMA instruction per ~11-16 cycles; when the four waves of a group all queue
  // theirs at the start of the load part the last one is accepted ~100 cycles later (phase stamps, tools/p8_stamp.py: DMA issue 120-150
  // cycles of a ~300-cycle load part against the partner's 250-cycle MFMA part).  order_mode (P8Args, TT_P8_ORDER): 0 every wave DMA
  // first, 1 every wave reads first, 2 odd waves read first, 3 (the default: 2-4 % faster at P = 1, ~1 % at P = 3, tools/p8_order.py)
  // waves 2, 3 (6, 7) of a group read first.  Measured and not kept: fragment reads issued in the order the MFMAs consume them with the
  // compiler's counted lgkmcnt before each MFMA instead of the lgkmcnt(0) behind the barrier (bit-identical, within +-1 % on seven of eight
  // shapes: the reads have landed by the time the barrier opens); the next phase's chunk issued behind this phase's MFMAs instead of in
  // the load part (same vmcnt positions, bit-identical; proj / fc2 3-6 % faster, fc1 and every P = 3 shape 2-6 % slower).
  const bool reads_first = g.order_mode == 1 || (g.order_mode == 2 && (wave & 1)) || (g.order_mode == 3 && (wave & 2));
  int post_epi = 0;          // phases left in which the stores of the last epilogue may still be outstanding
  bool post_half = false;    // ... and whether that epilogue was a half tile's
  bool c_half = false;       // the item being computed is a half tile

  // ---- one phase: [DMA issue | fragment reads | counted wait] barrier [MFMAs] barrier
  // `steady` (set per pair of K-tiles by the main loop): nothing rare can happen in these phases - whole tile being computed, DMA cursor
  // in a whole tile and not at the end of the stream, no stores of an epilogue left in the window - so ONE scalar branch skips all the
  // bookkeeping: the load part is then fragment reads, the cursor's scalar adds, the DMA instructions and one counted wait.  (The general
  // path carries ~10 scalar branches per phase; round-3 PMC showed the waves parked or issue-stalled for 72 % of their cycles.)
  auto phase = [&](auto buf_c, auto ph_c) {
    constexpr int B = decltype(buf_c)::value, PH = decltype(ph_c)::value;
    constexpr int base = B * BUF_B;
    const bool work = !(PH >= 2 && c_half);   // phases 2 and 3 of either schedule use the X1 chunk only
    auto frag_reads = [&]() {
    if (work) {
      if constexpr (P == 1) {
        // chunks W0 X0 W1 X1 in slots 0..3; quadrants (hA, hW): (0,0) (0,1) (1,1) (1,0)
        if constexpr (PH == 0) {
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) Wf[0][ks] = ld(base + 0 * HALF_B + w_slice + lo[ks], ks);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) Xf[mt][ks] = ld(base + 1 * HALF_B + x_slice + mt * 32 * ROWB + lo[ks], 1 + mt + ks);
        } else if constexpr (PH == 1) {
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) Wf[1][ks] = ld(base + 2 * HALF_B + w_slice + lo[ks], 2 + ks);
        } else if constexpr (PH == 2) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) Xf[mt][ks] = ld(base + 3 * HALF_B + x_slice + mt * 32 * ROWB + lo[ks], 3 + mt + ks);
        }
      } else {
        // chunks W X0 X1 in slots 0..2 (planes inside a chunk); phases (hA, ks): (0,0) (0,1) (1,1) (1,0)
        constexpr int HA = PH >> 1, KS = (PH == 1 || PH == 2) ? 1 : 0;
        if constexpr (PH < 2) {
#pragma unroll
          for (int p = 0; p < P; ++p) Wf[KS][p] = ld(base + 0 * HALF_B + p * PLANE_B + w_slice + lo[KS], p + KS);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int p = 0; p < P; ++p) Xf[mt][p] = ld(base + (1 + HA) * HALF_B + p * PLANE_B + x_slice + mt * 32 * ROWB + lo[KS], 1 + mt + p + HA);
      }
    }
    };
    // DMA: chunk (PH + D) of the stream = chunk (PH + D) & 3 of the K-tile (PH + D) / 4 further on
    constexpr int CI = (PH + D) & 3, BT = (B + (PH + D) / 4) & 1;
    // SHIFT0: phase 0 carries half of a K-tile's fragment reads (12 of 24 at P = 1, 9 of 30 at P = 3) and phase 3 the fewest, so phase 3
    // issues phase 0's chunk as well (one phase early: the same position in the vmcnt queue, the same K-tile of the cursor, still >= 2 phases
    // behind the last read of its slot - static_assert below) and phase 0 issues none.
    constexpr int CI0 = D & 3, BT0 = ((B ^ 1) + D / 4) & 1;
    static_assert(!P8_SHIFT0 || (CI0 != 0 && CF::exists(CI0) && CI0 != CF::X1 && 8 + CI0 - D - 1 >= CF::last_read(CI0) + 2),
                  "SHIFT0: phase 0's chunk must share the cursor's K-tile with phase 3's, exist in half tiles, and keep the WAR distance");
    auto dma_issue = [&]() {
      if constexpr (P8_SHIFT0 && PH == 0) return;
      if constexpr (CI == 0) cursor_next_ktile();
      issue(std::integral_constant<int, CI>{}, std::integral_constant<int, BT>{});
      if constexpr (P8_SHIFT0 && PH == 3) issue(std::integral_constant<int, CI0>{}, std::integral_constant<int, BT0>{});
    };
    // counted wait: everything but the youngest L chunks (and, for L phases behind an epilogue, its stores) has landed
    auto dma_wait = [&]() {
      // (SHIFT0, phase 3: one chunk more has been issued, the chunk to retire is the same)
      constexpr int EXTRA = (P8_SHIFT0 && PH == 3) ? GCH : 0;
      constexpr int WF = p8_window<P>(PH, GCH, false) + EXTRA, WH = p8_window<P>(PH, GCH, true) + EXTRA;
      if (steady && !(DBG & 2)) {
        p8_wait_vmcnt<WF>();
      } else {
        half_guard = (d_half && !d_done) ? L + 1 : (half_guard > 0 ? half_guard - 1 : 0);
        if (d_done || (DBG & 2)) {
          p8_wait_vmcnt<0>();
        } else if (post_epi > 0) {
          --post_epi;
          if (half_guard > 0 || post_half) p8_wait_vmcnt<WH + S_HALF>();
          else p8_wait_vmcnt<WF + S_FULL>();
        } else if (half_guard > 0) {
          p8_wait_vmcnt<WH>();
        } else {
          p8_wait_vmcnt<WF>();
        }
      }
    };
    if constexpr (DBG & 64) {   // timing study: the fragment reads ahead of the DMA instructions (3-4 % slower, profiles/r03_p8_variants.txt)
      frag_reads();
      dma_issue();
    } else {
      // DMA instructions, then fragment reads: the texture path works on them while the wave issues its ds_reads - except for the waves
      // that `reads_first` picks, which read first (one copy of the reads, two of the short DMA issue: no extra register pressure)
      P8_STAMP(ts0);
      if (!reads_first) dma_issue();
      __builtin_amdgcn_sched_barrier(0);
      P8_STAMP(ts1);
      frag_reads();
      __builtin_amdgcn_sched_barrier(0);
      if (reads_first) dma_issue();
      P8_STAMP(ts2);
    }
    dma_wait();
    P8_STAMP(ts3);
    __builtin_amdgcn_s_barrier();
    P8_STAMP(ts4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef TT_P8_STAMP
    if (steady) {   // (ts5 / ts6: the previous phase's MFMA part)
      st_issue += (unsigned)(ts1 - ts0); st_reads += (unsigned)(ts2 - ts1); st_wait += (unsigned)(ts3 - ts2); st_bar1 += (unsigned)(ts4 - ts3);
      if (ts6 > ts5 && ts0 > ts6) { st_mfma += (unsigned)(ts6 - ts5); st_bar2 += (unsigned)(ts0 - ts6); }
      ++st_n;
    }
    P8_STAMP(ts5);
#endif
    __builtin_amdgcn_sched_barrier(0);
    // MFMAs [lo, hi) of this phase's sequence (8 at P = 1, 12 at P = 3)
    auto mfmas = [&](auto lo_c, auto hi_c) {
      constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
      if constexpr (DBG & 1) {
        // keep the fragment reads alive without matrix work (one conversion + add per fragment)
        if constexpr (LO == 0) {
          float keep = 0.f;
#pragma unroll
          for (int i = 0; i < NF; ++i) keep += (float)Wf[0][i][0] + (float)Wf[1][i][0] + (float)Xf[0][i][0] + (float)Xf[1][i][0];
          acc[0][0][0][0] += keep;
        }
      } else if constexpr (P == 1) {
        constexpr int HA = PH >> 1, HW = (PH == 1 || PH == 2) ? 1 : 0;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            if (ks * 2 + mt >= LO && ks * 2 + mt < HI)
              acc[HA][HW][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[HW][ks], Xf[mt][ks], acc[HA][HW][mt], 0, 0, 0);
      } else {
        constexpr int HA = PH >> 1, KS = (PH == 1 || PH == 2) ? 1 : 0;
        int idx = 0;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int s = P - 1; s >= 0; --s)          // plane-index sum: small terms first (as gemm_planes_kernel)
#pragma unroll
            for (int pa = 0; pa <= s; ++pa, ++idx)
              if (idx >= LO && idx < HI)
                acc[HA][0][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[KS][s - pa], Xf[mt][pa], acc[HA][0][mt], 0, 0, 0);
      }
    };
    constexpr int NM = P == 1 ? 2 * NKS : 12;
    if (work) mfmas(std::integral_constant<int, 0>{}, std::integral_constant<int, NM>{});
    __builtin_amdgcn_sched_barrier(0);
    P8_STAMP(ts6);   // (all MFMAs issued)
    __builtin_amdgcn_s_barrier();
  };

  // ---- epilogue of one item: per wave, through its private scratch, no workgroup barrier.
  // Lane (m = r) holds columns 8 g + 4 h + {0..3} of a 32 x 32 MFMA tile in registers 4 g .. 4 g + 3.  A pass stages CW columns
  // ([32][CW] fp32, 16-byte chunks XOR-swizzled by the row) and reads them back row-major: fp32 outputs 4 columns per lane
  // (rows of CW * 4 bytes leave whole), bf16 outputs 8 columns per lane (16-byte stores).
  // Every global access is a buffer access on [M][N]: rows >= M are beyond the range and dropped, so the instruction count per
  // wave is FIXED - ST_FULL (ST_HALF) stores plus the bias / residual loads - which is what lets the next phases' waits count.
  unsigned char* scr = smem + RING_B + wave * SCR_B;
  constexpr int CPRW = CW / 4;                    // 16-byte chunks per staged row
  constexpr int LPR = F32OUT ? CPRW : CPRW / 2;   // lanes per staged row on the way back
  constexpr int RPW = 64 / LPR;                   // rows per read-back instruction
  constexpr int NRB = 32 / RPW;                   // read-back instructions per pass
  constexpr int NLD = NPASS * NRB;                // ... per MFMA tile (fp32: 4)
  const int rr = lane / LPR, cc = lane % LPR;
  const unsigned out_bytes = (unsigned)g.M * (unsigned)g.N * (F32OUT ? 4u : 2u);
  auto epilogue = [&](int row0, int n0, bool half) {
    if constexpr (DBG & 8) {
      float sres = 0.f;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NHW; ++b)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) { sres += acc[a][b][m][e]; acc[a][b][m][e] = 0.f; }
      if (g.C && sres == 12345.678f) g.C[threadIdx.x] = sres;
      return;
    }
    // MFMA tiles in the order (ha, hw, mt): a half item ends after the first 2 NHW
    constexpr int NT = 4 * NHW;
    const int nt = half ? NT / 2 : NT;
    auto tile_rc = [&](int j, int& mbase, int& nbase, int& hw, int& ha, int& mt) {
      ha = j / (2 * NHW); hw = (j / 2) % NHW; mt = j & 1;
      mbase = row0 + ha * 128 + wr * 64 + mt * 32;
      nbase = n0 + hw * 128 + wc * 32;
    };
    // bias of this lane's columns, per (hw, pass)
    f32x4 bias_lo[NHW][NPASS], bias_hi[NHW][NPASS];
#pragma unroll
    for (int hw = 0; hw < NHW; ++hw)
#pragma unroll
      for (int q = 0; q < NPASS; ++q) {
        const int n = n0 + hw * 128 + wc * 32 + q * CW + (F32OUT ? 4 : 8) * cc;
        bias_lo[hw][q] = *reinterpret_cast<const f32x4*>(g.bias + n);
        if constexpr (!F32OUT) bias_hi[hw][q] = *reinterpret_cast<const f32x4*>(g.bias + n + 4);
      }
    // residual rows, two MFMA tiles ahead of their use
    f32x4 rres[3][RES ? NLD : 1];
    auto prefetch = [&](int j, int slot) {
      if constexpr (RES && !(DBG & 16)) {
        int mbase, nbase, hw, ha, mt;
        tile_rc(j, mbase, nbase, hw, ha, mt);
#pragma unroll
        for (int q = 0; q < NPASS; ++q)
#pragma unroll
          for (int i = 0; i < NRB; ++i) {
            const unsigned off = ((unsigned)(mbase + i * RPW + rr) * (unsigned)g.N + (unsigned)(nbase + q * CW + 4 * cc)) * 4u;
            rres[slot][q * NRB + i] = p8_ld128(g.residual, out_bytes, off);
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
        int mbase, nbase, hw, ha, mt;
        tile_rc(j, mbase, nbase, hw, ha, mt);
        if constexpr (RES) {
          if (j + 2 < nt) prefetch(j + 2, (j + 2) % 3);
        }
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
#pragma unroll
          for (int gg = 0; gg < CW / 8; ++gg) {
            const int gi = q * (CW / 8) + gg;
            const int phys = (2 * gg + h) ^ (r & (CPRW - 1));
            f32x4 v = {acc[ha][hw][mt][4 * gi], acc[ha][hw][mt][4 * gi + 1], acc[ha][hw][mt][4 * gi + 2], acc[ha][hw][mt][4 * gi + 3]};
            *reinterpret_cast<f32x4*>(scr + r * (CW * 4) + phys * 16) = v;
          }
#pragma unroll
          for (int i = 0; i < NRB; ++i) {
            const int row = i * RPW + rr;
            const int m = mbase + row;
            if constexpr (F32OUT) {
              f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * (CW * 4) + ((cc ^ (row & (CPRW - 1))) << 4));
              v += bias_lo[hw][q];
              if constexpr (RES && !(DBG & 16)) v += rres[j % 3][q * NRB + i];
              const unsigned off = ((unsigned)m * (unsigned)g.N + (unsigned)(nbase + q * CW + 4 * cc)) * 4u;
              if constexpr (DBG & 16) {
                if (v[0] == 12345.678f) p8_st128(g.C, out_bytes, off, __builtin_bit_cast(u32x4, v));
              } else {
                p8_st128(g.C, out_bytes, off, __builtin_bit_cast(u32x4, v));
              }
            } else {
              f32x4 v0 = *reinterpret_cast<const f32x4*>(scr + row * (CW * 4) + (((2 * cc) ^ (row & (CPRW - 1))) << 4));
              f32x4 v1 = *reinterpret_cast<const f32x4*>(scr + row * (CW * 4) + (((2 * cc + 1) ^ (row & (CPRW - 1))) << 4));
              v0 += bias_lo[hw][q];
              v1 += bias_hi[hw][q];
              float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
              if constexpr (ACT) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = P == 1 ? gelu_bf16_f(v[e]) : gelu_fast_f(v[e]);
              }
              const unsigned off = ((unsigned)m * (unsigned)g.N + (unsigned)(nbase + q * CW + 8 * cc)) * 2u;
              bf16x8 q0, q1, q2;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const __bf16 p0 = (__bf16)v[e];
                q0[e] = p0;
                if constexpr (PO > 1) {
                  const float r1 = v[e] - (float)p0;
                  const __bf16 p1 = (__bf16)r1;
                  q1[e] = p1;
                  q2[e] = (__bf16)(r1 - (float)p1);
                }
              }
              const bool go = (DBG & 16) ? v[0] == 12345.678f : true;
              if (go) {
                p8_st128(g.Cp, out_bytes, off, __builtin_bit_cast(u32x4, q0));
                if constexpr (PO > 1) {
                  p8_st128(g.Cp + g.c_stride, out_bytes, off, __builtin_bit_cast(u32x4, q1));
                  p8_st128(g.Cp + 2 * g.c_stride, out_bytes, off, __builtin_bit_cast(u32x4, q2));
                }
              }
            }
          }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ha][hw][mt][e] = 0.f;
      }
    }
    // The stores are NOT waited for.  They share the vmcnt queue with the LDS-DMA, in issue order: a wait of the next item's
    // phases that must retire chunk c allows (its window of younger chunks) + (the stores issued between them) outstanding -
    // for the L phases whose window still reaches back across this epilogue.  The count used, S_FULL / S_HALF, is at most the
    // number of stores really issued (and there are loads on top): allowing FEWER than really sit behind c only waits for a few
    // of the oldest epilogue operations as well, which finished long ago.
    post_epi = L;
    post_half = half;
  };

  // ---- K-split item (P = 1): this wave's partial (128 values per lane, 32 KB) -> workspace; the wave arriving last at the (tile, wave)
  // counter sums the slices' partials in slice order - its own read back too: the sum does not depend on who finishes - one 128-row
  // half at a time (acc[0] the running sum, acc[1] the landing buffer: the accumulators are all the registers there are) and runs the
  // epilogue of that half as a half item.  Write-through stores / loads around the L2 (the slices may sit on different XCDs), stores
  // complete (vmcnt 0) before the agent-scope increment.
  auto slice_finish = [&](int row0, int n0) {
    if constexpr (P == 1) {
      const int j = cu / g.ks_S, sl = cu - j * g.ks_S;
      float* wbase = g.ks_ws + ((size_t)j * g.ks_S * 8 + wave) * (128 * 64);
      const unsigned sstride = 8u * 128u * 64u * 4u, wbytes = (unsigned)g.ks_S * sstride;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NHW; ++b)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 v = {acc[a][b][m][4 * q], acc[a][b][m][4 * q + 1], acc[a][b][m][4 * q + 2], acc[a][b][m][4 * q + 3]};
              p8_st128_sys(wbase, wbytes, (unsigned)sl * sstride + (unsigned)(((((a * NHW + b) * 2 + m) * 4 + q) * 1024) + lane * 16), __builtin_bit_cast(u32x4, v));
            }
      p8_wait_vmcnt<0>();
      int old = 0;
      if (lane == 0) old = __hip_atomic_fetch_add(g.ks_cnt + j * 8 + wave, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      old = __builtin_amdgcn_readfirstlane(old);
      if (old != g.ks_S - 1) return;
      if (lane == 0) __hip_atomic_store(g.ks_cnt + j * 8 + wave, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll 1
      for (int hh = 0; hh < 2; ++hh) {
        for (int t = 0; t < g.ks_S; ++t) {
#pragma unroll
          for (int b = 0; b < NHW; ++b)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const f32x4 v = p8_ld128_sys(wbase, wbytes, (unsigned)t * sstride + (unsigned)(((((hh * NHW + b) * 2 + m) * 4 + q) * 1024) + lane * 16));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[1][b][m][4 * q + e] = v[e];
              }
#pragma unroll
          for (int b = 0; b < NHW; ++b)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int e = 0; e < 16; ++e) acc[0][b][m][e] = t == 0 ? acc[1][b][m][e] : acc[0][b][m][e] + acc[1][b][m][e];
        }
        epilogue(row0 + hh * 128, n0, true);
      }
    }
  };

  // ---- prologue: chunks 0 .. D - 1 of the stream
  {
    auto pro = [&](auto c_c) {
      constexpr int CIDX = decltype(c_c)::value;
      if constexpr (CIDX > 0 && (CIDX & 3) == 0) cursor_next_ktile();
      issue(std::integral_constant<int, (CIDX & 3)>{}, std::integral_constant<int, ((CIDX >> 2) & 1)>{});
    };
    pro(std::integral_constant<int, 0>{}); pro(std::integral_constant<int, 1>{}); pro(std::integral_constant<int, 2>{});
    pro(std::integral_constant<int, 3>{}); pro(std::integral_constant<int, 4>{});
    if constexpr (D > 5) pro(std::integral_constant<int, 5>{});
    static_assert(D == 5 || D == 6, "prologue issues chunks 0 .. D - 1");
    constexpr int PRO_EXTRA = P8_SHIFT0 ? GCH : 0;
    if constexpr (P8_SHIFT0) {   // ... and chunk D, which phase 0 no longer issues
      if constexpr ((D & 3) == 0) cursor_next_ktile();
      issue(std::integral_constant<int, (D & 3)>{}, std::integral_constant<int, ((D >> 2) & 1)>{});
    }
    half_guard = d_half ? L + 1 : 0;
    // "phase -1": chunks D - L .. D - 1 (.. D) may stay in flight
    if (DBG & 2) p8_wait_vmcnt<0>();
    else if (d_half) p8_wait_vmcnt<p8_window<P>(3, GCH, true) + PRO_EXTRA>();
    else p8_wait_vmcnt<p8_window<P>(3, GCH, false) + PRO_EXTRA>();
    __builtin_amdgcn_s_barrier();
  }

  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  for (int it = 0; it < n_items; ++it) {
    int row0, n0, c_kt0, c_kend;
    item(it, row0, n0, c_half, c_kt0, c_kend);
    if (grp1) __builtin_amdgcn_s_barrier();   // the second group runs one barrier interval behind
    for (int kk = c_kt0; kk < c_kend; kk += 2) {
      // steady for these 8 phases?  The cursor advances two K-tiles in them: it must stay in whole tiles and short of the end.
      steady = post_epi == 0 && half_guard == 0 && !c_half && !d_half && !d_done;
      if (steady && d_kt + 2 >= d_kend) {   // it crosses into the next item
        bool nhalf = false;
        if (d_item + 1 < n_items) { int r0_, n0_, k0_, k1_; item(d_item + 1, r0_, n0_, nhalf, k0_, k1_); }
        steady = d_item + 1 < n_items && !nhalf;
      }
      phase(I0{}, I0{}); phase(I0{}, I1{}); phase(I0{}, I2{}); phase(I0{}, I3{});
      phase(I1{}, I0{}); phase(I1{}, I1{}); phase(I1{}, I2{}); phase(I1{}, I3{});
    }
    steady = false;
    if (!grp1) __builtin_amdgcn_s_barrier();  // realign: both groups run their epilogues at the same time
    if constexpr (P == 1) {
      if (has_slice && it == n_whole) {   // (the last item: nothing follows)
        slice_finish(row0, n0);
        continue;
      }
    }
    epilogue(row0, n0, c_half);
  }
#ifdef TT_P8_STAMP
  if (g.clock_print && lane == 0 && (wave == 0 || wave == 5) && blockIdx.x == 3 && st_n > 0)
    printf("p8 stamps: wave %d  %u steady phases, cycles per phase: DMA issue %.0f | fragment reads issued %.0f | counted wait %.0f | barrier 1 %.0f | "
           "lgkmcnt(0) + MFMAs issued %.0f | barrier 2 (to the next phase's start) %.0f\n", wave, st_n, (double)st_issue / st_n, (double)st_reads / st_n,
           (double)st_wait / st_n, (double)st_bar1 / st_n, (double)st_mfma / st_n, (double)st_bar2 / st_n);
#endif
#ifdef TT_P8_CLOCK
  if (g.clock_print && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 101)) {
    const unsigned long long dt = __builtin_amdgcn_s_memtime() - clk_t0, dr = __builtin_amdgcn_s_memrealtime() - clk_r0;
    printf("p8 clock: block %d  %llu cycles in %llu x 10 ns = %.3f GHz\n", (int)blockIdx.x, dt, dr, (double)dt / (double)dr * 0.1);
  }
#endif
}

static int p8_order_mode() { return tuning_knob(KNOB_P8_ORDER); }   // (tools A/B the orders in one process through tt_set_tuning_knob)

template <int P, int EPI, int DBG = 0>
static int launch_planes8(const P8Args& g, hipStream_t s) {
  hipLaunchKernelGGL((gemm_planes8_kernel<P, EPI, DBG>), dim3(g.ncu), dim3(512), 0, s, g);
  TT_CHECK_LAUNCH("gemm_planes8");
  return TT_OK;
}

// Shape / epilogue eligibility and the work decomposition.  Returns the epilogue kind or -1.
static int planes8_plan(long long x_plane_stride, long long w_plane_stride, int planes, bool has_bias, bool has_residual, bool has_y, int y_nplanes,
                        int M, int N, int K, int act, int* ntn_out, long long* ntiles_out, int* ncu_out, int* n_full_out, int* n_half_out,
                        int* ks_S_out = nullptr) {
  if (planes != 1 && planes != 3) return -1;
  const int BN = planes == 1 ? 256 : 128, BK = planes == 1 ? 64 : 32;
  if (N % BN != 0 || K % (2 * BK) != 0 || M < 256 || !has_bias) return -1;
  int epi = -1;
  if (has_y && !y_nplanes && !act) epi = has_residual ? P8_F32_RES : P8_F32;
  else if (!has_y && !has_residual && planes == 1 && y_nplanes == 1) epi = act ? P8_BF16_GELU : P8_BF16;
  else if (!has_y && !has_residual && planes == 3 && y_nplanes == 3 && act) epi = P8_PL3_GELU;
  if (epi < 0) return -1;
  // 32-bit buffer offsets
  if ((long long)(planes - 1) * x_plane_stride * 2 + (long long)M * K * 2 >= 0x7fffffffLL) return -1;
  if ((long long)(planes - 1) * w_plane_stride * 2 + (long long)N * K * 2 >= 0x7fffffffLL) return -1;
  if ((long long)M * N * 4 >= 0x7fffffffLL) return -1;
  const int ntm = (M + 255) / 256, ntn = N / BN;
  const long long ntiles = (long long)ntm * ntn;
  const int ncu_dev = device_cu_count();   // (of the CURRENT device)
  if (ntiles < ncu_dev / 2) return -1;   // a persistent grid that cannot fill the chip: the small-tile kernel does better
  // decomposition: R whole rounds of tiles over the CUs; the r tiles left over are cut into 2 r half tiles (one per workgroup)
  // when that is a shorter tail than another whole round (2 r <= CUs), else all tiles are dealt round-robin
  int ncu = (int)(ntiles < ncu_dev ? ntiles : ncu_dev), n_full = 0, n_half = 0;
  const long long R = ntiles / ncu_dev, rem = ntiles - R * ncu_dev;
  const bool no_half = tuning_knob(KNOB_P8_NO_HALF) != 0;   // tuning aid
  // K-split of the left-over tiles (P = 1, behind whole rounds; gemm_pairs8.hip's rule in this kernel's microseconds: ~ 2 us per 64-deep
  // K-tile of a 256 x 256 tile, the exchange ~ 12 + 4 per slice - two halves of S load rounds each)
  int ks_S = 0;
  if (ks_S_out && planes == 1 && tuning_knob(KNOB_Q8_KSPLIT) % 10 != 0 && R > 0 && rem > 0) {
    const int U = K / (2 * BK), nk = K / BK;
    int S = (int)(ncu_dev / rem);
    if (S > U) S = U;
    if (S > 6) S = 6;
    if (S >= 2) {
      const double t_tile = 2.0 * nk + 5.0;
      const double t_slice = 2.0 * 2 * ((U + S - 1) / S) + 5.0 + 12.0 + 4.0 * S;
      const double t_else = (!no_half && 2 * rem <= ncu_dev) ? 0.86 * t_tile : t_tile;
      if (t_slice < 0.9 * t_else) ks_S = S;
    }
  }
  if (ks_S >= 2) {
    ncu = ncu_dev;
    n_full = (int)R;
    n_half = 0;
  } else if (!no_half && rem > 0 && 2 * rem <= ncu_dev) {   // (round 5: this branch had been dropped when the K-split went in - every P = 3
    ncu = ncu_dev;                                          // shape and every P = 1 shape without a K-split ran its tail as whole tiles)
    n_full = (int)R;
    n_half = (int)(2 * rem);
  }
  if (ks_S_out) *ks_S_out = ks_S;
  *ntn_out = ntn; *ntiles_out = ntiles; *ncu_out = ncu; *n_full_out = n_full; *n_half_out = n_half;
  return epi;
}

// Would tt_linear_fwd_planes with these arguments run gemm_planes8_kernel?  (profilers' labels: tt_linear_fwd_planes_route)
int planes8_would_run(int planes, int M, int N, int K, int act, int has_bias, int has_residual, int has_y, int y_nplanes) {
  int ntn, ncu, n_full, n_half;
  long long ntiles;
  const long long xs = (long long)M * K, ws = (long long)N * K;
  return planes8_plan(xs, ws, planes, has_bias != 0, has_residual != 0, has_y != 0, y_nplanes, M, N, K, act, &ntn, &ntiles, &ncu, &n_full, &n_half) >= 0;
}

// Called by linear_planes_impl (gemm_planes.hip).  Returns TT_OK after a launch, 1 when the shape / epilogue is not this kernel's
// (the caller then takes gemm_planes_kernel), < 0 on a launch error.
int planes8_try(const void* x_planes, long long x_plane_stride, const void* w_planes, long long w_plane_stride, int planes, const float* bias,
                const float* residual, float* y, void* y_planes, long long y_plane_stride, int y_nplanes, int M, int N, int K, int act,
                void* ksplit_ws, size_t ksplit_ws_bytes_, hipStream_t s) {
  int ntn, ncu, n_full, n_half, ks_S = 0;
  // the K-split needs the caller's workspace (tt_linear_ksplit_workspace_bytes / _init); without one the left-over tiles are cut into halves
  KsplitWs kw{nullptr, nullptr};
  const bool have_ws = ksplit_ws_carve(ksplit_ws, ksplit_ws_bytes_, &kw);
  long long ntiles;
  if ((y != nullptr) == (y_planes != nullptr) && y) {}   // (y together with planes is not an epilogue of this kernel: the plan rejects it below)
  const int epi = (y && y_planes) ? -1
                                  : planes8_plan(x_plane_stride, w_plane_stride, planes, bias != nullptr, residual != nullptr, y != nullptr,
                                                 y_planes ? y_nplanes : 0, M, N, K, act, &ntn, &ntiles, &ncu, &n_full, &n_half, have_ws ? &ks_S : nullptr);
  if (epi < 0) return 1;
  float* ks_ws = ks_S >= 2 ? kw.partials : nullptr;
  int* ks_cnt = ks_S >= 2 ? kw.counters : nullptr;
  P8Args g{static_cast<const __bf16*>(x_planes), static_cast<const __bf16*>(w_planes), x_plane_stride, w_plane_stride, M, N, K, bias, residual, y,
           static_cast<__bf16*>(y_planes), y_plane_stride, ntn, (int)ntiles, ncu, n_full, n_half, tuning_knob(KNOB_P8_CLOCK_PRINT), p8_order_mode(),
           ks_S, ks_S >= 2 ? (int)(ntiles - (long long)n_full * ncu) : 0, ks_ws, ks_cnt};
#ifdef TT_P8_ABLATE   // timing-study build only (tools/build_variant.sh -DTT_P8_ABLATE): TT_P8_DBG selects a crippled instantiation
  {
    const char* e = getenv("TT_P8_DBG");
    const int dbg = e ? atoi(e) : 0;
#define P8_DBG_CASE(PV, EV)                                       \
  if (planes == PV && epi == EV) {                                \
    if (dbg == 1) return launch_planes8<PV, EV, 1>(g, s);         \
    if (dbg == 2) return launch_planes8<PV, EV, 2>(g, s);         \
    if (dbg == 4) return launch_planes8<PV, EV, 4>(g, s);         \
    if (dbg == 8) return launch_planes8<PV, EV, 8>(g, s);         \
    if (dbg == 16) return launch_planes8<PV, EV, 16>(g, s);       \
    if (dbg == 11) return launch_planes8<PV, EV, 11>(g, s);       \
    if (dbg == 13) return launch_planes8<PV, EV, 13>(g, s);       \
    if (dbg == 14) return launch_planes8<PV, EV, 14>(g, s);       \
    if (dbg == 15) return launch_planes8<PV, EV, 15>(g, s);       \
    if (dbg == 9) return launch_planes8<PV, EV, 9>(g, s);         \
    if (dbg == 64) return launch_planes8<PV, EV, 64>(g, s);       \
    if (dbg == 72) return launch_planes8<PV, EV, 72>(g, s);       \
    if (dbg == 10) return launch_planes8<PV, EV, 10>(g, s);       \
    if (dbg == 12) return launch_planes8<PV, EV, 12>(g, s);       \
    if (dbg == 32) return launch_planes8<PV, EV, 32>(g, s);       \
    if (dbg == 40) return launch_planes8<PV, EV, 40>(g, s);       \
    if (dbg == 41) return launch_planes8<PV, EV, 41>(g, s);       \
  }
    P8_DBG_CASE(1, P8_BF16) P8_DBG_CASE(1, P8_BF16_GELU) P8_DBG_CASE(1, P8_F32_RES)
    P8_DBG_CASE(3, P8_F32) P8_DBG_CASE(3, P8_F32_RES) P8_DBG_CASE(3, P8_PL3_GELU)
#undef P8_DBG_CASE
  }
#endif
  if (planes == 1) {
    switch (epi) {
      case P8_BF16: return launch_planes8<1, P8_BF16>(g, s);
      case P8_BF16_GELU: return launch_planes8<1, P8_BF16_GELU>(g, s);
      case P8_F32: return launch_planes8<1, P8_F32>(g, s);
      case P8_F32_RES: return launch_planes8<1, P8_F32_RES>(g, s);
      default: return 1;
    }
  }
  switch (epi) {
    case P8_F32: return launch_planes8<3, P8_F32>(g, s);
    case P8_F32_RES: return launch_planes8<3, P8_F32_RES>(g, s);
    case P8_PL3_GELU: return launch_planes8<3, P8_PL3_GELU>(g, s);
    default: return 1;
  }
}

}  // namespace tt
