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
//   * PERSISTENT: a workgroup walks a contiguous run of tiles; the DMA cursor runs D chunks ahead of the compute cursor ACROSS
//     tile boundaries, so a tile's first K-tiles land while the previous tile finishes - no per-tile prologue bubble.
//   * swapped operands: D = W_tile . X_tile^T puts an output ROW on a lane and 4 consecutive columns in 4 registers; the epilogue
//     stages one 32 x 32 MFMA tile at a time through the wave's private scratch (no workgroup barrier: both groups run their
//     epilogues concurrently) and leaves as full 128-byte row segments with bias / GELU / residual applied on the way.
//
// LDS images and the fragment / DMA lane maps are those of gemm_planes.hip (rows of BK bf16, 16-byte chunks XOR-swizzled through
// the per-lane SOURCE address; conflict-free ds_read_b128 for v_mfma_f32_32x32x16_bf16).
#include "common.hpp"

namespace tt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct P8Args {
  const __bf16* X;   // [P][M][K]
  const __bf16* W;   // [P][N][K]
  long long x_stride, w_stride;   // plane strides in elements
  int M, N, K;
  const float* bias;      // [N] or null
  const float* residual;  // [M][N] or null (may alias C)
  float* C;               // [M][N] fp32 or null
  __bf16* Cp;             // [po][M][N] bf16 planes or null
  long long c_stride;
  int po;                 // output planes 0..3
  int act;                // 1 = GELU
  int ntn, ntiles, ncu;   // column tiles, tiles, workgroups launched
};


__device__ __forceinline__ void p8_dma16(const void* base, unsigned char* lds_dst, int voffset, int soffset) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds_dst, 16, voffset, soffset, 0, 0);
}

template <int N>
__device__ __forceinline__ void p8_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- the schedule, per P.  A K-tile is NCH = 4 chunk slots issued one per phase; chunk i of K-tile t has stream index 4 t + i.
template <int P>
struct P8Cfg;
template <>
struct P8Cfg<1> {
  static constexpr int BK = 64, NHW = 2, D = 6, L = 4;
  // chunks in need order: W0, X0, W1, X1
  static constexpr bool exists(int i) { return true; }
  static constexpr bool is_x(int i) { return i & 1; }
  static constexpr int half(int i) { return i >> 1; }
  static constexpr int need(int i) { return i == 0 ? 0 : i - 1; }        // first phase (0..3) that reads the chunk
  static constexpr int last_read(int i) { return i == 0 ? 0 : i - 1; }   // W0 / W1 fragments stay in registers for the K-tile
};
template <>
struct P8Cfg<3> {
  static constexpr int BK = 32, NHW = 1, D = 5, L = 3;
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
// gch = wave-instructions per wave and chunk
template <int P>
constexpr int p8_window(int ph, int gch) {
  using C = P8Cfg<P>;
  int n = 0;
  for (int k = 0; k < C::L; ++k) n += C::exists((ph + C::D - k) & 3) ? gch : 0;
  return n;
}
static_assert(p8_schedule_ok<1>() && p8_schedule_ok<3>(), "LDS-DMA schedule violates a RAW / WAR rule");

template <int P>
__global__ __launch_bounds__(512) void gemm_planes8_kernel(P8Args g) {
  using CF = P8Cfg<P>;
  constexpr int BK = CF::BK, NHW = CF::NHW, D = CF::D;
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
  constexpr int BN = 128 * NHW;
  constexpr int NF = P == 1 ? NKS : P;               // fragments per register set
  static_assert(RING_B + 8 * SCR_B <= 160 * 1024 && (CW == 32 || CW == 16), "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char smem[160 * 1024];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool grp1 = wave >= 4;
  const int wr = wave >> 2, wc = wave & 3;
  const int r = lane & 31, h = lane >> 5;
  const int K2 = g.K * 2, nk = g.K / BK;

  // ---- this workgroup's tiles: workgroups that share an XCD (equal id mod 8) get adjacent runs of the row-major tile order,
  // so an activation row block is fetched into one L2 while the weight strips stay hot in all of them
  int cu = blockIdx.x;
  if ((g.ncu & 7) == 0) cu = (blockIdx.x & 7) * (g.ncu >> 3) + (blockIdx.x >> 3);
  const int t_begin = (int)(((long long)cu * g.ntiles) / g.ncu), t_end = (int)(((long long)(cu + 1) * g.ntiles) / g.ncu);
  if (t_begin >= t_end) return;   // whole workgroup

  // ---- LDS-DMA lane map (see gemm_planes.hip): lane -> (row, slot) of the 1 KiB piece, source chunk = slot ^ f(row)
  const int l_row = lane / CPR, l_slot = lane % CPR;
  const int d_row0 = wave * RPI + l_row;                                  // image row of piece j = wave; j = wave + 8 i adds 8 i RPI
  const int d_chunk = l_slot ^ ((d_row0 / WIN) & (CPR - 1));              // (8 RPI rows further: same f)
  const int w_voff = d_row0 * K2 + d_chunk * 16;
  const int xps = (int)(g.x_stride * 2), wps = (int)(g.w_stride * 2);     // plane strides in bytes

  // DMA cursor (scalar state + the X voffsets of its tile; rows beyond M are clamped to M - 1 and masked at the store)
  int d_tile = t_begin, d_kt = 0, d_mb = t_begin / g.ntn, d_ns = t_begin % g.ntn;
  bool d_done = false;
  int d_kofs = 0;       // d_kt * ROWB
  int d_wbase = 0;      // first W row of the tile * K2
  int x_voff[2][JPW];
  auto cursor_tile = [&]() {
    d_wbase = d_ns * BN * K2;
#pragma unroll
    for (int ha = 0; ha < 2; ++ha)
#pragma unroll
      for (int i = 0; i < JPW; ++i) {
        int row = d_mb * 256 + ha * 128 + 8 * i * RPI + d_row0;
        row = row < g.M ? row : g.M - 1;
        x_voff[ha][i] = row * K2 + d_chunk * 16;
      }
  };
  cursor_tile();
  auto cursor_next_ktile = [&]() {
    ++d_kt;
    d_kofs += ROWB;
    if (d_kt == nk) {
      d_kt = 0;
      d_kofs = 0;
      ++d_tile;
      if (d_tile >= t_end) {
        d_done = true;
      } else {
        if (++d_ns == g.ntn) { d_ns = 0; ++d_mb; }
        cursor_tile();
      }
    }
  };
  // issue chunk IDX of the cursor's K-tile into ring buffer B
  auto issue = [&](auto idx_c, auto buf_c) {
    constexpr int IDX = decltype(idx_c)::value, B = decltype(buf_c)::value;
    if constexpr (CF::exists(IDX)) {
      if (d_done) return;
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

  auto ld = [&](int off) { return *reinterpret_cast<const bf16x8*>(smem + off); };

  // ---- one phase: [fragment reads | DMA issue | counted wait] barrier [MFMAs] barrier
  auto phase = [&](auto buf_c, auto ph_c) {
    constexpr int B = decltype(buf_c)::value, PH = decltype(ph_c)::value;
    constexpr int base = B * BUF_B;
    if constexpr (P == 1) {
      // chunks W0 X0 W1 X1 in slots 0..3; quadrants (hA, hW): (0,0) (0,1) (1,1) (1,0)
      if constexpr (PH == 0) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) Wf[0][ks] = ld(base + 0 * HALF_B + w_slice + lo[ks]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) Xf[mt][ks] = ld(base + 1 * HALF_B + x_slice + mt * 32 * ROWB + lo[ks]);
      } else if constexpr (PH == 1) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) Wf[1][ks] = ld(base + 2 * HALF_B + w_slice + lo[ks]);
      } else if constexpr (PH == 2) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) Xf[mt][ks] = ld(base + 3 * HALF_B + x_slice + mt * 32 * ROWB + lo[ks]);
      }
    } else {
      // chunks W X0 X1 in slots 0..2 (planes inside a chunk); phases (hA, ks): (0,0) (0,1) (1,1) (1,0)
      constexpr int HA = PH >> 1, KS = (PH == 1 || PH == 2) ? 1 : 0;
      if constexpr (PH < 2) {
#pragma unroll
        for (int p = 0; p < P; ++p) Wf[KS][p] = ld(base + 0 * HALF_B + p * PLANE_B + w_slice + lo[KS]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int p = 0; p < P; ++p) Xf[mt][p] = ld(base + (1 + HA) * HALF_B + p * PLANE_B + x_slice + mt * 32 * ROWB + lo[KS]);
    }
    // DMA: chunk (PH + D) of the stream = chunk (PH + D) & 3 of the K-tile (PH + D) / 4 further on
    constexpr int CI = (PH + D) & 3, BT = (B + (PH + D) / 4) & 1;
    if constexpr (CI == 0) cursor_next_ktile();
    issue(std::integral_constant<int, CI>{}, std::integral_constant<int, BT>{});
    if (d_done) p8_wait_vmcnt<0>();
    else p8_wait_vmcnt<p8_window<P>(PH, GCH)>();
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if constexpr (P == 1) {
      constexpr int HA = PH >> 1, HW = (PH == 1 || PH == 2) ? 1 : 0;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[HA][HW][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[HW][ks], Xf[mt][ks], acc[HA][HW][mt], 0, 0, 0);
    } else {
      constexpr int HA = PH >> 1, KS = (PH == 1 || PH == 2) ? 1 : 0;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int s = P - 1; s >= 0; --s)          // plane-index sum: small terms first (as gemm_planes_kernel)
#pragma unroll
          for (int pa = 0; pa <= s; ++pa)
            acc[HA][0][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[KS][s - pa], Xf[mt][pa], acc[HA][0][mt], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  // ---- epilogue of one tile: per wave, through its private scratch, no workgroup barrier
  unsigned char* scr = smem + RING_B + wave * SCR_B;
  constexpr int CPRW = CW / 4;      // 16-byte chunks per staged row
  constexpr int RPW = 64 / CPRW;    // rows per read-back instruction
  const int rr = lane / CPRW, cc = lane % CPRW;
  auto epilogue = [&](int row0, int n0) {
#pragma unroll
    for (int hw = 0; hw < NHW; ++hw) {
      const int nbase = n0 + hw * 128 + wc * 32;
      f32x4 bias4[32 / CW];
#pragma unroll
      for (int q = 0; q < 32 / CW; ++q) {
        if (g.bias) bias4[q] = *reinterpret_cast<const f32x4*>(g.bias + nbase + q * CW + 4 * cc);
        else bias4[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int ha = 0; ha < 2; ++ha)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int mbase = row0 + ha * 128 + wr * 64 + mt * 32;
#pragma unroll
          for (int q = 0; q < 32 / CW; ++q) {
            // lane (m = r) holds columns 8 g + 4 h + {0..3} in registers 4 g .. 4 g + 3
#pragma unroll
            for (int gg = 0; gg < CW / 8; ++gg) {
              const int gi = q * (CW / 8) + gg;
              const int phys = (2 * gg + h) ^ (r & (CPRW - 1));
              f32x4 v = {acc[ha][hw][mt][4 * gi], acc[ha][hw][mt][4 * gi + 1], acc[ha][hw][mt][4 * gi + 2], acc[ha][hw][mt][4 * gi + 3]};
              *reinterpret_cast<f32x4*>(scr + r * (CW * 4) + phys * 16) = v;
            }
#pragma unroll
            for (int i = 0; i < 32 / RPW; ++i) {
              const int row = i * RPW + rr;
              const int phys = cc ^ (row & (CPRW - 1));
              f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * (CW * 4) + phys * 16);
              const int m = mbase + row;
              v += bias4[q];
              if (g.act == 1) {
                if constexpr (P == 1) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = gelu_bf16_f(v[e]);
                } else {
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = gelu_fast_f(v[e]);
                }
              }
              if (m < g.M) {
                const size_t off = (size_t)m * g.N + nbase + q * CW + 4 * cc;
                if (g.residual) v += *reinterpret_cast<const f32x4*>(g.residual + off);
                if (g.C) *reinterpret_cast<f32x4*>(g.C + off) = v;
                if (g.po > 0) {
                  bf16x4 q0, q1, q2;
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    const __bf16 p0 = (__bf16)v[e];
                    const float r1 = v[e] - (float)p0;
                    const __bf16 p1 = (__bf16)r1;
                    q0[e] = p0; q1[e] = p1; q2[e] = (__bf16)(r1 - (float)p1);
                  }
                  *reinterpret_cast<bf16x4*>(g.Cp + off) = q0;
                  if (g.po > 1) *reinterpret_cast<bf16x4*>(g.Cp + g.c_stride + off) = q1;
                  if (g.po > 2) *reinterpret_cast<bf16x4*>(g.Cp + 2 * g.c_stride + off) = q2;
                }
              }
            }
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[ha][hw][mt][e] = 0.f;
        }
    }
    // the stores share the vmcnt queue with the LDS-DMA in flight: retire them here so that the counted waits of the next tile's
    // phases see DMA pieces only (their count is what makes those waits correct)
    p8_wait_vmcnt<0>();
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
    p8_wait_vmcnt<p8_window<P>(3, GCH)>();   // "phase -1": chunks D - L .. D - 1 may stay in flight
    __builtin_amdgcn_s_barrier();
  }

  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  int mb = t_begin / g.ntn, ns = t_begin % g.ntn;
  for (int t = t_begin; t < t_end; ++t) {
    if (grp1) __builtin_amdgcn_s_barrier();   // the second group runs one barrier interval behind
    for (int kk = 0; kk < nk; kk += 2) {
      phase(I0{}, I0{}); phase(I0{}, I1{}); phase(I0{}, I2{}); phase(I0{}, I3{});
      phase(I1{}, I0{}); phase(I1{}, I1{}); phase(I1{}, I2{}); phase(I1{}, I3{});
    }
    if (!grp1) __builtin_amdgcn_s_barrier();  // realign: both groups run their epilogues at the same time
    epilogue(mb * 256, ns * BN);
    if (++ns == g.ntn) { ns = 0; ++mb; }
  }
}

template <int P>
static int launch_planes8(const P8Args& g, hipStream_t s) {
  hipLaunchKernelGGL((gemm_planes8_kernel<P>), dim3(g.ncu), dim3(512), 0, s, g);
  TT_CHECK_LAUNCH("gemm_planes8");
  return TT_OK;
}

// Called by linear_planes_impl (gemm_planes.hip).  Returns TT_OK after a launch, 1 when the shape / epilogue is not this kernel's
// (the caller then takes gemm_planes_kernel), < 0 on a launch error.
int planes8_try(const void* x_planes, long long x_plane_stride, const void* w_planes, long long w_plane_stride, int planes, const float* bias,
                const float* residual, float* y, void* y_planes, long long y_plane_stride, int y_nplanes, int M, int N, int K, int act,
                hipStream_t s) {
  if (planes != 1 && planes != 3) return 1;
  const int BN = planes == 1 ? 256 : 128, BK = planes == 1 ? 64 : 32;
  if (N % BN != 0 || K % (2 * BK) != 0 || M < 256) return 1;
  if ((long long)M * K * 2 >= 0x7fffffffLL || (long long)N * K * 2 >= 0x7fffffffLL) return 1;                      // 32-bit buffer offsets
  if ((long long)(planes - 1) * x_plane_stride * 2 + (long long)M * K * 2 >= 0x7fffffffLL) return 1;
  if ((long long)(planes - 1) * w_plane_stride * 2 + (long long)N * K * 2 >= 0x7fffffffLL) return 1;
  const int ntm = (M + 255) / 256, ntn = N / BN;
  const long long ntiles = (long long)ntm * ntn;
  static const int ncu_dev = [] {
    hipDeviceProp_t p;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
    return p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }();
  if (ntiles < (3 * ncu_dev) / 4) return 1;   // a persistent grid that cannot fill the chip: the small-tile kernel does better
  P8Args g{static_cast<const __bf16*>(x_planes), static_cast<const __bf16*>(w_planes), x_plane_stride, w_plane_stride, M, N, K, bias, residual, y,
           static_cast<__bf16*>(y_planes), y_plane_stride, y_nplanes, act, ntn, (int)ntiles, (int)(ntiles < ncu_dev ? ntiles : ncu_dev)};
  return planes == 1 ? launch_planes8<1>(g, s) : launch_planes8<3>(g, s);
}

}  // namespace tt
