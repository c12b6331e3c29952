// Fused attention forward on fp16-PAIR operands (dino_vision_transformer.py:120-132) - the attention of the fp32-accurate split mode
// "f16x3" (round 4): qkv [F N][2 x 3 H 64] in pairs as the qkv Linear's epilogue left it (common.hpp split_pair: groups of 32 columns as
// [hi x 32][lo x 32] fp16, a head's 64 dims = 256 contiguous bytes) -> the attention output in pairs [F N][2 H 64] (the proj Linear's
// operand) and / or in fp32 (+ the log-sum-exp rows the backward recomputes from).  head_dim 64; N <= 256: the kernel below with K / V of a
// head resident in LDS; any N: the KV-tiled kernel at the end of this file.
//
// Both matrix products take three v_mfma_f32_32x32x16_f16 per term, as the pair GEMMs do (gemm_pairs8.hip): S = K Q^T as kh qh into one
// accumulator and kh ql + kl qh into a second one, folded with the exact 2^-11 per key tile; the probabilities p = 2^(s c - m c) in (0, 1] are
// split into (hi, lo) on their way into O^T = V^T P^T, which runs the same way.  fp32 scores, softmax statistics and accumulation.
//
// Structure: that of attention_bf16.hip - an 8-wave workgroup works on one (frame, head) at a time (one workgroup per CU, looping over its
// items: round 5, below); K and V of that head go HBM -> LDS once, by LDS-DMA, and stay:
//   K image  [key][256 B]: 16-byte chunks XOR-swizzled by key & 15 -> conflict-free ds_read_b128 A fragments (rows of 256 B are whole bank rows)
//   V image  [key][256 B]: the four 64-byte quarters (hi / lo of dims 0-31, 32-63) XOR-swizzled by key & 3 -> conflict-free
//            ds_read_b64_tr_b16: V stays row-major (coalesced DMA) and is consumed transposed, as V^T fragments.
// A wave owns one 32-query tile (7 of the 8 waves work at 197 tokens).  S^T = K Q^T puts a query's score row in one lane column (2 lanes):
// register-local softmax plus one cross-half exchange, and P^T is the B operand of the second product as it stands (the accumulator-as-operand
// map of the 32 x 32 C layout: element j of lane half h of k-step s is key 16 s + 8 (j >> 2) + 4 h + (j & 3)); the V^T fragments are gathered
// in that key order by transposed LDS reads.  Keys >= N are clamped on load and masked to -inf.
#include "common.hpp"

namespace tt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#ifndef TT_AP_PACKED
#define TT_AP_PACKED 1   // the probabilities' exp / split in packed fp32 instructions (0: the scalar form of round 5, for A/B builds)
#endif
typedef __attribute__((address_space(3))) unsigned char lds_u8;
#ifndef TT_APF_LOADERS
#define TT_APF_LOADERS 1   // KV-tiled kernel: the waves without a query tile issue the stage's LDS-DMA (0: every wave its share; A/B builds)
#endif
// (p - hi) 2^11 for two probabilities as fma(hi, -2^11, p 2^11), hi read AS fp16 by v_fma_mix_f32 (op_sel_hi = 1: an fp16 source, op_sel:
// which half of its register) - no conversion back to fp32.  Exact, like the subtraction and the scaling it replaces: p 2^11 and hi 2^11
// are exact and their difference has at most 13 significant bits.  (The compiler does not select the instruction by itself here.)
__device__ __forceinline__ f32x2 pair_residual_scaled(f32x2 p_scaled, f16x2 hi2) {
  const unsigned hb = __builtin_bit_cast(unsigned, hi2);
  const float k = -kPairScale;
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hb), "s"(k), "v"(p_scaled.x));
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hb), "s"(k), "v"(p_scaled.y));
  return (f32x2){r0, r1};
}

// TT_AP_DBG (timing studies only, tools/ap_ablate.py; 0 in the shipped build), a bit mask: 1 no score MFMAs, 2 no exp / split of the
// probabilities, 4 no P V MFMAs, 8 no V fragment reads, 16 no K fragment reads, 32 no K / V DMA
#ifndef TT_AP_DBG
#define TT_AP_DBG 0
#endif
#ifndef TT_AP_VPOS
#define TT_AP_VPOS 2   // the V image goes out behind this key tile of the score product (0: at the top of the item, in front of the Q loads)
#endif
template <int NKT>
__global__ __launch_bounds__(512) void attention_fwd_pairs_kernel(const _Float16* __restrict__ qkv, _Float16* __restrict__ out_pairs,
                                                                  float* __restrict__ out_f32, float* __restrict__ lse, int N, int H, float scale,
                                                                  int FH) {
  constexpr int KROWS = NKT * 32;
  constexpr int NPC = KROWS / 32;   // DMA pieces (4 keys x 256 B) per wave and image
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * KROWS * 256 + 8 * 4096];
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + KROWS * 256;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* Os = smem + 2 * KROWS * 256 + wave * 4096;   // this wave's output staging: 32 queries x 128 B
  const int Dm = H * 64;
  const long long RS = 6ll * Dm;                                // fp16 elements per qkv row (2 x 3 D)
  const int nqt = (N + 31) / 32;   // <= 8: one query tile per wave
  const int qt = wave;
  int lane_o = tid & 63;
#ifdef TT_AP_STAMP   // diagnostic build only (tools/ap_stamp.py): s_memtime stamps at the phase boundaries of every item, written over out_f32
  long long stamp[7];
  long long* stamp_out = reinterpret_cast<long long*>(out_f32);
  out_f32 = nullptr;
  int stamp_it = 0;
#define AP_STAMP(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[i])::"memory")
#else
#define AP_STAMP(i)
#endif

  // K or V image of one (frame, head): 4 keys x 256 B per DMA piece (56 or 64 pieces); lane -> (key, slot of 16), source chunk = slot ^
  // swizzle(key), one (clamped: no lane is out of range) offset register per piece under a buffer descriptor of the image.  The pieces
  // w, w + 8, ... are "wave w's"; the swizzle terms are those of the first of them (a piece step is 32 keys).
  auto issue_pieces = [&](int fh, int lane, bool v_image, int w) {
    if (TT_AP_DBG & 32) return;
    const int f = fh / H, hd = fh - f * H;
    const char* src0 = reinterpret_cast<const char*>(qkv + (long long)f * N * RS + hd * 128 + (v_image ? 4 : 2) * Dm);   // (uniform)
    const int l_row = lane >> 4, l_slot = lane & 15;
    const int key0 = w * 4 + l_row;
    const int ch = v_image ? (l_slot ^ ((key0 & 3) << 2)) : (l_slot ^ (key0 & 15));
    unsigned char* dst = (v_image ? Vs : Ks) + w * 1024;
    const unsigned rs2 = (unsigned)RS * 2u;   // bytes per qkv row (an image spans < 2^31 bytes: N <= 256 rows)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src0), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int key = key0 + 32 * j;
      const unsigned off = (unsigned)(key < N ? key : N - 1) * rs2 + (unsigned)ch * 16u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (void __attribute__((address_space(3)))*)(dst + j * 8192), 16, off, 0, 0, 0);
    }
  };
  auto issue_image = [&](int fh, int lane, bool v_image) { issue_pieces(fh, lane, v_image, wave); };

  // Persistent over the (frame, head) items of this workgroup (round 5: 768 items on 256 CUs at ViT-S/16).  The K image is only read by the
  // score product and the V image only by P V, so the loads of one item run under the other product of its neighbour with no LDS beyond the
  // two images: V (item) is issued behind the second key tile of the scores (behind the Q loads in the memory queue - in front of them the
  // score product waited ~5 k cycles for Q) and lands under the rest of them, K (next item) is issued behind the scores and lands under
  // softmax + P V.  Two workgroup barriers per item.  With the probabilities made per key tile inside P V and the K fragments read one tile
  // ahead: 58.3 -> 49.0 us per ViT-S/16 layer of 128 frames, bits equal (profiles/r05_attention_pairs_persistent.txt).  Measured and
  // dropped there: the next item's Q loads under the end of P V (the 32 registers spill), a touch of the next Q tile's lines instead (no
  // change), the pieces spread over the key tiles of both products (slower: 53 us), the idle eighth wave issuing every piece (51 us).
  auto load_q = [&](int fh, int lane, f16x8* qh, f16x8* ql) {
    const int f = fh / H, hd = fh - f * H;
    const int query = qt * 32 + (lane & 31);
    const int qrow = query < N ? query : N - 1;
    const _Float16* p0 = qkv + ((long long)f * N + qrow) * RS + hd * 128 + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const _Float16* p = p0 + (ks >> 1) * 64 + (ks & 1) * 16;
      qh[ks] = *reinterpret_cast<const f16x8*>(p);
      ql[ks] = *reinterpret_cast<const f16x8*>(p + 32);
    }
  };
  int fh = blockIdx.x;
  issue_image(fh, lane_o, false);
  for (; fh < FH; fh += gridDim.x) {
    asm volatile("" : "+v"(lane_o));   // the per-lane addresses below are recomputed per item: hoisted out of the loop they spill
    const int lane = lane_o;
    const int r = lane & 31, h = lane >> 5;
    const int f = fh / H, hd = fh - f * H;
    AP_STAMP(0);
    // K (item) has landed (issued one product ago) and every wave is past the V reads of the previous item
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    AP_STAMP(1);
    if (TT_AP_VPOS == 0 || qt >= nqt) issue_image(fh, lane, true);

    const int query = qt * 32 + r;
    f32x16 sacc[NKT];
    if (qt < nqt) {
    // Q fragments (B operand): lane (query r, half h) holds Q[query][16 ks + 8 h + j], hi and lo
    f16x8 qh[4], ql[4];
    load_q(fh, lane, qh, ql);

    // K fragment addresses: row kt * 32 + r -> a per-lane base (r) + an immediate (kt); the swizzle term depends on r & 15 only
    int kofs[8];   // [ks] hi, [4 + ks] lo
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int ch = (ks >> 1) * 8 + 2 * (ks & 1) + h;   // chunk of the hi half of this k-step; lo: + 4
      kofs[ks] = r * 256 + ((ch ^ (r & 15)) << 4);
      kofs[4 + ks] = r * 256 + (((ch + 4) ^ (r & 15)) << 4);
    }
    // the K fragments of key tile kt + 1 are read in front of the products of tile kt (two sets of eight)
    f16x8 kf[2][8];
    auto read_k = [&](int kt, f16x8* dst) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (TT_AP_DBG & 16) dst[i] = (i < 4 ? qh : ql)[(i + kt) & 3];
        else dst[i] = *reinterpret_cast<const f16x8*>(Ks + kt * 8192 + kofs[i]);
      }
    };
    read_k(0, kf[0]);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt + 1 < NKT) read_k(kt + 1, kf[(kt + 1) & 1]);
      f32x16 s1, s2;
#pragma unroll
      for (int e = 0; e < 16; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const f16x8 kfh = kf[kt & 1][ks], kfl = kf[kt & 1][4 + ks];
        if constexpr (TT_AP_DBG & 1) { s1[ks] += (float)kfh[0]; s2[ks] += (float)kfl[0]; }
        else {
          s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh, qh[ks], s1, 0, 0, 0);
          s2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh, ql[ks], s2, 0, 0, 0);
          s2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl, qh[ks], s2, 0, 0, 0);
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[kt][e] = fmaf(s2[e], kPairInvScale, s1[e]);
      __builtin_amdgcn_sched_barrier(0);   // one key tile at a time: the compiler otherwise hoists the later tiles' reads and spills
      if (TT_AP_VPOS != 0 && kt == TT_AP_VPOS - 1) {
        issue_image(fh, lane, true);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    }
    AP_STAMP(2);
    // V (item) has landed; every wave is past its K reads: the next item's K goes out
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    AP_STAMP(3);
    const bool more = fh + (int)gridDim.x < FH;
    if (more) issue_image(fh + gridDim.x, lane, false);
    if (qt < nqt) {
    // softmax over the keys of this lane column (attention_bf16.hip): p = 2^(s c - m c), c = scale log2(e)
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
      if (kt * 32 + 31 >= N) {   // (uniform) a key tile that reaches past N: masked to -inf before the max
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (key >= N) sacc[kt][e] = -INFINITY;
        }
      }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sacc[kt][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    AP_STAMP(4);
    const float c = scale * 1.44269504088896340736f;
    const float mc = mx * c;
    const int g16 = (lane >> 4) & 1, q4 = (lane >> 2) & 3, p4 = lane & 3;   // position inside the 16-lane transpose group
    // V^T gather addresses: block rows = keys key0 .. key0 + 3 (this lane supplies row q4), block columns = 16 dims of one 64-byte quarter
    // (hi or lo of a 32-dim group).  key0 = kt * 32 + 16 s + 4 h (+ 8) is a multiple of 4, so the quarter swizzle is q4 for every block:
    // four per-lane bases (quarter) + immediates (kt, s, the + 8 keys).
    // (round 6: the V image's base is part of the per-lane register and hidden from constant folding - the image sits 56 KB into the LDS
    // array, and with its base folded into the immediates most of them passed the 16-bit offset field: a v_add_u32 per read)
    lds_u8* vofs[4];   // [2 dt + plane]
#pragma unroll
    for (int qn = 0; qn < 4; ++qn) {
      vofs[qn] = (lds_u8*)Vs + ((4 * h + q4) * 256 + ((qn ^ q4) << 6) + (16 * g16 + 4 * p4) * 2);
      asm volatile("" : "+v"(vofs[qn]));
    }
    const int c8 = lane & 7;
    // O^T [64 d x 32 queries] = V^T P^T, three products per term, both 32-dim groups per key tile (four independent accumulator chains).
    // The probabilities of a key tile are made where they are consumed (round 5): exp, row sum and the split into (hi, lo) of tile kt + 1
    // are VALU work the scheduler places under the matrix products of tile kt, and at most one tile of P fragments is live.  P^T is the B
    // operand as it stands (k-step s of key tile kt = registers 8 s .. 8 s + 7).
    f32x16 o1[2], o2[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) { o1[dt][e] = 0.f; o2[dt][e] = 0.f; }
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f16x8 ph[2], pl[2];
#if TT_AP_PACKED
      // exp, row sum and split of this key tile, two elements at a time in the packed fp32 instructions (round 6: the loop is VALU-bound -
      // 16 elements cost ~ 160 issue slots as the compiler scheduled the scalar form, 64 of them the quarter-rate exponentials): scale and
      // shift in one v_pk_fma_f32, hi = one v_cvt_pk_f16_f32, and lo = (p - hi) 2^11 as fma(hi, -2^11, p 2^11) - exact, like the subtraction
      // it replaces (p 2^11 and hi 2^11 are exact, their difference has at most 13 significant bits), with hi read as fp16 by the mixed-
      // precision fma (no conversion back).  Bits equal to split_pair's.
#pragma unroll
      for (int e2 = 0; e2 < 8; ++e2) {
        const f32x2 sv = {sacc[kt][2 * e2], sacc[kt][2 * e2 + 1]};
        const f32x2 t = __builtin_elementwise_fma(sv, (f32x2){c, c}, (f32x2){-mc, -mc});
        const f32x2 p = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
        sum += p.x;
        sum += p.y;
        const f16x2 h2 = __builtin_convertvector(p, f16x2);
        const f32x2 pk = p * (f32x2){kPairScale, kPairScale};
        const f32x2 r = pair_residual_scaled(pk, h2);
        const f16x2 l2 = __builtin_convertvector(r, f16x2);
        ph[e2 >> 2][2 * (e2 & 3)] = h2.x; ph[e2 >> 2][2 * (e2 & 3) + 1] = h2.y;
        pl[e2 >> 2][2 * (e2 & 3)] = l2.x; pl[e2 >> 2][2 * (e2 & 3) + 1] = l2.y;
      }
#else
#pragma unroll
      for (int e = 0; e < 16; ++e) {   // exp, row sum and split of this key tile
        const float p = (TT_AP_DBG & 2) ? sacc[kt][e] : __builtin_amdgcn_exp2f(fmaf(sacc[kt][e], c, -mc));
        sum += p;
        _Float16 hi_, lo_;
        if constexpr (TT_AP_DBG & 2) { hi_ = (_Float16)p; lo_ = hi_; }
        else split_pair(p, hi_, lo_);
        ph[e >> 3][e & 7] = hi_;
        pl[e >> 3][e & 7] = lo_;
      }
#endif
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int vrow = (kt * 32 + 16 * s) * 256;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          union { s16x4 s2[2]; f16x8 v; } vh, vl;
          if constexpr (TT_AP_DBG & 8) { vh.v = ph[s ^ 1]; vl.v = pl[s ^ 1]; }
          else {
            vh.s2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vofs[2 * dt] + vrow));
            vh.s2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vofs[2 * dt] + vrow + 2048));
            vl.s2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vofs[2 * dt + 1] + vrow));
            vl.s2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vofs[2 * dt + 1] + vrow + 2048));
          }
          if constexpr (TT_AP_DBG & 4) { o1[dt][s] += (float)vh.v[0] * (float)ph[s][0]; o2[dt][s] += (float)vl.v[0] * (float)pl[s][0]; }
          else {
            o1[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh.v, ph[s], o1[dt], 0, 0, 0);
            o2[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh.v, pl[s], o2[dt], 0, 0, 0);
            o2[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl.v, ph[s], o2[dt], 0, 0, 0);
          }
        }
      }
    }
    AP_STAMP(5);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = __builtin_amdgcn_rcpf(sum);
    if (lse && h == 0 && query < N) lse[((long long)f * H + hd) * N + query] = (mc + __log2f(sum)) * 0.69314718055994530942f;   // natural-log units
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      // The output tile leaves through the wave's private LDS scratch so that a store instruction writes eight whole 128-byte rows (16 B
      // per lane): this 32-dim group as pairs [hi x 32][lo x 32] and / or as 32 floats.  Chunk c of row q sits at c ^ (q & 7).
      float o[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) o[e] = fmaf(o2[dt][e], kPairInvScale, o1[dt][e]) * inv;   // d = (e & 3) + 8 (e >> 2) + 4 h of this group
      if (out_pairs) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          f16x4 vh4, vl4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            _Float16 hi_, lo_;
            split_pair(o[4 * g4 + e], hi_, lo_);
            vh4[e] = hi_;
            vl4[e] = lo_;
          }
          // dims 8 g4 + 4 h .. + 3: hi at byte 16 g4 + 8 h, lo 64 bytes on
          *reinterpret_cast<f16x4*>(Os + r * 128 + ((g4 ^ (r & 7)) << 4) + 8 * h) = vh4;
          *reinterpret_cast<f16x4*>(Os + r * 128 + (((4 + g4) ^ (r & 7)) << 4) + 8 * h) = vl4;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = (lane >> 3) + 8 * i;
          const f16x8 v = *reinterpret_cast<const f16x8*>(Os + row * 128 + ((c8 ^ (row & 7)) << 4));
          const int q = qt * 32 + row;
          if (q < N) *reinterpret_cast<f16x8*>(out_pairs + ((long long)f * N + q) * (2 * Dm) + hd * 128 + dt * 64 + 8 * c8) = v;
        }
      }
      if (out_f32) {
        // fp32 [32 queries][32 dims]: dims 8 g4 + 4 h .. + 3 = 16-byte chunk 2 g4 + h
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 v = {o[4 * g4], o[4 * g4 + 1], o[4 * g4 + 2], o[4 * g4 + 3]};
          *reinterpret_cast<f32x4*>(Os + r * 128 + (((2 * g4 + h) ^ (r & 7)) << 4)) = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = (lane >> 3) + 8 * i;
          const f32x4 v = *reinterpret_cast<const f32x4*>(Os + row * 128 + ((c8 ^ (row & 7)) << 4));
          const int q = qt * 32 + row;
          if (q < N) *reinterpret_cast<f32x4*>(out_f32 + ((long long)f * N + q) * Dm + hd * 64 + dt * 32 + 4 * c8) = v;
        }
      }
    }
    }
#ifdef TT_AP_STAMP
    AP_STAMP(6);
    if (qt >= nqt) { stamp[4] = stamp[3]; stamp[5] = stamp[3]; }
    if (lane == 0 && stamp_it < 4)
      for (int i = 0; i < 7; ++i) stamp_out[((long long)(blockIdx.x * 8 + wave) * 4 + stamp_it) * 8 + i] = stamp[i];
    ++stamp_it;
#endif
  }
}


// ---- Any N: the KV-tiled ("flash") form.  One 8-wave workgroup per (frame, head, query block); a wave owns one 32-query tile and keeps
// its running max / sum / O^T accumulators (64 registers: two 32-dim groups x the two accumulators of the split) across the key loop;
// K and V stream through LDS in stages of 128 keys (the images of the kernel above: 32 KB + 32 KB), double-buffered by LDS-DMA - the
// stage t + 1 loads run under the products of stage t, one workgroup barrier per stage.  A stage is consumed as two 64-key steps
// (32 score + 32 probability registers live): S^T = K Q^T (3 MFMAs per term), the online-softmax update (the rescale of O^T is a
// per-lane scalar: a query is a lane column of both S^T and O^T), P split into pairs, O^T += V^T P^T.  Query tiles are dealt to the
// ceil(tiles / 8) blocks evenly (785 tokens: 25 tiles -> 7 + 6 + 6 + 6).
__global__ __launch_bounds__(512) void attention_fwd_pairs_flash_kernel(const _Float16* __restrict__ qkv, _Float16* __restrict__ out_pairs,
                                                                        float* __restrict__ out_f32, float* __restrict__ lse, int N, int H, float scale,
                                                                        int nb, int nqt, int FH) {
  constexpr int SK = 128;                 // keys per stage
  constexpr int STAGE_B = 2 * SK * 256;   // K image + V image
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_B + 8 * 4096];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* Os = smem + 2 * STAGE_B + wave * 4096;
  const int r = lane & 31, h = lane >> 5;
  // the query blocks of one (frame, head) stream the SAME K / V: they take consecutive dispatch slots of ONE XCD (common.hpp
  // xcd_group_decode), so that K / V come out of that L2 after the first fetch (PMC before: L2 hit rate 0.09, 714 MB fetched per
  // launch of 64 frames x 6 heads x 785 tokens for 231 MB of qkv)
  int blk, fh;
  if (!xcd_group_decode(blockIdx.x, nb, FH, fh, blk)) return;   // (padding workgroups of the last group of eight)
  const int f = fh / H, hd = fh - f * H;
  const int Dm = H * 64;
  const long long RS = 6ll * Dm;
  const _Float16* base = qkv + (long long)f * N * RS + hd * 128;
  const int nst = (N + SK - 1) / SK;

  // K / V of stage st -> buffer st & 1: 32 + 32 pieces of 4 keys x 256 B, four of each per wave - or (TT_APF_LOADERS, round 6) all of
  // them by the waves that own no query tile in this block (785 tokens: 25 tiles on 4 x 8 waves leave one or two per block): an LDS-DMA
  // instruction stalls its wave at the memory queue, and those waves have nothing else to do
  const int l_row = lane >> 4, l_slot = lane & 15;
  const int t0 = blk * nqt / nb, t1 = (blk + 1) * nqt / nb;
#if TT_APF_LOADERS
  const int n_idle = 8 - (t1 - t0);
#else
  const int n_idle = 0;
#endif
  auto issue_stage = [&](int st) {
    unsigned char* Kd = smem + (st & 1) * STAGE_B;
    unsigned char* Vd = Kd + SK * 256;
    if (n_idle > 0 && wave < 8 - n_idle) return;   // (uniform per wave)
    const int first = n_idle > 0 ? wave - (8 - n_idle) : wave, step = n_idle > 0 ? n_idle : 8;
#pragma unroll 4
    for (int piece = first; piece < 32; piece += step) {
      const int key = st * SK + piece * 4 + l_row;
      const int krow = key < N ? key : N - 1;
      const _Float16* src = base + (long long)krow * RS;
      const int kc = l_slot ^ (key & 15);
      const int vc = l_slot ^ ((key & 3) << 2);
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + 2 * Dm + kc * 8),
                                       (void __attribute__((address_space(3)))*)(Kd + piece * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + 4 * Dm + vc * 8),
                                       (void __attribute__((address_space(3)))*)(Vd + piece * 1024), 16, 0, 0);
    }
  };
  issue_stage(0);

  const int qt = t0 + wave;
  const bool active = qt < t1;
  const int query = qt * 32 + r;
  const int qrow = query < N ? query : N - 1;
  f16x8 qh[4], ql[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const _Float16* p = base + (long long)qrow * RS + (ks >> 1) * 64 + (ks & 1) * 16 + 8 * h;
    qh[ks] = *reinterpret_cast<const f16x8*>(p);
    ql[ks] = *reinterpret_cast<const f16x8*>(p + 32);
  }
  int kofs[8];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int ch = (ks >> 1) * 8 + 2 * (ks & 1) + h;
    kofs[ks] = r * 256 + ((ch ^ (r & 15)) << 4);
    kofs[4 + ks] = r * 256 + (((ch + 4) ^ (r & 15)) << 4);
  }
  const int g16 = (lane >> 4) & 1, q4 = (lane >> 2) & 3, p4 = lane & 3;
  int vofs[4];
#pragma unroll
  for (int qn = 0; qn < 4; ++qn) vofs[qn] = (4 * h + q4) * 256 + ((qn ^ q4) << 6) + (16 * g16 + 4 * p4) * 2;

  const float c = scale * 1.44269504088896340736f;
  float m_run = -INFINITY, l_run = 0.f;   // running max (raw score units, both lanes of a query agree) and this lane's share of the sum
  f32x16 o1[2], o2[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) { o1[dt][e] = 0.f; o2[dt][e] = 0.f; }

  for (int st = 0; st < nst; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // stage st has landed for every wave; every wave is done with the buffer stage st + 1 goes into
    if (st + 1 < nst) issue_stage(st + 1);
    if (active) {
      const unsigned char* Ks = smem + (st & 1) * STAGE_B;
      const unsigned char* Vs = Ks + SK * 256;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const int key0 = st * SK + sub * 64;
        if (key0 < N) {                       // (uniform)
          const bool two = key0 + 32 < N;     // the second 32-key tile of this step holds a real key
          f32x16 sacc[2];
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            if (kt == 0 || two) {
              f32x16 s1, s2;
#pragma unroll
              for (int e = 0; e < 16; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) {
                const f16x8 kfh = *reinterpret_cast<const f16x8*>(Ks + (sub * 2 + kt) * 8192 + kofs[ks]);
                const f16x8 kfl = *reinterpret_cast<const f16x8*>(Ks + (sub * 2 + kt) * 8192 + kofs[4 + ks]);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh, qh[ks], s1, 0, 0, 0);
                s2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh, ql[ks], s2, 0, 0, 0);
                s2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl, qh[ks], s2, 0, 0, 0);
              }
#pragma unroll
              for (int e = 0; e < 16; ++e) sacc[kt][e] = fmaf(s2[e], kPairInvScale, s1[e]);
              if (key0 + kt * 32 + 31 >= N) {   // (uniform) the tile reaches past N
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                  const int key = key0 + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                  if (key >= N) sacc[kt][e] = -INFINITY;
                }
              }
            } else {
#pragma unroll
              for (int e = 0; e < 16; ++e) sacc[kt][e] = -INFINITY;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          float mx = m_run;
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sacc[kt][e]);
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));          // finite: key0 < N and the first keys of the step sit in lane half 0 / 1 alike
          const float mc = mx * c;
          const float alpha = __builtin_amdgcn_exp2f(fmaf(m_run, c, -mc));   // 0 at the first step (m_run = -inf)
          m_run = mx;
          float sum = 0.f;
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kt][e], c, -mc));
              sacc[kt][e] = p;
              sum += p;
            }
          l_run = fmaf(l_run, alpha, sum);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { o1[dt][e] *= alpha; o2[dt][e] *= alpha; }
          f16x8 ph[2][2], pl[2][2];
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                _Float16 hi_, lo_;
                split_pair(sacc[kt][8 * s + j], hi_, lo_);
                ph[kt][s][j] = hi_;
                pl[kt][s][j] = lo_;
              }
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
              if (kt == 0 || two) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                  const unsigned char* vrow = Vs + ((sub * 2 + kt) * 32 + 16 * s) * 256;
                  union { s16x4 s2[2]; f16x8 v; } vh, vl;
                  vh.s2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vrow + vofs[2 * dt]));
                  vh.s2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vrow + 2048 + vofs[2 * dt]));
                  vl.s2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vrow + vofs[2 * dt + 1]));
                  vl.s2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vrow + 2048 + vofs[2 * dt + 1]));
                  o1[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh.v, ph[kt][s], o1[dt], 0, 0, 0);
                  o2[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh.v, pl[kt][s], o2[dt], 0, 0, 0);
                  o2[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl.v, ph[kt][s], o2[dt], 0, 0, 0);
                }
              }
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  if (!active) return;
  const float lsum = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = __builtin_amdgcn_rcpf(lsum);
  if (lse && h == 0 && query < N) lse[((long long)f * H + hd) * N + query] = (m_run * c + __log2f(lsum)) * 0.69314718055994530942f;
  const int c8 = lane & 7;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    float o[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = fmaf(o2[dt][e], kPairInvScale, o1[dt][e]) * inv;
    if (out_pairs) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f16x4 vh4, vl4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          _Float16 hi_, lo_;
          split_pair(o[4 * g4 + e], hi_, lo_);
          vh4[e] = hi_;
          vl4[e] = lo_;
        }
        *reinterpret_cast<f16x4*>(Os + r * 128 + ((g4 ^ (r & 7)) << 4) + 8 * h) = vh4;
        *reinterpret_cast<f16x4*>(Os + r * 128 + (((4 + g4) ^ (r & 7)) << 4) + 8 * h) = vl4;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (lane >> 3) + 8 * i;
        const f16x8 v = *reinterpret_cast<const f16x8*>(Os + row * 128 + ((c8 ^ (row & 7)) << 4));
        const int q = qt * 32 + row;
        if (q < N) *reinterpret_cast<f16x8*>(out_pairs + ((long long)f * N + q) * (2 * Dm) + hd * 128 + dt * 64 + 8 * c8) = v;
      }
    }
    if (out_f32) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 v = {o[4 * g4], o[4 * g4 + 1], o[4 * g4 + 2], o[4 * g4 + 3]};
        *reinterpret_cast<f32x4*>(Os + r * 128 + (((2 * g4 + h) ^ (r & 7)) << 4)) = v;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (lane >> 3) + 8 * i;
        const f32x4 v = *reinterpret_cast<const f32x4*>(Os + row * 128 + ((c8 ^ (row & 7)) << 4));
        const int q = qt * 32 + row;
        if (q < N) *reinterpret_cast<f32x4*>(out_f32 + ((long long)f * N + q) * Dm + hd * 64 + dt * 32 + 4 * c8) = v;
      }
    }
  }
}

}  // namespace tt

using namespace tt;

extern "C" int tt_attention_fwd_pairs(const void* qkv_pairs, void* out_pairs, float* out_f32, float* lse, int F, int N, int H, int head_dim,
                                      float scale, tt_stream_t stream) {
  TT_REQUIRE(qkv_pairs && (out_pairs || out_f32), "attention_fwd_pairs: null input / no output");
  TT_REQUIRE(F > 0 && N > 0 && H > 0 && scale > 0.f, "attention_fwd_pairs: bad shape / non-positive scale");
  TT_REQUIRE(head_dim == 64, "attention_fwd_pairs: head_dim must be 64 (got %d)", head_dim);
  TT_REQUIRE(aligned16(qkv_pairs) && (!out_pairs || aligned16(out_pairs)) && (!out_f32 || aligned16(out_f32)),
             "attention_fwd_pairs: buffers must be 16-byte aligned");
  hipStream_t s = as_stream(stream);
  const _Float16* q = static_cast<const _Float16*>(qkv_pairs);
  _Float16* o = static_cast<_Float16*>(out_pairs);
  const bool resident = tuning_knob(KNOB_ATTN_PAIRS_FLASH) == 0;   // (knob 1: the KV-tiled kernel at every N - tests and A/B)
  const int grid = tuning_knob(KNOB_ATTN_PAIRS_PERSIST) ? (F * H < device_cu_count() ? F * H : device_cu_count()) : F * H;
  if (N <= 224 && resident) {
    hipLaunchKernelGGL((attention_fwd_pairs_kernel<7>), dim3(grid), dim3(512), 0, s, q, o, out_f32, lse, N, H, scale, F * H);
  } else if (N <= 256 && resident) {
    hipLaunchKernelGGL((attention_fwd_pairs_kernel<8>), dim3(grid), dim3(512), 0, s, q, o, out_f32, lse, N, H, scale, F * H);
  } else {
    const int nqt = (N + 31) / 32, nb = (nqt + 7) / 8;
    TT_REQUIRE((long long)F * H * nb < 0x7fffffffLL, "attention_fwd_pairs: grid too large");
    hipLaunchKernelGGL(attention_fwd_pairs_flash_kernel, dim3(xcd_group_grid(F * H, nb)), dim3(512), 0, s, q, o, out_f32, lse, N, H, scale, nb, nqt,
                       F * H);
  }
  TT_CHECK_LAUNCH("attention_fwd_pairs");
  return TT_OK;
}
