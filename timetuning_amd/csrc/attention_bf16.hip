// Fused attention forward on bf16 operands (dino_vision_transformer.py:122-129) for the bf16 storage path of the frozen
// blocks (BASELINE config C4's "MFMA bf16 path"): qkv [F, N, 3 H 64] bf16 as the qkv Linear's epilogue left it ->
// out [F, N, H 64] bf16, the proj Linear's pre-split A operand.  fp32 scores, softmax and accumulation; N <= 256, head_dim 64.
//
// One workgroup (4 waves) per (frame, head).  K and V of that head go HBM -> LDS once, by LDS-DMA, and stay:
//   K image  [key][64 d], 128-byte rows, 16-byte chunks XOR-swizzled by (key >> 1) & 7  -> conflict-free ds_read_b128 A fragments
//   V image  [key][64 d], 128-byte rows, the two 64-byte halves swapped when (key >> 1) & 1 -> conflict-free
//            ds_read_b64_tr_b16: V stays row-major (coalesced DMA) and is consumed transposed, as V^T fragments.
// A wave owns 32-query tiles.  S^T = K Q^T (v_mfma_f32_32x32x16_bf16, A = K rows, B = Q rows straight from global memory):
// a query's whole score row then sits in ONE lane column (2 lanes), so the softmax is register-local plus one cross-half
// exchange, and P^T is already the B operand of O^T = V^T P^T (the accumulator-as-operand map of the 32x32 C layout:
// element j of lane half h of k-step s is key 16 s + 8 (j >> 2) + 4 h + (j & 3)); the V^T fragment is gathered in the same key
// order by two transposed LDS reads.  Keys >= N are clamped on load and masked to -inf; nothing N x N ever leaves registers.
#include "common.hpp"

namespace tt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// DBG (timing-study builds only, -DTT_ABF_ABLATE + TT_ABF_DBG): 1 no MFMAs, 2 no K/V DMA, 4 no Q loads, 8 no exponentials, 16 no stores
template <int NKT, int DBG = 0>
__global__ __launch_bounds__(256, 2) void attention_fwd_bf16_kernel(const __bf16* __restrict__ qkv, __bf16* __restrict__ out, int N, int H,
                                                                    float scale) {
  constexpr int KROWS = NKT * 32;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * KROWS * 128 + 4 * 4096];
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + KROWS * 128;
  unsigned char* Os = smem + 2 * KROWS * 128 + (threadIdx.x >> 6) * 4096;   // this wave's output tile: 32 queries x 128 B, chunks swizzled
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int fh = blockIdx.x, f = fh / H, hd = fh - f * H;
  const int D3 = 3 * H * 64, Dm = H * 64;
  const __bf16* base = qkv + (long long)f * N * D3 + hd * 64;

  // ---- K and V: 8 keys x 128 B per DMA piece; lane -> (key, slot), source chunk = slot ^ swizzle(key)
  {
    const int l_row = lane >> 3, l_slot = lane & 7;
    for (int piece = wave; piece < ((DBG & 2) ? 0 : KROWS / 8); piece += 4) {
      const int key = piece * 8 + l_row;
      const int krow = key < N ? key : N - 1;
      const __bf16* src = base + (long long)krow * D3;
      const int kc = l_slot ^ ((key >> 1) & 7);
      const int vc = l_slot ^ (((key >> 1) & 1) << 2);
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + Dm + kc * 8),
                                       (void __attribute__((address_space(3)))*)(Ks + piece * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + 2 * Dm + vc * 8),
                                       (void __attribute__((address_space(3)))*)(Vs + piece * 1024), 16, 0, 0);
    }
  }
  __syncthreads();

  const int nqt = (N + 31) / 32;
  for (int qt = wave; qt < nqt; qt += 4) {
    // Q fragments (B operand): lane (query r, half h) holds Q[query][16 ks + 8 h + j]
    const int query = qt * 32 + r;
    const int qrow = query < N ? query : N - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if constexpr (DBG & 4) {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[ks][j] = (__bf16)(0.01f * (float)((lane + j + ks) & 15));
      } else {
        qf[ks] = *reinterpret_cast<const bf16x8*>(base + (long long)qrow * D3 + 16 * ks + 8 * h);
      }
    }

    f32x16 sacc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[kt][e] = 0.f;
      const int krow = kt * 32 + r;
      const unsigned char* kp = Ks + krow * 128;
      const int sw = (krow >> 1) & 7;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kp + (((2 * ks + h) ^ sw) << 4));
        if constexpr (DBG & 1) sacc[kt][ks] += (float)kf[0] * (float)qf[ks][1];
        else sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[kt], 0, 0, 0);
      }
    }
    // softmax over the keys of this lane column: registers, then the other half (lane ^ 32).  The softmax phase is VALU time the
    // matrix pipe idles through (112 elements per lane and query tile), so it is kept to a max, one fma and one v_exp per element:
    // the logit scale is folded into the exponent's constant, p = 2^(s c - m c) with c = scale log2(e), and only the key tiles that
    // reach past N pay for the mask.
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
    const float c = scale * 1.44269504088896340736f;   // (scale > 0: the max of the raw scores is the max of the scaled ones)
    const float mc = mx * c;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = (DBG & 8) ? fmaf(sacc[kt][e], c, -mc) : __builtin_amdgcn_exp2f(fmaf(sacc[kt][e], c, -mc));
        sacc[kt][e] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = __builtin_amdgcn_rcpf(sum);

    // O^T [64 d x 32 queries] = V^T P^T
    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
    const int g16 = (lane >> 4) & 1, q4 = (lane >> 2) & 3, p4 = lane & 3;   // position inside the 16-lane transpose group
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (__bf16)sacc[kt][8 * s + j];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          // block rows = keys key0 .. key0 + 3, block columns = d0 .. d0 + 15; this lane supplies row q4, columns 4 p4 .. 4 p4 + 3
          s16x4 lo, hi;
          {
            const int key = kt * 32 + 16 * s + 4 * h + q4;
            const int col = dt * 32 + 16 * g16 + 4 * p4;                     // bf16 index inside the 64-wide row
            const int phys = col * 2 ^ ((((key >> 1) & 1)) << 6);            // swap the 64-byte halves
            lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Vs + key * 128 + phys));
          }
          {
            const int key = kt * 32 + 16 * s + 8 + 4 * h + q4;
            const int col = dt * 32 + 16 * g16 + 4 * p4;
            const int phys = col * 2 ^ ((((key >> 1) & 1)) << 6);
            hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Vs + key * 128 + phys));
          }
          union { s16x4 s2[2]; bf16x8 v; } vf;
          vf.s2[0] = lo;
          vf.s2[1] = hi;
          if constexpr (DBG & 1) oacc[dt][s] += (float)vf.v[0] * (float)pf[1];
          else oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf, oacc[dt], 0, 0, 0);
        }
      }
    }
    // The output tile leaves through the wave's private LDS scratch so that a store instruction writes EIGHT WHOLE 128-byte rows (16 B per
    // lane): written straight from the accumulator layout a row would go out as sixteen 8-byte pieces in eight instructions - 42 of the
    // kernel's 56 us (tools/abf_ablate.py, round 3).  16-byte chunk c of row q sits at chunk position c ^ (q & 7).
    if constexpr (!(DBG & 16)) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          bf16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (__bf16)(oacc[dt][4 * g4 + e] * inv);
          *reinterpret_cast<bf16x4*>(Os + r * 128 + (((dt * 4 + g4) ^ (r & 7)) << 4) + 8 * h) = v;
        }
      const int c8 = lane & 7;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (lane >> 3) + 8 * i;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(Os + row * 128 + ((c8 ^ (row & 7)) << 4));
        const int q = qt * 32 + row;
        if (q < N) *reinterpret_cast<bf16x8*>(out + ((long long)f * N + q) * Dm + hd * 64 + 8 * c8) = v;
      }
    }
  }
}

}  // namespace tt

using namespace tt;

extern "C" int tt_attention_fwd_bf16(const void* qkv, void* out, int F, int N, int H, int head_dim, float scale, tt_stream_t stream) {
  TT_REQUIRE(qkv && out, "attention_fwd_bf16: null pointer");
  TT_REQUIRE(F > 0 && N > 0 && H > 0 && scale > 0.f, "attention_fwd_bf16: bad shape / non-positive scale");
  TT_REQUIRE(head_dim == 64, "attention_fwd_bf16: head_dim must be 64 (got %d)", head_dim);
  TT_REQUIRE(N <= 256, "attention_fwd_bf16: N <= 256 tokens (got %d); longer sequences use the fp32 kernel", N);
  TT_REQUIRE(aligned16(qkv) && aligned16(out), "attention_fwd_bf16: buffers must be 16-byte aligned");
  hipStream_t s = as_stream(stream);
  const __bf16* q = static_cast<const __bf16*>(qkv);
  __bf16* o = static_cast<__bf16*>(out);
#ifdef TT_ABF_ABLATE
  {
    const char* e = getenv("TT_ABF_DBG");
    const int dbg = e ? atoi(e) : 0;
    TT_REQUIRE(N <= 224, "ablation build: N <= 224");
#define ABF(D) case D: hipLaunchKernelGGL((attention_fwd_bf16_kernel<7, D>), dim3(F * H), dim3(256), 0, s, q, o, N, H, scale); break;
    switch (dbg) { ABF(0) ABF(1) ABF(2) ABF(4) ABF(8) ABF(16) ABF(22) ABF(9) ABF(31) ABF(30) default: TT_REQUIRE(false, "TT_ABF_DBG"); }
#undef ABF
    TT_CHECK_LAUNCH("attention_fwd_bf16");
    return TT_OK;
  }
#endif
  if (N <= 224) hipLaunchKernelGGL((attention_fwd_bf16_kernel<7>), dim3(F * H), dim3(256), 0, s, q, o, N, H, scale);
  else hipLaunchKernelGGL((attention_fwd_bf16_kernel<8>), dim3(F * H), dim3(256), 0, s, q, o, N, H, scale);
  TT_CHECK_LAUNCH("attention_fwd_bf16");
  return TT_OK;
}
