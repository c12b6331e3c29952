// What one CU can stream from L2 into LDS (development aid, round 3): the 8-phase GEMM's main loop turned out to sit on the rate of its
// LDS-DMA stream (~21 B/clk/CU, whether the operands come from HBM or from a warm L2), so this measures the alternatives side by side.
// One 8-wave workgroup per CU, every wave streams 1 KiB pieces of an L2-resident 2 MiB region into its own LDS slots with a window of
// batches in flight (counted vmcnt, no barrier):
//   mode 0: buffer_load_dwordx4 ... lds     (16 B per lane, the GEMM's form)
//   mode 1: buffer_load_dword ... lds       (4 B per lane)
//   mode 2: global_load_dwordx4 -> VGPR     (no LDS write; registers kept alive)
//   mode 3: global_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 4: half the bytes as mode 0, half as mode 3
//   mode 5: mode 0 with the eight waves sharing ONE 8 KiB source window per step (all eight read the same lines)
//   mode 9: mode 0 with a dependent chain of 28 v_fma_f32 (~120 cycles) after every DMA instruction; mode 10: that chain alone
//            (does a DMA instruction hold its wave, or only the wave's NEXT vector-memory instruction?)
//   mode 6 / 7 / 8: mode 0 with 32 / 8 / 4 instead of 16 instructions per wave in flight (is a wave's stream latency-bound?)
// usage: dma_rate <mode> [iters] [waves (1..8)]      prints bytes per clock per CU and GB/s per CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void dma16(const void* base, void* lds, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void dma4(const void* base, void* lds, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 4, voff, soff, 0, 0);
}
__device__ __forceinline__ f32x4 ld16(const void* base, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

constexpr int REGION = 2 << 20;   // bytes streamed over and over (L2 resident: 2 MiB per workgroup's view, shared by all)
constexpr int PIECES = 8;         // pieces (1 KiB at 16 B per lane) per wave per step

template <int MODE>
__global__ __launch_bounds__(512) void k(const unsigned char* __restrict__ src, float* __restrict__ sink, int iters, int waves,
                                         unsigned long long* __restrict__ stamps) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[8 * 2 * PIECES * 1024];   // 128 KiB: wave x (2 batches) x 8 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: no waterfall loops)
  if (wave >= waves) return;
  const unsigned char* rs = src;
  unsigned char* mine = lds + wave * 2 * PIECES * 1024;
  f32x4 keep[PIECES], acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < PIECES; ++j) keep[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  int pos = (MODE == 5 ? 0 : wave * PIECES * 1024) + (blockIdx.x & 31) * 65536;
  for (int it = 0; it < iters; ++it) {
    unsigned char* dst = mine + (it & 1) * PIECES * 1024;
    if constexpr (MODE == 9 || MODE == 10) {
#pragma unroll
      for (int j = 0; j < PIECES; ++j) {
        if constexpr (MODE == 9) dma16(rs, dst + j * 1024, lane * 16, (pos + j * 1024) & (REGION - 1));
#pragma unroll
        for (int q = 0; q < 28; ++q) acc[0] = __builtin_fmaf(acc[0], 1.0001f, 0.5f);
        asm volatile("" : "+v"(acc[0]));
      }
      if constexpr (MODE == 9) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if constexpr (MODE == 0 || MODE >= 5) {
#pragma unroll
      for (int j = 0; j < PIECES; ++j) {
        dma16(rs, dst + j * 1024, lane * 16, (pos + j * 1024) & (REGION - 1));
        if constexpr (MODE == 8) { if (j == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
      }
      if constexpr (MODE == 6) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if constexpr (MODE == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if constexpr (MODE == 8) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int j = 0; j < PIECES; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) dma4(rs, dst + j * 1024 + q * 256, lane * 4, (pos + j * 1024 + q * 256) & (REGION - 1));
      asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    } else if constexpr (MODE == 2 || MODE == 3) {
      // two register sets: the set loaded in the previous step is consumed while this step's loads are in flight
      f32x4 v[PIECES];
#pragma unroll
      for (int j = 0; j < PIECES; ++j) v[j] = ld16(rs, lane * 16, (pos + j * 1024) & (REGION - 1));
#pragma unroll
      for (int j = 0; j < PIECES; ++j) {
        if constexpr (MODE == 3) *reinterpret_cast<f32x4*>(dst + j * 1024 + lane * 16) = keep[j];
        else acc += keep[j];
      }
      // (second half of the unrolled pair)
      const int pos2 = pos + 8 * PIECES * 1024;
#pragma unroll
      for (int j = 0; j < PIECES; ++j) keep[j] = ld16(rs, lane * 16, (pos2 + j * 1024) & (REGION - 1));
#pragma unroll
      for (int j = 0; j < PIECES; ++j) {
        if constexpr (MODE == 3) *reinterpret_cast<f32x4*>(dst + PIECES * 1024 * 0 + j * 1024 + lane * 16) = v[j];
        else acc += v[j];
      }
      pos += 8 * PIECES * 1024;
      ++it;
    } else {
      f32x4 v[PIECES / 2];
#pragma unroll
      for (int j = 0; j < PIECES / 2; ++j) dma16(rs, dst + j * 1024, lane * 16, (pos + j * 1024) & (REGION - 1));
#pragma unroll
      for (int j = 0; j < PIECES / 2; ++j) v[j] = ld16(rs, lane * 16, (pos + (PIECES / 2 + j) * 1024) & (REGION - 1));
#pragma unroll
      for (int j = 0; j < PIECES / 2; ++j) *reinterpret_cast<f32x4*>(dst + (PIECES / 2 + j) * 1024 + lane * 16) = keep[j];
      const int pos2 = pos + 8 * PIECES * 1024;
      unsigned char* dst2 = mine + ((it + 1) & 1) * PIECES * 1024;
#pragma unroll
      for (int j = 0; j < PIECES / 2; ++j) dma16(rs, dst2 + j * 1024, lane * 16, (pos2 + j * 1024) & (REGION - 1));
#pragma unroll
      for (int j = 0; j < PIECES / 2; ++j) keep[j] = ld16(rs, lane * 16, (pos2 + (PIECES / 2 + j) * 1024) & (REGION - 1));
#pragma unroll
      for (int j = 0; j < PIECES / 2; ++j) *reinterpret_cast<f32x4*>(dst2 + (PIECES / 2 + j) * 1024 + lane * 16) = v[j];
      pos += 8 * PIECES * 1024;
      ++it;
    }
    pos += (MODE == 5 ? 1 : 8) * PIECES * 1024;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  f32x4 s = keep[0] + acc;
#pragma unroll
  for (int j = 1; j < PIECES; ++j) s += keep[j];
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) sink[tid] = s[0] + lds[tid];
  if (tid == 0 && blockIdx.x == 0) {
    stamps[0] = t1 - t0;
    stamps[1] = r1 - r0;
  }
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0, iters = argc > 2 ? atoi(argv[2]) : 4000, waves = argc > 3 ? atoi(argv[3]) : 8;
  unsigned char* src;
  float* sink;
  unsigned long long* stamps;
  (void)hipMalloc(&src, REGION);
  (void)hipMemset(src, 1, REGION);
  (void)hipMalloc(&sink, 4096);
  (void)hipMalloc(&stamps, 16);
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    switch (mode) {
      case 0: hipLaunchKernelGGL(k<0>, dim3(ncu), dim3(512), 0, 0, src, sink, iters, waves, stamps); break;
      case 1: hipLaunchKernelGGL(k<1>, dim3(ncu), dim3(512), 0, 0, src, sink, iters, waves, stamps); break;
      case 2: hipLaunchKernelGGL(k<2>, dim3(ncu), dim3(512), 0, 0, src, sink, iters, waves, stamps); break;
      case 3: hipLaunchKernelGGL(k<3>, dim3(ncu), dim3(512), 0, 0, src, sink, iters, waves, stamps); break;
      case 4: hipLaunchKernelGGL(k<4>, dim3(ncu), dim3(512), 0, 0, src, sink, iters, waves, stamps); break;
      case 6: hipLaunchKernelGGL(k<6>, dim3(ncu), dim3(512), 0, 0, src, sink, iters, waves, stamps); break;
      case 7: hipLaunchKernelGGL(k<7>, dim3(ncu), dim3(512), 0, 0, src, sink, iters, waves, stamps); break;
      case 8: hipLaunchKernelGGL(k<8>, dim3(ncu), dim3(512), 0, 0, src, sink, iters, waves, stamps); break;
      case 9: hipLaunchKernelGGL(k<9>, dim3(ncu), dim3(512), 0, 0, src, sink, iters, waves, stamps); break;
      case 10: hipLaunchKernelGGL(k<10>, dim3(ncu), dim3(512), 0, 0, src, sink, iters, waves, stamps); break;
      default: hipLaunchKernelGGL(k<5>, dim3(ncu), dim3(512), 0, 0, src, sink, iters, waves, stamps); break;
    }
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long h[2];
  (void)hipMemcpy(h, stamps, 16, hipMemcpyDeviceToHost);
  const double bytes = (double)iters * waves * PIECES * 1024;
  printf("mode %d  waves %d  %d CUs: %.1f us  %.0f cycles/step  %.1f B/clk/CU  %.1f GB/s/CU  (%.2f GHz, chip %.1f TB/s)\n", mode, waves, ncu, ms * 1e3, (double)h[0] / iters, bytes / (double)h[0],
         bytes / (ms * 1e-3) / 1e9, (double)h[0] / ((double)h[1] * 10.0), bytes * ncu / (ms * 1e-3) / 1e12);
  return 0;
}
