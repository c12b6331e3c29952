// Bare bf16 MFMA loops on random operands: v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 at equal flops per wave
// (MI355X_MICROARCH.md, DVFS give-back item 7: the chip holds a higher clock on the 16x16x32 shape).
//   usage: mfma_bf16_shapes [waves_per_simd=1] [zero=0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(256) void loop_kernel(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(tid * 8 + i) & 0xffff]; b[i] = src[(tid * 8 + 4 + i) & 0xffff]; }
  float sum = 0.f;
  if (SHAPE == 0) {   // 64 x 64 wave tile as 2 x 2 blocks of 32x32, k = 16 per MFMA: 4 MFMAs = 64*64*16 MACs
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)      // 2 k-steps of 16 = 32 k
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + 2 * ks], b[j + 2 * ks], acc[i * 2 + j], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) sum += acc[i][e];
  } else {            // the same tile as 4 x 4 blocks of 16x16, k = 32 per MFMA: 16 MFMAs = 64*64*32 MACs
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) sum += acc[i][e];
  }
  out[tid] = sum;
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 1, zero = argc > 2 ? atoi(argv[2]) : 0;
  const int blocks = 256 * wps, iters = 20000;
  std::vector<unsigned short> h(65536 * 8);
  srand(1);
  for (auto& v : h) v = zero ? 0 : (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));   // bf16 around +-1
  bf16x8* src; float* out;
  hipMalloc(&src, h.size() * 2); hipMalloc(&out, blocks * 256 * 4);
  hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flops = 2.0 * 64 * 64 * 32 * (double)iters * blocks * 4;   // per iteration a wave does 64*64*32 MACs in both shapes
  for (int rd = 0; rd < 4; ++rd)
    for (int shape = 0; shape < 2; ++shape) {      // interleaved rounds
      hipEventRecord(e0);
      for (int r = 0; r < 20; ++r) {
        if (shape == 0) hipLaunchKernelGGL(loop_kernel<0>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
        else hipLaunchKernelGGL(loop_kernel<1>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rd) printf("round %d %s waves/SIMD %d %s: %.0f TFLOP/s\n", rd, shape ? "16x16x32" : "32x32x16", wps, zero ? "zeros " : "random", flops * 20 / (ms * 1e-3) / 1e12);
    }
  return 0;
}
