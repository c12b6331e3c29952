// Probe (round 4): does v_mfma_f32_32x32x16_f16 keep fp16 subnormal A/B inputs, how does it round inside a k-step, and is its
// A/B lane map the bf16 one?  Build: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_f16_probe.hip -o tools/bin/mfma_f16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A [32][16], B [16][32] row-major in global memory; lane l: r = l & 31, h = l >> 5 holds A[r][8h + j], B[8h + j][r]
__global__ void probe(const _Float16* A, const _Float16* B, float* D) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[r * 16 + 8 * h + j]; b[j] = B[(8 * h + j) * 32 + r]; }
  f32x16 c;
  for (int e = 0; e < 16; ++e) c[e] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int e = 0; e < 16; ++e) D[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = c[e];   // row (reg), col = lane & 31
}

int main() {
  _Float16 hA[32 * 16], hB[16 * 32];
  float hD[32 * 32];
  _Float16 *dA, *dB; float* dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  auto run = [&]() {
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  };
  // 1. lane map / layout with exact integers: A[i][k] = i + 1 if k == (i % 16) else 0; B[k][j] = (k + 1) * 64 + j  (asymmetric)
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) hA[i * 16 + k] = (_Float16)((k == i % 16) ? (i + 1) : 0);
  for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) hB[k * 32 + j] = (_Float16)((k + 1) * 64 + j);
  run();
  int bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    const float ref = (float)(i + 1) * (float)((i % 16 + 1) * 64 + j);
    if (hD[i * 32 + j] != ref) ++bad;
  }
  printf("layout check (bf16 lane map on the f16 instruction): %d mismatches of 1024\n", bad);
  // 2. subnormal inputs: A = 2^-20 (fp16 subnormal), B = 2^10 -> every product 2^-10, 16 of them = 2^-6 = 0.015625
  for (int i = 0; i < 32 * 16; ++i) hA[i] = (_Float16)9.5367431640625e-07f;
  for (int i = 0; i < 16 * 32; ++i) hB[i] = (_Float16)1024.f;
  run();
  printf("subnormal A (2^-20) x 2^10, 16 terms: D = %.9g (kept: 0.015625, flushed: 0)\n", hD[0]);
  for (int i = 0; i < 32 * 16; ++i) hA[i] = (_Float16)1024.f;
  for (int i = 0; i < 16 * 32; ++i) hB[i] = (_Float16)9.5367431640625e-07f;
  run();
  printf("subnormal B (2^-20) x 2^10, 16 terms: D = %.9g\n", hD[0]);
  // 3. rounding inside one instruction: 1 + 2^-24 + 2^-24 (k = 0, 1, 2)
  for (int i = 0; i < 32 * 16; ++i) hA[i] = (_Float16)0.f;
  for (int i = 0; i < 16 * 32; ++i) hB[i] = (_Float16)0.f;
  for (int i = 0; i < 32; ++i) { hA[i * 16 + 0] = (_Float16)1.f; hA[i * 16 + 1] = (_Float16)0.000244140625f; hA[i * 16 + 9] = (_Float16)0.000244140625f; }
  for (int j = 0; j < 32; ++j) { hB[0 * 32 + j] = (_Float16)1.f; hB[1 * 32 + j] = (_Float16)0.000244140625f; hB[9 * 32 + j] = (_Float16)0.000244140625f; }
  run();
  printf("1 + 2^-24 + 2^-24 in one MFMA: D - 1 = %.9g (2^-23 = %.9g: summed wider than fp32 per term; 0: fp32 RNE per term)\n", (double)hD[0] - 1.0, ldexp(1.0, -23));
  // 4. many tiny terms against a big one: 1 + 15 * 2^-26 -> exact 1 + 5.6e-8... (below half ulp 2^-24 = 5.96e-8): expect 1
  //    and 1 + 15 * 2^-25 = 1 + 4.47e-7 -> nearest fp32: 1 + 4 * 2^-23 (4.77e-7) or 1 + 3 * 2^-23 (3.58e-7)
  for (int i = 0; i < 32; ++i) for (int k = 1; k < 16; ++k) { hA[i * 16 + k] = (_Float16)0.000244140625f; }
  for (int j = 0; j < 32; ++j) for (int k = 1; k < 16; ++k) { hB[k * 32 + j] = (_Float16)0.0001220703125f; }   // 2^-12 * 2^-13 = 2^-25
  run();
  printf("1 + 15 * 2^-25 in one MFMA: D - 1 = %.9g (exact 4.47034836e-07; fp32 neighbours 3.57627869e-07 / 4.76837158e-07)\n", (double)hD[0] - 1.0);
  return 0;
}
