// Clip input pipeline on the GPU: the pixel work of the reference's training transforms
// (video_transformations.py, wired at time_tuning.py:588-593), which run per clip on the host through Pillow in the
// DataLoader workers.  Frames arrive as interleaved uint8 RGB [F, H, W, 3] (what a decoder produces) and every kernel
// reproduces Pillow's integer / float32 arithmetic bit for bit (oracle/image_ops.py is pinned against Pillow itself):
//   resample_h / resample_v   Resample.c's two-pass bilinear convolution with 22-bit fixed-point taps and uint8 rounding after
//                             each pass; crop offsets fold Image.crop() in; the vertical pass can finish the chain:
//                             horizontal flip + ToTensor + (x - mean) / std -> float32 [F, 3, H, W]
//   gray / enhance            rgb2l, ImageEnhance.{Brightness, Contrast, Color} = Blend.c against black / mean gray / gray
//   hue                       rgb -> hsv -> H += shift (uint8 wrap) -> rgb, Convert.c's float / double mix
//   box_blur                  BoxBlur.c's extended box filter (3 passes per direction approximate the Gaussian)
// All of these are byte work bound by HBM: one thread per output pixel (3 channels), coalesced along the row.
#include "common.hpp"

// Pillow's C code runs as separate multiplies and adds; a fused multiply-add rounds once and can land on the other side of a
// uint8 truncation, so contraction is off for this whole file (bit-exactness is the contract here, not speed).
#pragma clang fp contract(off)

namespace tt {

constexpr int IM_THREADS = 256;
constexpr int IM_PB = 22;  // PRECISION_BITS of Resample.c

__device__ __forceinline__ unsigned char clip8(long long v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// ---- Resample.c, horizontal pass: out[f][y][xx][c] = clip8((2^21 + sum_x in[f][y0 + y][x0 + xmin + x][c] * k[xx][x]) >> 22)
__global__ __launch_bounds__(IM_THREADS) void resample_h_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                                                const int* __restrict__ kk, const int* __restrict__ bounds, int H, int W,
                                                                int y0, int x0, int h, int OW, int ksize) {
  const long long idx = (long long)blockIdx.x * IM_THREADS + threadIdx.x;
  const long long per_frame = (long long)h * OW;
  if (idx >= per_frame) return;
  const int f = blockIdx.y, y = (int)(idx / OW), xx = (int)(idx - (long long)y * OW);
  const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
  const unsigned char* row = in + (((size_t)f * H + (y0 + y)) * W + x0 + xmin) * 3;
  const int* k = kk + (size_t)xx * ksize;
  int a0 = 1 << (IM_PB - 1), a1 = a0, a2 = a0;
  for (int x = 0; x < cnt; ++x) {
    const int w = k[x];
    a0 += row[3 * x] * w;
    a1 += row[3 * x + 1] * w;
    a2 += row[3 * x + 2] * w;
  }
  unsigned char* o = out + ((size_t)f * per_frame + idx) * 3;
  o[0] = clip8(a0 >> IM_PB);
  o[1] = clip8(a1 >> IM_PB);
  o[2] = clip8(a2 >> IM_PB);
}

// vertical pass over in [F][Hin][W][3] rows y0 .. ; out either uint8 [F][OH][W][3] or, with `fout`, the finished tensor
// float32 [F][3][OH][W] = (flip_x(pixel) / 255 - mean) / std
__global__ __launch_bounds__(IM_THREADS) void resample_v_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                                                float* __restrict__ fout, const int* __restrict__ kk,
                                                                const int* __restrict__ bounds, int Hin, int W, int y0, int OH, int ksize,
                                                                int flip, float m0, float m1, float m2, float s0, float s1, float s2) {
  const long long idx = (long long)blockIdx.x * IM_THREADS + threadIdx.x;
  const long long per_frame = (long long)OH * W;
  if (idx >= per_frame) return;
  const int f = blockIdx.y, yy = (int)(idx / W), x = (int)(idx - (long long)yy * W);
  const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
  const unsigned char* col = in + (((size_t)f * Hin + (y0 + ymin)) * W + x) * 3;
  const int* k = kk + (size_t)yy * ksize;
  int a0 = 1 << (IM_PB - 1), a1 = a0, a2 = a0;
  for (int y = 0; y < cnt; ++y) {
    const int w = k[y];
    const unsigned char* p = col + (size_t)y * W * 3;
    a0 += p[0] * w;
    a1 += p[1] * w;
    a2 += p[2] * w;
  }
  const unsigned char r = clip8(a0 >> IM_PB), g = clip8(a1 >> IM_PB), b = clip8(a2 >> IM_PB);
  if (fout) {
    const int ox = flip ? W - 1 - x : x;
    const size_t plane = (size_t)OH * W, base = (size_t)f * 3 * plane + (size_t)yy * W + ox;
    fout[base] = ((float)r / 255.0f - m0) / s0;
    fout[base + plane] = ((float)g / 255.0f - m1) / s1;
    fout[base + 2 * plane] = ((float)b / 255.0f - m2) / s2;
  } else {
    unsigned char* o = out + ((size_t)f * per_frame + idx) * 3;
    o[0] = r;
    o[1] = g;
    o[2] = b;
  }
}

__device__ __forceinline__ unsigned char gray_of(unsigned r, unsigned g, unsigned b) {
  return (unsigned char)((r * 19595u + g * 38470u + b * 7471u + 0x8000u) >> 16);
}

// per-frame sum of the gray image (for ImageEnhance.Contrast's mean): partial sums per block, folded by the host-free stage 2
__global__ __launch_bounds__(IM_THREADS) void gray_sum_kernel(const unsigned char* __restrict__ img, unsigned long long* __restrict__ sums,
                                                              long long npix) {
  __shared__ unsigned long long red[IM_THREADS / 64];
  const int f = blockIdx.y;
  unsigned long long s = 0;
  for (long long i = (long long)blockIdx.x * IM_THREADS + threadIdx.x; i < npix; i += (long long)gridDim.x * IM_THREADS) {
    const unsigned char* p = img + ((size_t)f * npix + i) * 3;
    s += gray_of(p[0], p[1], p[2]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&sums[f], red[0] + red[1] + red[2] + red[3]);  // integer sum: order-independent
}

__device__ __forceinline__ unsigned char blend8(int in1, int in2, float alpha, bool inside) {
  const float t = (float)in1 + alpha * (float)(in2 - in1);
  if (inside) return (unsigned char)(int)t;
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (unsigned char)(int)t);
}

// mode 0 grayscale (RandomGrayscale), 1 brightness, 2 contrast (needs gray_sums), 3 saturation, 4 hue (shift in `ishift`)
__global__ __launch_bounds__(IM_THREADS) void color_kernel(unsigned char* __restrict__ img, const unsigned long long* __restrict__ gray_sums,
                                                           long long npix, int mode, float factor, int ishift) {
  const long long i = (long long)blockIdx.x * IM_THREADS + threadIdx.x;
  if (i >= npix) return;
  const int f = blockIdx.y;
  unsigned char* p = img + ((size_t)f * npix + i) * 3;
  const int r = p[0], g = p[1], b = p[2];
  const bool inside = factor >= 0.f && factor <= 1.0f;
  if (mode == 0) {
    const unsigned char y = gray_of(r, g, b);
    p[0] = p[1] = p[2] = y;
  } else if (mode == 1) {
    p[0] = blend8(0, r, factor, inside);
    p[1] = blend8(0, g, factor, inside);
    p[2] = blend8(0, b, factor, inside);
  } else if (mode == 2) {
    const int mean = (int)((double)gray_sums[f] / (double)npix + 0.5);
    p[0] = blend8(mean, r, factor, inside);
    p[1] = blend8(mean, g, factor, inside);
    p[2] = blend8(mean, b, factor, inside);
  } else if (mode == 3) {
    const int y = gray_of(r, g, b);
    p[0] = blend8(y, r, factor, inside);
    p[1] = blend8(y, g, factor, inside);
    p[2] = blend8(y, b, factor, inside);
  } else {
    // rgb2hsv (Convert.c): float for the ratios, double for the h / 6 + 1 wrap and the * 255 scalings
    const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
    int uh = 0, us = 0;
    const int uv = maxc;
    if (minc != maxc) {
      const float cr = (float)(maxc - minc);
      const float s = cr / (float)maxc;
      const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
      float h;
      if (r == maxc) h = bc - gc;
      else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
      else h = (float)(4.0 + (double)gc - (double)rc);
      h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
      uh = (int)((double)h * 255.0);
      us = (int)((double)s * 255.0);
      uh = uh < 0 ? 0 : (uh > 255 ? 255 : uh);
      us = us < 0 ? 0 : (us > 255 ? 255 : us);
    }
    const int hh = (uh + ishift) & 255;
    // hsv2rgb
    if (us == 0) {
      p[0] = p[1] = p[2] = (unsigned char)uv;
    } else {
      const float hf = (float)hh * 6.0f / 255.0f;
      const int ii = (int)floorf(hf);
      const float fr = hf - (float)ii;
      const float fs = (float)us / 255.0f, fv = (float)uv;
      const int pp = (int)round((double)(fv * (1.0f - fs)));
      const int qq = (int)round((double)(fv * (1.0f - fs * fr)));
      const int tt_ = (int)round((double)(fv * (1.0f - fs * (1.0f - fr))));
      const unsigned char P = clip8(pp), Q = clip8(qq), T = clip8(tt_), V = (unsigned char)uv;
      switch (ii % 6) {
        case 0: p[0] = V; p[1] = T; p[2] = P; break;
        case 1: p[0] = Q; p[1] = V; p[2] = P; break;
        case 2: p[0] = P; p[1] = V; p[2] = T; break;
        case 3: p[0] = P; p[1] = Q; p[2] = V; break;
        case 4: p[0] = T; p[1] = P; p[2] = V; break;
        default: p[0] = V; p[1] = P; p[2] = Q; break;
      }
    }
  }
}

// BoxBlur.c, one pass along x (dir 0) or y (dir 1), edge pixels replicated:
// out = (ww * sum_{|d| <= radius} in[clamp(x + d)] + fw * (in[clamp(x - radius - 1)] + in[clamp(x + radius + 1)]) + 2^23) >> 24
__global__ __launch_bounds__(IM_THREADS) void box_blur_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, int H,
                                                              int W, int dir, int radius, unsigned ww, unsigned fw) {
  const long long idx = (long long)blockIdx.x * IM_THREADS + threadIdx.x;
  const long long npix = (long long)H * W;
  if (idx >= npix) return;
  const int f = blockIdx.y, y = (int)(idx / W), x = (int)(idx - (long long)y * W);
  const unsigned char* base = in + (size_t)f * npix * 3;
  const int len = dir == 0 ? W : H, pos = dir == 0 ? x : y;
  const size_t step = dir == 0 ? 3 : (size_t)W * 3;
  const unsigned char* line = base + (dir == 0 ? (size_t)y * W * 3 : (size_t)x * 3);
  unsigned long long a0 = 0, a1 = 0, a2 = 0;
  for (int d = -radius; d <= radius; ++d) {
    int q = pos + d;
    q = q < 0 ? 0 : (q > len - 1 ? len - 1 : q);
    const unsigned char* p = line + (size_t)q * step;
    a0 += p[0];
    a1 += p[1];
    a2 += p[2];
  }
  int ql = pos - radius - 1, qr = pos + radius + 1;
  ql = ql < 0 ? 0 : ql;
  qr = qr > len - 1 ? len - 1 : qr;
  const unsigned char *pl = line + (size_t)ql * step, *pr = line + (size_t)qr * step;
  unsigned char* o = out + ((size_t)f * npix + idx) * 3;
  o[0] = (unsigned char)((a0 * ww + (unsigned long long)(pl[0] + pr[0]) * fw + (1ull << 23)) >> 24);
  o[1] = (unsigned char)((a1 * ww + (unsigned long long)(pl[1] + pr[1]) * fw + (1ull << 23)) >> 24);
  o[2] = (unsigned char)((a2 * ww + (unsigned long long)(pl[2] + pr[2]) * fw + (1ull << 23)) >> 24);
}

}  // namespace tt

using namespace tt;

static dim3 pix_grid(long long npix, int F) { return dim3((unsigned)((npix + IM_THREADS - 1) / IM_THREADS), (unsigned)F); }

extern "C" int tt_img_resample_h(const unsigned char* in, unsigned char* out, const int* coeffs, const int* bounds, int F, int H, int W, int y0,
                                 int x0, int h, int OW, int ksize, tt_stream_t stream) {
  TT_REQUIRE(in && out && coeffs && bounds && F > 0 && H > 0 && W > 0 && h > 0 && OW > 0 && ksize > 0, "img_resample_h: bad arguments");
  TT_REQUIRE(y0 >= 0 && x0 >= 0 && y0 + h <= H && x0 < W, "img_resample_h: crop outside the image");
  hipLaunchKernelGGL(resample_h_kernel, pix_grid((long long)h * OW, F), dim3(IM_THREADS), 0, as_stream(stream), in, out, coeffs, bounds, H, W, y0,
                     x0, h, OW, ksize);
  TT_CHECK_LAUNCH("img_resample_h");
  return TT_OK;
}

extern "C" int tt_img_resample_v(const unsigned char* in, unsigned char* out_u8, float* out_f32, const int* coeffs, const int* bounds, int F,
                                 int Hin, int W, int y0, int OH, int ksize, int flip, const float* mean3, const float* std3,
                                 tt_stream_t stream) {
  TT_REQUIRE(in && coeffs && bounds && (out_u8 != nullptr) != (out_f32 != nullptr), "img_resample_v: exactly one output buffer");
  TT_REQUIRE(F > 0 && Hin > 0 && W > 0 && OH > 0 && ksize > 0 && y0 >= 0 && y0 < Hin, "img_resample_v: bad arguments");
  TT_REQUIRE(out_u8 || (mean3 && std3), "img_resample_v: the float output needs mean and std (host pointers to 3 floats)");
  const float m0 = mean3 ? mean3[0] : 0.f, m1 = mean3 ? mean3[1] : 0.f, m2 = mean3 ? mean3[2] : 0.f;
  const float s0 = std3 ? std3[0] : 1.f, s1 = std3 ? std3[1] : 1.f, s2 = std3 ? std3[2] : 1.f;
  hipLaunchKernelGGL(resample_v_kernel, pix_grid((long long)OH * W, F), dim3(IM_THREADS), 0, as_stream(stream), in, out_u8, out_f32, coeffs,
                     bounds, Hin, W, y0, OH, ksize, flip, m0, m1, m2, s0, s1, s2);
  TT_CHECK_LAUNCH("img_resample_v");
  return TT_OK;
}

extern "C" int tt_img_color(unsigned char* img, int F, int H, int W, int mode, float factor, int hue_shift, unsigned long long* gray_sums,
                            tt_stream_t stream) {
  TT_REQUIRE(img && F > 0 && H > 0 && W > 0 && mode >= 0 && mode <= 4, "img_color: bad arguments");
  hipStream_t s = as_stream(stream);
  const long long npix = (long long)H * W;
  if (mode == 2) {
    TT_REQUIRE(gray_sums, "img_color: contrast needs a gray_sums workspace of F uint64");
    if (hipMemsetAsync(gray_sums, 0, sizeof(unsigned long long) * F, s) != hipSuccess) {
      set_error("img_color: memset failed");
      return TT_ELAUNCH;
    }
    long long blocks = (npix + IM_THREADS * 8 - 1) / (IM_THREADS * 8);
    blocks = blocks > 256 ? 256 : (blocks < 1 ? 1 : blocks);
    hipLaunchKernelGGL(gray_sum_kernel, dim3((unsigned)blocks, F), dim3(IM_THREADS), 0, s, img, gray_sums, npix);
  }
  hipLaunchKernelGGL(color_kernel, pix_grid(npix, F), dim3(IM_THREADS), 0, s, img, gray_sums, npix, mode, factor, hue_shift & 255);
  TT_CHECK_LAUNCH("img_color");
  return TT_OK;
}

extern "C" int tt_img_box_blur(const unsigned char* in, unsigned char* out, int F, int H, int W, int direction, int radius, unsigned ww,
                               unsigned fw, tt_stream_t stream) {
  TT_REQUIRE(in && out && in != out && F > 0 && H > 0 && W > 0 && radius >= 0 && (direction == 0 || direction == 1), "img_box_blur: bad arguments");
  hipLaunchKernelGGL(box_blur_kernel, pix_grid((long long)H * W, F), dim3(IM_THREADS), 0, as_stream(stream), in, out, H, W, direction, radius, ww,
                     fw);
  TT_CHECK_LAUNCH("img_box_blur");
  return TT_OK;
}
