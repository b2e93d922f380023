// Memory-encoder convolution kernels (MaskDownSampler / CXBlock of
// /root/reference/sam2/sam2/modeling/memory_encoder.py:19-119) on NHWC ("token-major") tensors.
#include "kernels.h"

namespace {

// mask_for_mem at pixel (y, x) of the 1024^2 grid (sam2_base_official.py:454-459,:1000-1010): bilinear x4 of the 256^2 low-res
// logits (align_corners=False, F.interpolate semantics), then (binarize ? mask > 0 : sigmoid) * scale + bias
static __device__ __forceinline__ float mask_prep_value(const float* __restrict__ low, int y, int x, int binarize, float scale, float bias) {
  float sy = (y + 0.5f) * 0.25f - 0.5f, sx = (x + 0.5f) * 0.25f - 0.5f;
  sy = fmaxf(sy, 0.f);
  sx = fmaxf(sx, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = min(y0 + 1, 255), x1 = min(x0 + 1, 255);
  const float ly = sy - y0, lx = sx - x0;
  const float v = (1.f - ly) * ((1.f - lx) * low[y0 * 256 + x0] + lx * low[y0 * 256 + x1]) +
                  ly * ((1.f - lx) * low[y1 * 256 + x0] + lx * low[y1 * 256 + x1]);
  const float m = binarize ? (v > 0.f ? 1.f : 0.f) : 1.f / (1.f + expf(-v));
  return m * scale + bias;
}
__global__ void mask_prep_kernel(const float* __restrict__ low, float* __restrict__ out, int binarize, float scale, float bias) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 1024 * 1024) return;
  out[i] = mask_prep_value(low, i >> 10, i & 1023, binarize, scale, bias);
}

// one thread = one output pixel, all COUT channels in registers; weights (COUT, CIN, 3, 3) staged in LDS.
// PREP (CIN == 1, Hin == 1024): `in` is the 256^2 low-res mask and the 1024^2 input pixel is computed on the fly
// (mask_prep_value), so the 4-MB mask_for_mem tensor is never written or read.
template <int CIN, int COUT, bool PREP = false>
__global__ __launch_bounds__(128) void conv3x3s2_ln_gelu_kernel(const float* __restrict__ in, int Hin, const float* __restrict__ w,
                                                                 const float* __restrict__ b, const float* __restrict__ lnw,
                                                                 const float* __restrict__ lnb, float* out32, half_t* out16, size_t lo_off,
                                                                 int binarize = 0, float scale = 0.f, float bias = 0.f) {
  __shared__ float sw[COUT * CIN * 9];
  for (int i = threadIdx.x; i < COUT * CIN * 9; i += blockDim.x) sw[i] = w[i];
  __syncthreads();
  const int Hout = Hin / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Hout * Hout) return;
  const int oy = i / Hout, ox = i % Hout;
  float acc[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) acc[o] = b[o];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * 2 - 1 + ky;
    if (iy < 0 || iy >= Hin) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox * 2 - 1 + kx;
      if (ix < 0 || ix >= Hin) continue;
      const float* ip = in + ((size_t)iy * Hin + ix) * CIN;
#pragma unroll 4
      for (int c = 0; c < CIN; ++c) {
        const float v = PREP ? mask_prep_value(in, iy, ix, binarize, scale, bias) : ip[c];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] += v * sw[(o * CIN + c) * 9 + ky * 3 + kx];
      }
    }
  }
  float mean = 0.f;
#pragma unroll
  for (int o = 0; o < COUT; ++o) mean += acc[o];
  mean /= COUT;
  float var = 0.f;
#pragma unroll
  for (int o = 0; o < COUT; ++o) var += (acc[o] - mean) * (acc[o] - mean);
  const float rstd = 1.f / sqrtf(var / COUT + 1e-6f);
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
    const float y = gelu_erf((acc[o] - mean) * rstd * lnw[o] + lnb[o]);
    if (out32) out32[(size_t)i * COUT + o] = y;
    if (out16) store_h1(out16 + (size_t)i * COUT + o, lo_off, y);
  }
}

__global__ void im2col3x3s2_kernel(const half_t* __restrict__ in, int Hin, int CIN, half_t* __restrict__ A, size_t lo_off) {
  const int Hout = Hin / 2;
  const int cpr = 9 * CIN / 8;                       // 8-wide chunks per output row
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)Hout * Hout * cpr) return;
  const int ch = (int)(i % cpr);
  const int pix = (int)(i / cpr);
  const int oy = pix / Hout, ox = pix % Hout;
  const int k = ch * 8, tap = k / CIN, c = k % CIN;  // CIN % 8 == 0: a chunk never straddles taps
  const int iy = oy * 2 - 1 + tap / 3, ix = ox * 2 - 1 + tap % 3;
  half8 v, vl;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = vl[j] = (half_t)0.f;
  const bool inside = iy >= 0 && iy < Hin && ix >= 0 && ix < Hin;
  if (inside) v = *reinterpret_cast<const half8*>(in + ((size_t)iy * Hin + ix) * CIN + c);
  *reinterpret_cast<half8*>(A + (size_t)pix * 9 * CIN + k) = v;
  if (lo_off) {                                        // split-f16 mode: the lo plane is gathered the same way
    if (inside) vl = *reinterpret_cast<const half8*>(in + lo_off + ((size_t)iy * Hin + ix) * CIN + c);
    *reinterpret_cast<half8*>(A + lo_off + (size_t)pix * 9 * CIN + k) = vl;
  }
}

// depth-wise 7x7: one workgroup = an 8x8 pixel tile x 64 channels; the 14x14 halo patch is staged in LDS
// (channel innermost, so global reads are 256-B rows and LDS reads are conflict-free across lanes = channels).
// A thread owns one channel and two output rows: its 49 weights sit in registers, and every halo row it touches is read ONCE
// from LDS (14 values) for the 8 outputs x 7 taps that use it - 196 LDS reads per thread instead of 2 per multiply-add.
__global__ __launch_bounds__(256) void dwconv7_kernel(const float* __restrict__ in, int H, int C, const float* __restrict__ w,
                                                      const float* __restrict__ b, float* __restrict__ out) {
  __shared__ float tile[14 * 14 * 64];
  const int c0 = blockIdx.z * 64, ty0 = blockIdx.y * 8, tx0 = blockIdx.x * 8;
  const int lane_c = threadIdx.x & 63, grp = threadIdx.x >> 6;         // 4 groups of output rows
  float wr[49];
#pragma unroll
  for (int k = 0; k < 49; ++k) wr[k] = w[(size_t)(c0 + lane_c) * 49 + k];
  const float bias = b[c0 + lane_c];
  for (int i = threadIdx.x; i < 14 * 14 * 64; i += 256) {
    const int c = i & 63, p = i >> 6, py = p / 14, px = p % 14;
    const int iy = ty0 - 3 + py, ix = tx0 - 3 + px;
    tile[i] = (iy >= 0 && iy < H && ix >= 0 && ix < H) ? in[((size_t)iy * H + ix) * C + c0 + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int oy = grp + 4 * half;
    float acc[8];
#pragma unroll
    for (int ox = 0; ox < 8; ++ox) acc[ox] = bias;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      float row[14];
#pragma unroll
      for (int x = 0; x < 14; ++x) row[x] = tile[((oy + ky) * 14 + x) * 64 + lane_c];
#pragma unroll
      for (int kx = 0; kx < 7; ++kx)
#pragma unroll
        for (int ox = 0; ox < 8; ++ox) acc[ox] = fmaf(row[ox + kx], wr[ky * 7 + kx], acc[ox]);
    }
#pragma unroll
    for (int ox = 0; ox < 8; ++ox) out[((size_t)(ty0 + oy) * H + tx0 + ox) * C + c0 + lane_c] = acc[ox];
  }
}
}  // namespace

hipError_t mask_prep_launch(const float* low, float* out, int binarize, float scale, float bias, hipStream_t s) {
  mask_prep_kernel<<<dim3(4096), dim3(256), 0, s>>>(low, out, binarize, scale, bias);
  return hipGetLastError();
}

hipError_t conv3x3s2_ln_gelu_launch(const float* in, int Hin, int CIN, int COUT, const float* w, const float* b,
                                    const float* lnw, const float* lnb, float* out32, half_t* out16, hipStream_t s, size_t lo_off) {
  const int n = (Hin / 2) * (Hin / 2);
  const dim3 grid((n + 127) / 128), block(128);
  if (CIN == 1 && COUT == 4)
    conv3x3s2_ln_gelu_kernel<1, 4><<<grid, block, 0, s>>>(in, Hin, w, b, lnw, lnb, out32, out16, lo_off);
  else if (CIN == 4 && COUT == 16)
    conv3x3s2_ln_gelu_kernel<4, 16><<<grid, block, 0, s>>>(in, Hin, w, b, lnw, lnb, out32, out16, lo_off);
  else if (CIN == 16 && COUT == 64)
    conv3x3s2_ln_gelu_kernel<16, 64><<<grid, block, 0, s>>>(in, Hin, w, b, lnw, lnb, out32, out16, lo_off);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t conv3x3s2_ln_gelu_from_low_launch(const float* low256, int binarize, float scale, float bias, const float* w, const float* b,
                                             const float* lnw, const float* lnb, float* out32, hipStream_t s) {
  const int n = 512 * 512;
  conv3x3s2_ln_gelu_kernel<1, 4, true><<<dim3((n + 127) / 128), dim3(128), 0, s>>>(low256, 1024, w, b, lnw, lnb, out32, nullptr, 0, binarize, scale, bias);
  return hipGetLastError();
}

hipError_t im2col3x3s2_launch(const half_t* in, int Hin, int CIN, half_t* A, hipStream_t s, size_t lo_off) {
  if (CIN % 8) return hipErrorInvalidValue;
  const size_t total = (size_t)(Hin / 2) * (Hin / 2) * (9 * CIN / 8);
  im2col3x3s2_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(in, Hin, CIN, A, lo_off);
  return hipGetLastError();
}

hipError_t dwconv7_launch(const float* in, int H, int C, const float* w, const float* b, float* out, hipStream_t s) {
  if (H % 8 || C % 64) return hipErrorInvalidValue;
  dwconv7_kernel<<<dim3(H / 8, H / 8, C / 64), dim3(256), 0, s>>>(in, H, C, w, b, out);
  return hipGetLastError();
}
