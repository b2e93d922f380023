// Memory-bound helper kernels of the image encoder and the glue around the GEMMs.
#include <algorithm>

#include "kernels.h"

namespace {

// ------------------------------------------------------------------ LayerNorm
// One wave per row, the row held in registers (VPL float4 per lane): a single global read, two wave
// reductions (mean, then centred variance - the two-pass form PyTorch uses), packed 8-B f16 stores.
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                            const float* __restrict__ b, float eps, int M, int C,
                                                            half_t* y16, int ldy16, float* y32, int ldy32, int act, size_t lo_off) {
  const int lane = threadIdx.x & 63;
  const int nv = C >> 2;                                   // float4 per row
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  f32x4 wv[VPL], bv[VPL];
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int v = lane + 64 * k;
    if (v < nv) {
      wv[k] = *reinterpret_cast<const f32x4*>(w + 4 * v);
      bv[k] = *reinterpret_cast<const f32x4*>(b + 4 * v);
    }
  }
  for (int row = wave_global; row < M; row += nwaves) {
    const float* xr = x + (size_t)row * ldx;
    f32x4 xv[VPL];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int v = lane + 64 * k;
      if (v < nv) {
        xv[k] = *reinterpret_cast<const f32x4*>(xr + 4 * v);
        s += (xv[k][0] + xv[k][1]) + (xv[k][2] + xv[k][3]);
      }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int v = lane + 64 * k;
      if (v < nv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = xv[k][e] - mean;
          q += d * d;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const int v = lane + 64 * k;
      if (v < nv) {
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = (xv[k][e] - mean) * rstd * wv[k][e] + bv[k][e];
          if (act == 1) t = gelu_erf(t);
          y[e] = t;
        }
        if (y16) store_h4(y16 + (size_t)row * ldy16 + 4 * v, lo_off, y);
        if (y32) *reinterpret_cast<f32x4*>(y32 + (size_t)row * ldy32 + 4 * v) = y;
      }
    }
  }
}

// generic fallback (any C / alignment)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                        const float* __restrict__ b, float eps, int M, int C,
                                                        half_t* y16, int ldy16, float* y32, int ldy32, int act, size_t lo_off) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (size_t)row * ldx;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  const float mean = wave_sum(s) / (float)C;
  float v = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float d = xr[c] - mean;
    v += d * d;
  }
  const float rstd = rsqrtf(wave_sum(v) / (float)C + eps);
  for (int c = lane; c < C; c += 64) {
    float y = (xr[c] - mean) * rstd * w[c] + b[c];
    if (act == 1) y = gelu_erf(y);
    if (y16) store_h1(y16 + (size_t)row * ldy16 + c, lo_off, y);
    if (y32) y32[(size_t)row * ldy32 + c] = y;
  }
}

// y = f16/f32(a + sb * b): 4 elements per thread when everything is 4-aligned
__global__ void cast_add_vec_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, int bmod,
                                    float sb, int M, int C4, half_t* y16, int ldy16, float* y32, int ldy32, size_t lo_off) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * C4) return;
  const int m = (int)(i / C4), c = (int)(i % C4) * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(a + (size_t)m * lda + c);
  if (b) {
    const f32x4 u = *reinterpret_cast<const f32x4*>(b + (size_t)(bmod ? m % bmod : m) * ldb + c);
    v[0] += sb * u[0]; v[1] += sb * u[1]; v[2] += sb * u[2]; v[3] += sb * u[3];
  }
  if (y16) store_h4(y16 + (size_t)m * ldy16 + c, lo_off, v);
  if (y32) *reinterpret_cast<f32x4*>(y32 + (size_t)m * ldy32 + c) = v;
}

// y16 = f16(a), z16 = f16(a + b[m % bmod]) in one pass over a (the two image-side operands of the decoder's token <-> image attentions)
__global__ void cast_pair_kernel(const float* __restrict__ a, const float* __restrict__ b, int bmod, int M, int C4, half_t* y16, half_t* z16,
                                 size_t lo_off) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * C4) return;
  const int m = (int)(i / C4), c = (int)(i % C4) * 4;
  const size_t C = (size_t)C4 * 4;
  const f32x4 v = *reinterpret_cast<const f32x4*>(a + (size_t)m * C + c);
  const f32x4 u = *reinterpret_cast<const f32x4*>(b + (size_t)(bmod ? m % bmod : m) * C + c);
  store_h4(y16 + (size_t)m * C + c, lo_off, v);
  const f32x4 w = {v[0] + u[0], v[1] + u[1], v[2] + u[2], v[3] + u[3]};
  store_h4(z16 + (size_t)m * C + c, lo_off, w);
}

__global__ void cast_add_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, int bmod,
                                float sb, int M, int C, half_t* y16, int ldy16, float* y32, int ldy32, size_t lo_off) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * C) return;
  const int m = (int)(i / C), c = (int)(i % C);
  float v = a[(size_t)m * lda + c];
  if (b) v += sb * b[(size_t)(bmod ? m % bmod : m) * ldb + c];
  if (y16) store_h1(y16 + (size_t)m * ldy16 + c, lo_off, v);
  if (y32) y32[(size_t)m * ldy32 + c] = v;
}

// ------------------------------------------------------------------ patch-embed im2col
// one thread = one (token, 8-wide k chunk); 20 chunks per token (K padded 147 -> 160)
// U8: img is uint8 [B, S, S, 3] (decoded HWC frames); the /255, -mean, /std of load_video_frames (utils/misc.py:270-276)
// is applied here in f32, in that order, so the f16 operand is bit-identical to the one built from a normalised f32 frame
template <bool U8>
__global__ void im2col_patch_kernel(const void* __restrict__ img_, int B, int S, half_t* __restrict__ A, size_t lo_off, int row_major) {
  const float* img = static_cast<const float*>(img_);
  const uint8_t* img8 = static_cast<const uint8_t*>(img_);
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  const int G = S / 4;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * G * G * 20;
  if (i >= total) return;
  const int ch = (int)(i % 20);
  const size_t tok = i / 20;
  const int b = (int)(tok / ((size_t)G * G));
  const int t = (int)(tok % ((size_t)G * G));
  // window-major (w = 8) token -> (y, x)
  const int win = t >> 6, in = t & 63;
  const int wpr = G / 8;
  const int y = row_major ? t / G : (win / wpr) * 8 + (in >> 3), x = row_major ? t % G : (win % wpr) * 8 + (in & 7);
  half8 out, out_lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = ch * 8 + j;
    float v = 0.f;
    if (k < 147) {
      const int c = k / 49, r = k % 49, ky = r / 7, kx = r % 7;
      const int iy = y * 4 - 3 + ky, ix = x * 4 - 3 + kx;
      if (iy >= 0 && iy < S && ix >= 0 && ix < S) {
        if (U8) v = ((float)img8[(((size_t)b * S + iy) * S + ix) * 3 + c] / 255.0f - mean[c]) / stdv[c];
        else v = img[(((size_t)b * 3 + c) * S + iy) * S + ix];
      }
    }
    out[j] = (half_t)v;
    out_lo[j] = split_lo(v, out[j]);
  }
  *reinterpret_cast<half8*>(A + tok * 160 + ch * 8) = out;
  if (lo_off) *reinterpret_cast<half8*>(A + lo_off + tok * 160 + ch * 8) = out_lo;
}

// ------------------------------------------------------------------ 2x2 max-pool inside windows
template <typename T>
__global__ void pool_tokens_kernel(const T* __restrict__ in, int ldin, T* __restrict__ out, int ldout, int nwin, int w, int C) {
  const int hw = w / 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)nwin * hw * hw * C;
  if (i >= total) return;
  const int c = (int)(i % C);
  const size_t ot = i / C;
  const int win = (int)(ot / (hw * hw)), p = (int)(ot % (hw * hw));
  const int py = p / hw, px = p % hw;
  const T* base = in + ((size_t)win * w * w) * ldin + c;
  float m = -3.0e38f;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) m = fmaxf(m, (float)base[(size_t)((2 * py + dy) * w + 2 * px + dx) * ldin]);
  out[ot * ldout + c] = (T)m;
}

// the same on a 2-term f16 split (hi plane + lo plane lo_off elements behind it, common.h): the maximum of the four VALUES
// hi + lo * 2^-11, written as the (hi, lo) pair of the token that holds it (exact: a max-pool selects, it does not compute)
// one thread = 8 consecutive channels (16-B loads / stores of both planes; C % 8 == 0, ldin / ldout / plane offsets multiples of 8)
__global__ void pool_tokens_split_kernel(const half_t* __restrict__ in, size_t in_lo, int ldin, half_t* __restrict__ out, size_t out_lo, int ldout,
                                         int nwin, int w, int C) {
  const int hw = w / 2, C8 = C >> 3;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)nwin * hw * hw * C8;
  if (i >= total) return;
  const int c = (int)(i % C8) * 8;
  const size_t ot = i / C8;
  const int win = (int)(ot / (hw * hw)), p = (int)(ot % (hw * hw));
  const int py = p / hw, px = p % hw;
  const half_t* base = in + ((size_t)win * w * w) * ldin + c;
  float m[8];
  half8 mh, ml;
#pragma unroll
  for (int e = 0; e < 8; ++e) { m[e] = -3.0e38f; mh[e] = (half_t)0.f; ml[e] = (half_t)0.f; }
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const half_t* q = base + (size_t)((2 * py + dy) * w + 2 * px + dx) * ldin;
      const half8 h = *reinterpret_cast<const half8*>(q), l = *reinterpret_cast<const half8*>(q + in_lo);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = (float)h[e] + (float)l[e] * SPLIT_INV;
        if (v > m[e]) { m[e] = v; mh[e] = h[e]; ml[e] = l[e]; }
      }
    }
  *reinterpret_cast<half8*>(out + ot * ldout + c) = mh;
  *reinterpret_cast<half8*>(out + ot * ldout + c + out_lo) = ml;
}

// [R, 64] f16 row-major -> [64, ld] (column r = input row r): 64 x 64 tiles through LDS.  The memory tokens as the V^T operand of the
// d = 256 flash kernel (DV = 64, attn_flash256.hip)
__global__ __launch_bounds__(256) void transpose_rows64_f16_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, int R, int ld) {
  __shared__ half_t t[64][66];
  const int r0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    t[r][c] = (r0 + r < R) ? in[(size_t)(r0 + r) * 64 + c] : (half_t)0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < ld) out[(size_t)c * ld + r0 + r] = t[r][c];
  }
}

// ------------------------------------------------------------------ token re-ordering between window sizes
// 4 channels per thread (C % 4 == 0); frame b of the output goes to dst.p[b] (the feature-cache slots of the frames are not
// contiguous), or to out + b * H * W * C when dst.p[0] is null.
__global__ void permute_tokens_kernel(const float* __restrict__ in, float* __restrict__ out, PermuteDst dst, int B, int H, int W, int C,
                                      int w_in, int w_out, const float* __restrict__ up, int w_up) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int C4 = C >> 2;
  const size_t total = (size_t)B * H * W * C4;
  if (i >= total) return;
  const int c = (int)(i % C4) * 4;
  const size_t ot = i / C4;
  const int b = (int)(ot / ((size_t)H * W));
  const int t = (int)(ot % ((size_t)H * W));
  // destination token t (window size w_out) -> (y, x)
  const int ww = w_out * w_out, wpr = W / w_out;
  const int win = t / ww, in_w = t % ww;
  const int y = (win / wpr) * w_out + in_w / w_out, x = (win % wpr) * w_out + in_w % w_out;
  f32x4 v = *reinterpret_cast<const f32x4*>(in + ((size_t)b * H * W + tok_of_yx(y, x, W, w_in)) * C + c);
  if (up) {
    const f32x4 u = *reinterpret_cast<const f32x4*>(up + ((size_t)b * (H / 2) * (W / 2) + tok_of_yx(y / 2, x / 2, W / 2, w_up)) * C + c);
    v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
  }
  float* ob = dst.p[0] ? dst.p[b] : out + (size_t)b * H * W * C;
  *reinterpret_cast<f32x4*>(ob + (size_t)t * C + c) = v;
}

// ------------------------------------------------------------------ batched transpose through LDS
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const float* ib = in + (size_t)b * R * Cc;
  float* ob = out + (size_t)b * R * Cc;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    if (r < R && c < Cc) tile[j][tx] = ib[(size_t)r * Cc + c];
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (r < R && c < Cc) ob[(size_t)c * R + r] = tile[tx][j];
  }
}

__global__ void add_rowvec_kernel(float* x, int ld, const float* __restrict__ v, int M, int C, const float* flag) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * C) return;
  const float f = flag ? ((flag[0] > 0.f) ? 0.f : 1.f) : 1.f;
  const int m = (int)(i / C), c = (int)(i % C);
  x[(size_t)m * ld + c] += f * v[c];
}

__global__ void round_bf16_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float f = in[i];
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7F800000u) != 0x7F800000u) {           // finite: round to nearest even on bit 16
    u += 0x7FFFu + ((u >> 16) & 1u);
    u &= 0xFFFF0000u;
  }
  out[i] = __uint_as_float(u);
}

// out = bf16_round(x + [flag <= 0] * v[c]): the "+ (1 - appearing) * no_obj_embed_spatial" of _encode_new_memory and the bf16 storage of
// the memory bank (sam2_base_official.py:1018-1024, sam2_video_predictor_official.py:887) in one launch; x is left as it is
__global__ void add_rowvec_round_bf16_kernel(const float* __restrict__ x, const float* __restrict__ v, int C, const float* __restrict__ flag,
                                             float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = (flag[0] > 0.f) ? 0.f : 1.f;
  const float f = x[i] + g * v[i % C];
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7F800000u) != 0x7F800000u) {
    u += 0x7FFFu + ((u >> 16) & 1u);
    u &= 0xFFFF0000u;
  }
  out[i] = __uint_as_float(u);
}

__global__ void fill_f32_kernel(float* p, float v, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

inline dim3 grid1d(size_t n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }
}  // namespace

hipError_t layernorm_launch(const float* x, int ldx, const float* w, const float* b, float eps, int M, int C, half_t* y16,
                            int ldy16, float* y32, int ldy32, int act, hipStream_t s, size_t lo_off) {
  const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && (!y16 || ldy16 % 4 == 0) && (!y32 || ldy32 % 4 == 0) && C <= 1280 &&
                   ((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0) && ((uintptr_t)b % 16 == 0);
  if (vec) {
    const int nv = C / 4, vpl = (nv + 63) / 64;
    const int blocks = std::min((M + 3) / 4, 256 * 16);
#define LN_LAUNCH(V) layernorm_vec_kernel<V><<<dim3(blocks), dim3(256), 0, s>>>(x, ldx, w, b, eps, M, C, y16, ldy16, y32, ldy32, act, lo_off)
    if (vpl <= 1) LN_LAUNCH(1);
    else if (vpl == 2) LN_LAUNCH(2);
    else if (vpl == 3) LN_LAUNCH(3);
    else LN_LAUNCH(5);
#undef LN_LAUNCH
  } else {
    layernorm_kernel<<<dim3((M + 3) / 4), dim3(256), 0, s>>>(x, ldx, w, b, eps, M, C, y16, ldy16, y32, ldy32, act, lo_off);
  }
  return hipGetLastError();
}
hipError_t cast_add_launch(const float* a, int lda, const float* b, int ldb, int bmod, float sb, int M, int C, half_t* y16,
                           int ldy16, float* y32, int ldy32, hipStream_t s, size_t lo_off) {
  const bool vec = (C % 4 == 0) && (lda % 4 == 0) && (!b || ldb % 4 == 0) && (!y16 || ldy16 % 4 == 0) && (!y32 || ldy32 % 4 == 0) &&
                   ((uintptr_t)a % 16 == 0) && (!b || (uintptr_t)b % 16 == 0) && (!y16 || (uintptr_t)y16 % 8 == 0) &&
                   (!y32 || (uintptr_t)y32 % 16 == 0);
  if (vec)
    cast_add_vec_kernel<<<grid1d((size_t)M * (C / 4)), dim3(256), 0, s>>>(a, lda, b, ldb, bmod, sb, M, C / 4, y16, ldy16, y32, ldy32, lo_off);
  else
    cast_add_kernel<<<grid1d((size_t)M * C), dim3(256), 0, s>>>(a, lda, b, ldb, bmod, sb, M, C, y16, ldy16, y32, ldy32, lo_off);
  return hipGetLastError();
}
hipError_t cast_pair_launch(const float* a, const float* b, int bmod, int M, int C, half_t* y16, half_t* z16, hipStream_t s, size_t lo_off) {
  if ((C & 3) || !a || !b || !y16 || !z16) return hipErrorInvalidValue;
  cast_pair_kernel<<<grid1d((size_t)M * (C / 4)), dim3(256), 0, s>>>(a, b, bmod, M, C / 4, y16, z16, lo_off);
  return hipGetLastError();
}
hipError_t im2col_patch_launch(const float* img, int B, int S, half_t* A, hipStream_t s, size_t lo_off, int row_major) {
  const size_t total = (size_t)B * (S / 4) * (S / 4) * 20;
  im2col_patch_kernel<false><<<grid1d(total), dim3(256), 0, s>>>(img, B, S, A, lo_off, row_major);
  return hipGetLastError();
}
hipError_t im2col_patch_u8_launch(const uint8_t* img_hwc, int B, int S, half_t* A, hipStream_t s, size_t lo_off, int row_major) {
  const size_t total = (size_t)B * (S / 4) * (S / 4) * 20;
  im2col_patch_kernel<true><<<grid1d(total), dim3(256), 0, s>>>(img_hwc, B, S, A, lo_off, row_major);
  return hipGetLastError();
}
hipError_t pool_tokens_f32_launch(const float* in, int ldin, float* out, int ldout, int nwin, int w, int C, hipStream_t s) {
  pool_tokens_kernel<float><<<grid1d((size_t)nwin * (w / 2) * (w / 2) * C), dim3(256), 0, s>>>(in, ldin, out, ldout, nwin, w, C);
  return hipGetLastError();
}
hipError_t pool_tokens_f16_launch(const half_t* in, int ldin, half_t* out, int ldout, int nwin, int w, int C, hipStream_t s) {
  pool_tokens_kernel<half_t><<<grid1d((size_t)nwin * (w / 2) * (w / 2) * C), dim3(256), 0, s>>>(in, ldin, out, ldout, nwin, w, C);
  return hipGetLastError();
}
hipError_t pool_tokens_split_launch(const half_t* in, size_t in_lo, int ldin, half_t* out, size_t out_lo, int ldout, int nwin, int w, int C, hipStream_t s) {
  if ((C & 7) || (ldin & 7) || (ldout & 7) || (in_lo & 7) || (out_lo & 7)) return hipErrorInvalidValue;
  pool_tokens_split_kernel<<<grid1d((size_t)nwin * (w / 2) * (w / 2) * (C / 8)), dim3(256), 0, s>>>(in, in_lo, ldin, out, out_lo, ldout, nwin, w, C);
  return hipGetLastError();
}
hipError_t transpose_rows64_f16_launch(const half_t* in, half_t* out, int R, int ld, hipStream_t s) {
  if (R <= 0 || ld < R) return hipErrorInvalidValue;
  transpose_rows64_f16_kernel<<<dim3((ld + 63) / 64), dim3(256), 0, s>>>(in, out, R, ld);
  return hipGetLastError();
}
hipError_t permute_tokens_launch(const float* in, float* out, int B, int H, int W, int C, int w_in, int w_out,
                                 const float* up, int w_up, hipStream_t s, float* const* dst) {
  if ((C & 3) || (dst && B > PermuteDst::MAX_B)) return hipErrorInvalidValue;
  PermuteDst d;
  for (int b = 0; b < PermuteDst::MAX_B; ++b) d.p[b] = (dst && b < B) ? dst[b] : nullptr;
  permute_tokens_kernel<<<grid1d((size_t)B * H * W * (C / 4)), dim3(256), 0, s>>>(in, out, d, B, H, W, C, w_in, w_out, up, w_up);
  return hipGetLastError();
}
hipError_t transpose_f32_launch(const float* in, float* out, int batch, int R, int Cc, hipStream_t s) {
  transpose_f32_kernel<<<dim3((Cc + 31) / 32, (R + 31) / 32, batch), dim3(256), 0, s>>>(in, out, R, Cc);
  return hipGetLastError();
}
hipError_t add_rowvec_launch(float* x, int ld, const float* v, int M, int C, const float* flag, hipStream_t s) {
  add_rowvec_kernel<<<grid1d((size_t)M * C), dim3(256), 0, s>>>(x, ld, v, M, C, flag);
  return hipGetLastError();
}
hipError_t round_bf16_launch(const float* in, float* out, size_t n, hipStream_t s) {
  round_bf16_kernel<<<grid1d(n), dim3(256), 0, s>>>(in, out, n);
  return hipGetLastError();
}
hipError_t add_rowvec_round_bf16_launch(const float* x, const float* v, int M, int C, const float* flag, float* out, hipStream_t s) {
  const size_t n = (size_t)M * C;
  add_rowvec_round_bf16_kernel<<<grid1d(n), dim3(256), 0, s>>>(x, v, C, flag, out, n);
  return hipGetLastError();
}
hipError_t fill_f32_launch(float* p, float v, size_t n, hipStream_t s) {
  fill_f32_kernel<<<grid1d(n), dim3(256), 0, s>>>(p, v, n);
  return hipGetLastError();
}
