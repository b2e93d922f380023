// Common types and helpers for the gfx950 (MI355X / CDNA4) SAM2 kernels.
// Wavefront = 64 lanes; MFMA fragments follow the gfx950 32x32x16 f16 maps:
//   A: lane l holds A[row l&31][k = 8*(l>>5) + j], j = 0..7
//   B: lane l holds B[k = 8*(l>>5) + j][col l&31]
//   C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5), reg in [0,16)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WAVE 64

static __device__ __forceinline__ f32x16 mfma32(half8 a, half8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// row of accumulator register `reg` for this lane inside a 32x32 tile
static __device__ __forceinline__ int acc_row(int reg, int lane) {
  return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

static __device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// Branch-free erf-GELU for the GEMM epilogue: GELU(x) = relu(x) - |x| * q(|x|), q = 0.5 * erfc(|x| / sqrt 2) in the
// Abramowitz-Stegun 7.1.26 form (t = 1 / (1 + p z), 5-term polynomial in t times exp(-z^2)).  |error| <= 3.5e-7 absolute
// over all x (erff: 1 ulp) at 12 VALU + v_rcp_f32 + v_exp_f32, about a third of the issue cycles of the libm erff path
// (two divergent branches, both executed by a wave that holds mixed |x|).
static __device__ __forceinline__ float gelu_erf_fast(float x) {
  const float ax = fminf(fabsf(x), 8.0f);                     // q(8) < 1e-15; keeps inf * 0 out of the last fma
  const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.23164189f, 1.0f));          // p / sqrt(2), p = 0.3275911
  const float m = ax * 0.84932180f;                            // sqrt(0.5 * log2 e): exp(-x^2 / 2) = exp2(-m^2)
  const float e = __builtin_amdgcn_exp2f(-(m * m));
  float poly = fmaf(0.5307027145f, t, -0.7265760135f);         // 0.5 * a5, 0.5 * a4
  poly = fmaf(poly, t, 0.7107068705f);
  poly = fmaf(poly, t, -0.142248368f);
  poly = fmaf(poly, t, 0.127414796f);
  return fmaf(-ax, poly * t * e, fmaxf(x, 0.0f));
}

// ---- split-f16 ("f16x3") precision mode: an f32 value v travels as hi = f16(v) and lo = f16((v - hi) * 2^11); the lo array
// sits `lo_off` elements behind the hi array (one arena offset for every f16 activation buffer, 0 = mode off).  The scale
// keeps lo in the normal f16 range whenever hi is; a product of two split values is hi*hi + (hi*lo + lo*hi) * 2^-11.
#define SPLIT_SCALE 2048.0f
#define SPLIT_INV (1.0f / 2048.0f)
static __device__ __forceinline__ half_t split_lo(float v, half_t hi) { return (half_t)((v - (float)hi) * SPLIT_SCALE); }
static __device__ __forceinline__ void store_h1(half_t* p, size_t lo_off, float v) {
  const half_t h = (half_t)v;
  *p = h;
  if (lo_off) p[lo_off] = split_lo(v, h);
}
static __device__ __forceinline__ void store_h4(half_t* p, size_t lo_off, f32x4 v) {      // 8-byte aligned
  const half4 h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
  *reinterpret_cast<half4*>(p) = h;
  if (lo_off) {
    const half4 l = {split_lo(v[0], h[0]), split_lo(v[1], h[1]), split_lo(v[2], h[2]), split_lo(v[3], h[3])};
    *reinterpret_cast<half4*>(p + lo_off) = l;
  }
}

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
static __device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// LayerNorm fused into an MFMA operand load.  A token row of K = 16 KS channels is spread over the two lane halves of a
// wave in fragment order (lane (fr, fh) holds channels 16 s + 8 fh + e): `xp` = row base + 8 fh.  The row is read twice (the
// second read comes from cache): once for the f32 sum / sum of squares, once to normalise into f16 fragments.  The affine
// part is NOT applied here - the packer folds it into the consumer:  W (g * xhat + b) + c = (W diag g) xhat + (W b + c).
template <int KS>
static __device__ __forceinline__ void ln_row_fragments(const float* __restrict__ xp, float eps, half8 (&xf)[KS]) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(xp + 16 * s), b = *reinterpret_cast<const f32x4*>(xp + 16 * s + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s1 += a[e] + b[e];
      s2 = fmaf(a[e], a[e], fmaf(b[e], b[e], s2));
    }
  }
  s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 32, 64);
  const float mean = s1 * (1.0f / (16 * KS));
  const float rstd = rsqrtf(fmaxf(s2 * (1.0f / (16 * KS)) - mean * mean, 0.f) + eps);
  const float shift = -mean * rstd;
  asm volatile("" ::: "memory");                 // the second pass RE-READS the row (288 live f32 registers would not fit)
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(xp + 16 * s), b = *reinterpret_cast<const f32x4*>(xp + 16 * s + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xf[s][e] = (half_t)fmaf(a[e], rstd, shift);
      xf[s][4 + e] = (half_t)fmaf(b[e], rstd, shift);
    }
  }
}

// XCD-aware block remap (8 XCDs, blocks dealt round-robin): give each XCD a contiguous
// chunk of the logical grid so neighbouring tiles share an L2.  Bijective for any nwg.
static __device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + loc;
}

// window-major token order helpers: a (H x W) grid cut into w x w windows, windows row-major,
// tokens row-major inside a window.
static __host__ __device__ __forceinline__ int tok_of_yx(int y, int x, int W, int w) {
  return ((y / w) * (W / w) + (x / w)) * (w * w) + (y % w) * w + (x % w);
}

#define HIP_CHECK_RET(expr)                                                        \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) return sam2mi_set_error(ctx, #expr, hipGetErrorString(_e)); \
  } while (0)
