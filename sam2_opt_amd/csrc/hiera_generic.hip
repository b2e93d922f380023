// Hiera blocks for the model sizes whose windows do NOT tile the token grid (hiera tiny / small / base+: window_spec 8, 4, 14, 7;
// head_dim 96 or 56).  Reference: MultiScaleBlock.forward / MultiScaleAttention.forward
// (/root/reference/sam2/sam2/modeling/backbones/hieradet.py:56-81,:134-166) with window_partition / window_unpartition
// (backbones/utils.py:16-60): the LayerNorm output is ZERO-padded to a multiple of the window size, pad tokens go through the
// QKV projection like any other (q = k = v = bias) and are attended to; the output is cropped back.
//
// The large model's path keeps the residual stream in window-major order and never materialises windows (engine_encoder.hip);
// that trick needs windows that tile the grid.  Here the residual stream is plain row-major [B, H, W, C] and every block
//   gathers LN(x) into a window layout (window_gather_kernel: zero rows for image padding and for the row padding that brings
//   a window to a multiple of 32 tokens), runs the QKV GEMM on those rows, max-pools q inside windows on the stage transitions
//   (window_pool_q_kernel), attends (generic_attn_kernel: any head_dim <= 96, block-diagonal window mask, valid-key count per
//   window) and scatters the result back to the cropped row-major grid (window_scatter_kernel).
// Written for coverage, not speed: these sizes are not the benchmark configuration (BASELINE.json configs[1..4] are hiera-large).
#include "kernels.h"
#include "attn.h"

namespace {

// src [B, H, W, C] f16 row-major -> dst [B * nW * nW * wk, C]: window (wy, wx) of image b owns rows ((b nW + wy) nW + wx) wk + t,
// t = ty * w + tx < w * w; rows t >= w * w and tokens outside the image are zero.
__global__ void window_gather_kernel(const half_t* __restrict__ src, half_t* __restrict__ dst, int B, int H, int W, int C8, int w, int nW, int wk) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * nW * nW * wk * C8;
  if (i >= total) return;
  const int c = (int)(i % C8);
  const size_t row = i / C8;
  const int t = (int)(row % wk);
  const size_t win = row / wk;
  const int wx = (int)(win % nW), wy = (int)((win / nW) % nW), b = (int)(win / ((size_t)nW * nW));
  half8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (half_t)0.f;
  if (t < w * w) {
    const int y = wy * w + t / w, x = wx * w + t % w;
    if (y < H && x < W) v = *reinterpret_cast<const half8*>(src + (((size_t)b * H + y) * W + x) * (size_t)C8 * 8 + c * 8);
  }
  *reinterpret_cast<half8*>(dst + row * (size_t)C8 * 8 + c * 8) = v;
}

// q [nwin * wk, ldq] (window layout, w x w valid tokens per window) -> out [nwin * wq, C]: 2x2 max-pool inside the window
// (hieradet.py:64-67: do_pool on the (Bw, w, w, C) view), rows tq >= (w/2)^2 zero
__global__ void window_pool_q_kernel(const half_t* __restrict__ q, int ldq, half_t* __restrict__ out, int C, int nwin, int w, int wk, int wq) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)nwin * wq * C) return;
  const int c = (int)(i % C);
  const size_t row = i / C;
  const int tq = (int)(row % wq);
  const size_t win = row / wq;
  const int hw = w / 2;
  float m = 0.f;
  if (tq < hw * hw) {
    const int py = tq / hw, px = tq % hw;
    m = -3.0e38f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) m = fmaxf(m, (float)q[(win * wk + (size_t)(2 * py + dy) * w + 2 * px + dx) * ldq + c]);
  }
  out[row * C + c] = (half_t)m;
}

// src [B * nW * nW * wq, C] (window layout over a (nW wq_edge)^2 padded grid, wq_edge = edge of the query window) -> dst [B, H, W, C]
__global__ void window_scatter_kernel(const half_t* __restrict__ src, half_t* __restrict__ dst, int B, int H, int W, int C8, int we, int nW, int wq) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * H * W * C8) return;
  const int c = (int)(i % C8);
  const size_t tok = i / C8;
  const int x = (int)(tok % W), y = (int)((tok / W) % H), b = (int)(tok / ((size_t)H * W));
  const size_t row = (((size_t)b * nW + y / we) * nW + x / we) * wq + (size_t)(y % we) * we + x % we;
  *reinterpret_cast<half8*>(dst + tok * (size_t)C8 * 8 + c * 8) = *reinterpret_cast<const half8*>(src + row * (size_t)C8 * 8 + c * 8);
}

// One wave = 32 queries of one (group, head).  Query i of a group sees key j iff i / wq == j / wk and j % wk < vk.
// Swapped products like the other attention kernels: S^T = K Q^T, O^T += V^T P^T with P from the S^T accumulator.
template <int HD>
__global__ __launch_bounds__(256) void generic_attn_kernel(const GenericAttnParams p) {
  constexpr int KS = (HD + 15) / 16, DT = (HD + 31) / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int qtiles = p.GQ / 32;
  const int total = p.num_groups * p.heads * qtiles;
  const int task = blockIdx.x * 4 + wave;
  if (task >= total) return;
  const int qt = task % qtiles;
  const int gh = task / qtiles;
  const int head = gh % p.heads, grp = gh / p.heads;
  const size_t qrow = (size_t)grp * p.GQ + qt * 32 + fr;
  half8 qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int d0 = s * 16 + fh * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[s][j] = (half_t)0.f;
    if (d0 < HD) qf[s] = *reinterpret_cast<const half8*>(p.q + qrow * p.ldq + head * HD + d0);
  }
  const int q_in_grp = qt * 32 + fr;
  const int my_win = q_in_grp / p.wq;
  int k_begin, k_end;
  if (p.wq >= 32) {
    const int w = (qt * 32) / p.wq;
    k_begin = w * p.wk;
    k_end = k_begin + (p.vk + 31) / 32 * 32;           // tiles wholly past the valid keys are skipped
  } else {
    const int nwin = 32 / p.wq;
    k_begin = qt * nwin * p.wk;
    k_end = k_begin + nwin * p.wk;
  }
  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  for (int k0 = k_begin; k0 < k_end; k0 += 32) {
    f32x16 sa;
#pragma unroll
    for (int r = 0; r < 16; ++r) sa[r] = 0.f;
    const half_t* krow = p.k + ((size_t)grp * p.GK + k0 + fr) * p.ldk + head * HD;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int d0 = s * 16 + fh * 8;
      half8 kf;
#pragma unroll
      for (int j = 0; j < 8; ++j) kf[j] = (half_t)0.f;
      if (d0 < HD) kf = *reinterpret_cast<const half8*>(krow + d0);
      sa = mfma32(kf, qf[s], sa);
    }
    float sv[16];
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = k0 + acc_row(r, lane);               // key index inside the group
      const bool ok = (j / p.wk == my_win) && (j % p.wk < p.vk);
      sv[r] = ok ? sa[r] : -INFINITY;
      tmax = fmaxf(tmax, sv[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = exp2f(m_run - m_safe);
    float psum = 0.f;
    float pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pv[r] = exp2f(sv[r] - m_safe);
      psum += pv[r];
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8 pf;
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = (half_t)pv[8 * ks + j];
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const int d = t * 32 + fr;
        half8 vf;
#pragma unroll
        for (int j = 0; j < 8; ++j) vf[j] = (half_t)0.f;
        if (d < HD) {
          const half_t* vsrc = p.vT + (size_t)(head * HD + d) * p.ldvT + (size_t)grp * p.GK + k0 + 16 * ks + 4 * fh;
          const half4 a = *reinterpret_cast<const half4*>(vsrc), b = *reinterpret_cast<const half4*>(vsrc + 8);
          vf = half8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        }
        o[t] = mfma32(vf, pf, o[t]);
      }
    }
  }
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;      // a padding query row of a packed group may see no key: its output is dropped anyway
  half_t* orow = p.o + qrow * p.ldo + head * HD;
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = t * 32 + 8 * g + 4 * fh;
      if (d < HD) {
        const half4 h = {(half_t)(o[t][4 * g] * inv), (half_t)(o[t][4 * g + 1] * inv), (half_t)(o[t][4 * g + 2] * inv), (half_t)(o[t][4 * g + 3] * inv)};
        *reinterpret_cast<half4*>(orow + d) = h;
      }
    }
}
inline dim3 grid1d(size_t n) { return dim3((unsigned)((n + 255) / 256)); }
}  // namespace

hipError_t window_gather_launch(const half_t* src, half_t* dst, int B, int H, int W, int C, int w, int nW, int wk, hipStream_t s) {
  if (C % 8) return hipErrorInvalidValue;
  window_gather_kernel<<<grid1d((size_t)B * nW * nW * wk * (C / 8)), dim3(256), 0, s>>>(src, dst, B, H, W, C / 8, w, nW, wk);
  return hipGetLastError();
}
hipError_t window_pool_q_launch(const half_t* q, int ldq, half_t* out, int C, int nwin, int w, int wk, int wq, hipStream_t s) {
  window_pool_q_kernel<<<grid1d((size_t)nwin * wq * C), dim3(256), 0, s>>>(q, ldq, out, C, nwin, w, wk, wq);
  return hipGetLastError();
}
hipError_t window_scatter_launch(const half_t* src, half_t* dst, int B, int H, int W, int C, int we, int nW, int wq, hipStream_t s) {
  if (C % 8) return hipErrorInvalidValue;
  window_scatter_kernel<<<grid1d((size_t)B * H * W * (C / 8)), dim3(256), 0, s>>>(src, dst, B, H, W, C / 8, we, nW, wq);
  return hipGetLastError();
}
hipError_t generic_attn_launch(const GenericAttnParams& p, int head_dim, hipStream_t s) {
  if (p.GQ % 32 || p.GK % 32 || p.num_groups <= 0 || (p.ldq & 7) || (p.ldk & 7) || (p.ldvT & 3) || (p.ldo & 3)) return hipErrorInvalidValue;
  if (p.wq < 32 ? (32 % p.wq || ((32 / p.wq) * p.wk) % 32) : (p.wq % 32 || p.wk % 32)) return hipErrorInvalidValue;
  if (p.vk < 1 || p.vk > p.wk || (head_dim & 7)) return hipErrorInvalidValue;
  const int total = p.num_groups * p.heads * (p.GQ / 32);
  const dim3 grid((total + 3) / 4), block(256);
  switch (head_dim) {
    case 56: generic_attn_kernel<56><<<grid, block, 0, s>>>(p); break;
    case 72: generic_attn_kernel<72><<<grid, block, 0, s>>>(p); break;
    case 96: generic_attn_kernel<96><<<grid, block, 0, s>>>(p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
