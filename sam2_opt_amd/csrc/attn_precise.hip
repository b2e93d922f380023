// Hiera head_dim-72 attention of the f16x3 precision mode (sam2mi_config.precision = 1).
// Reference: MultiScaleAttention.forward, /root/reference/sam2/sam2/modeling/backbones/hieradet.py:56-81.
//
// Why it exists: with every linear layer on split-f16 operands the f16 rounding of q and k inside the Hiera attention is
// what is left of the error budget (tools/precision_sim_video.py: mask logits 6.8e-4 of max|ref| with f16 q/k/v against
// 4e-6 without).  q, k and V^T therefore arrive as f32 (the QKV GEMM's f32 outputs) and every MFMA operand is split in
// registers into hi = f16(v), lo = f16((v - hi) * 2^11):
//   S^T[key][q] = K . Q^T       3 products (hi*hi, lo*hi, hi*lo), cross terms in their own accumulator, folded in x 2^-11
//   O^T[d][q]  += V^T . P^T     3 products as well (P is split like everything else)
// Same task decomposition, grouping and block-diagonal window mask as hiera_attn_kernel (attn_hiera.hip): one wave = 32
// queries of one (group, head), flash-style online softmax with per-lane statistics (a lane's accumulator column is its
// query), P taken straight from the S^T accumulator as the B operand of the second product (its rows are the keys).
// Two variants: the generic one takes its operands straight from global memory / L2 (any grouping, block-diagonal mask for packed
// small windows); precise_attn_shared_kernel serves the groupings where four consecutive query tiles see the same keys (stage-3
// windows and the global blocks - where the time is): the workgroup splits each 32-key K / V^T tile ONCE into hi + lo f16 and
// shares it through a double-buffered LDS tile, so the L2 traffic and the split arithmetic drop 4x.  The softmax rescale is applied
// on every tile (exactness before speed).  Neither is on the default (f16) path.
#include "attn.h"

namespace {
constexpr int HD = 72;

struct Split8 { half8 hi, lo; };

static __device__ __forceinline__ Split8 split8(const f32x4 a, const f32x4 b) {
  Split8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r.hi[j] = (half_t)a[j];
    r.lo[j] = split_lo(a[j], r.hi[j]);
    r.hi[4 + j] = (half_t)b[j];
    r.lo[4 + j] = split_lo(b[j], r.hi[4 + j]);
  }
  return r;
}
static __device__ __forceinline__ Split8 zero8() {
  Split8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r.hi[j] = r.lo[j] = (half_t)0.f;
  return r;
}

template <bool MASK>
__global__ __launch_bounds__(256, 2) void precise_attn_kernel(const PreciseAttnParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int qtiles = p.GQ / 32;
  const int total = p.num_groups * p.heads * qtiles;
  const int task = blockIdx.x * 4 + wave;
  if (task >= total) return;                          // no barriers in this kernel
  const int qt = task % qtiles;
  const int gh = task / qtiles;
  const int head = gh % p.heads, grp = gh / p.heads;

  // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q = fr][d = 16 s + 8 fh + j], zero beyond d = 72
  const size_t qrow = (size_t)grp * p.GQ + qt * 32 + fr;
  Split8 qf[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int d0 = s * 16 + fh * 8;
    if (d0 < HD) {
      const float* src = p.q + qrow * p.ldq + head * HD + d0;
      qf[s] = split8(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4));
    } else {
      qf[s] = zero8();
    }
  }

  // ---- key range of this query tile
  const int q_in_grp = qt * 32 + fr;
  int k_begin, k_end;
  if (p.wq >= 32) {
    const int w = (qt * 32) / p.wq;                   // the whole tile lies in one window
    k_begin = w * p.wk;
    k_end = k_begin + p.wk;
  } else {
    const int nwin = 32 / p.wq;                       // the tile covers nwin packed windows: block-diagonal mask below
    k_begin = qt * nwin * p.wk;
    k_end = k_begin + nwin * p.wk;
  }
  const int my_win = q_in_grp / p.wq;

  f32x16 oh[3], ox[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oh[t][r] = ox[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  for (int k0 = k_begin; k0 < k_end; k0 += 32) {
    // ---- S^T tile: rows = keys k0 + acc_row(r, lane), column = query fr
    f32x16 sh, sx;
#pragma unroll
    for (int r = 0; r < 16; ++r) sh[r] = sx[r] = 0.f;
    const float* krow = p.k + ((size_t)grp * p.GK + k0 + fr) * p.ldk + head * HD;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int d0 = s * 16 + fh * 8;
      Split8 kf = zero8();
      if (d0 < HD) kf = split8(*reinterpret_cast<const f32x4*>(krow + d0), *reinterpret_cast<const f32x4*>(krow + d0 + 4));
      sh = mfma32(kf.hi, qf[s].hi, sh);
      sx = mfma32(kf.lo, qf[s].hi, sx);
      sx = mfma32(kf.hi, qf[s].lo, sx);
    }
    float sv[16];
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = fmaf(sx[r], SPLIT_INV, sh[r]);
      if (MASK) {
        const int key_in_grp = k0 + acc_row(r, lane);
        if (key_in_grp / p.wk != my_win) v = -INFINITY;
      }
      sv[r] = v;
      tmax = fmaxf(tmax, v);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));      // lanes fr and fr + 32 hold the two key halves of query fr
    const float m_new = fmaxf(m_run, tmax);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = exp2f(m_run - m_safe);          // m_run = -inf -> 0
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
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        oh[t][r] *= alpha;
        ox[t][r] *= alpha;
      }
    // ---- O^T += V^T P^T.  B operand of k-step ks = registers 8 ks .. 8 ks + 7 of the S^T accumulator, i.e. keys
    // k0 + 16 ks + 8 (j >> 2) + 4 fh + (j & 3): the V^T fragment gathers the same keys (two 16-B loads per lane).
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Split8 pf;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pf.hi[j] = (half_t)pv[8 * ks + j];
        pf.lo[j] = split_lo(pv[8 * ks + j], pf.hi[j]);
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int d = t * 32 + fr;
        Split8 vf = zero8();
        if (d < HD) {
          const float* vsrc = p.vT + (size_t)(head * HD + d) * p.ldvT + (size_t)grp * p.GK + k0 + 16 * ks + 4 * fh;
          vf = split8(*reinterpret_cast<const f32x4*>(vsrc), *reinterpret_cast<const f32x4*>(vsrc + 8));
        }
        oh[t] = mfma32(vf.hi, pf.hi, oh[t]);
        ox[t] = mfma32(vf.lo, pf.hi, ox[t]);
        ox[t] = mfma32(vf.hi, pf.lo, ox[t]);
      }
    }
  }

  const float inv = 1.f / l_run;
  half_t* orow = p.o + qrow * p.ldo + head * HD;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = t * 32 + 8 * g + 4 * fh;
      if (d < HD) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(ox[t][4 * g + e], SPLIT_INV, oh[t][4 * g + e]) * inv;
        store_h4(orow + d, p.o_lo_off, v);
      }
    }
  }
}

// ---- shared-tile variant: 4 waves = 4 consecutive query tiles of one (group, head) whose key range is the same
constexpr int KROW = 88;                 // K tile row stride in halfs (176 B: conflict-free ds_read_b128), 80 used
constexpr int VROW = 40;                 // V^T tile row stride in halfs (80 B), 32 used
constexpr int K_TILE = 32 * KROW;        // halfs per plane
constexpr int V_TILE = 96 * VROW;
constexpr int TILE_H = 2 * K_TILE + 2 * V_TILE;      // K hi | K lo | V hi | V lo = 13,312 halfs = 26 KB; two buffers = 52 KB

__global__ __launch_bounds__(256, 2) void precise_attn_shared_kernel(const PreciseAttnParams p) {
  __shared__ __attribute__((aligned(16))) half_t smem[2 * TILE_H];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int qtiles = p.GQ / 32;                        // multiple of 4 (checked by the launcher)
  const int task0 = blockIdx.x * 4;                    // first of this workgroup's 4 query tiles
  const int qt = task0 % qtiles + wave;
  const int gh = task0 / qtiles;
  const int head = gh % p.heads, grp = gh / p.heads;

  const size_t qrow = (size_t)grp * p.GQ + qt * 32 + fr;
  Split8 qf[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int d0 = s * 16 + fh * 8;
    if (d0 < HD) {
      const float* src = p.q + qrow * p.ldq + head * HD + d0;
      qf[s] = split8(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4));
    } else {
      qf[s] = zero8();
    }
  }
  const int w = ((task0 % qtiles) * 32) / p.wq;        // all four tiles lie in one window (wq % 128 == 0)
  const int k_begin = w * p.wk, k_end = k_begin + p.wk;

  // zero the pad columns (d = 72..79 of K rows; rows 72..95 of V^T) of both buffers once: staging never touches them
  for (int i = tid; i < 2 * TILE_H; i += 256) smem[i] = (half_t)0.f;
  __syncthreads();

  // staging: K tile = 32 keys x 72 f32 (18 float4 per key), V^T tile = 72 rows x 32 keys f32 (8 float4 per row)
  auto stage = [&](int buf, int k0) {
    half_t* sK = smem + buf * TILE_H;
    half_t* sV = sK + 2 * K_TILE;
    for (int i = tid; i < 32 * 18; i += 256) {
      const int key = i / 18, c4 = (i % 18) * 4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(p.k + ((size_t)grp * p.GK + k0 + key) * p.ldk + head * HD + c4);
      half4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) { h[e] = (half_t)v[e]; l[e] = split_lo(v[e], h[e]); }
      *reinterpret_cast<half4*>(sK + key * KROW + c4) = h;
      *reinterpret_cast<half4*>(sK + K_TILE + key * KROW + c4) = l;
    }
    for (int i = tid; i < HD * 8; i += 256) {
      const int d = i >> 3, c4 = (i & 7) * 4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(p.vT + (size_t)(head * HD + d) * p.ldvT + (size_t)grp * p.GK + k0 + c4);
      half4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) { h[e] = (half_t)v[e]; l[e] = split_lo(v[e], h[e]); }
      *reinterpret_cast<half4*>(sV + d * VROW + c4) = h;
      *reinterpret_cast<half4*>(sV + V_TILE + d * VROW + c4) = l;
    }
  };

  f32x16 oh[3], ox[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oh[t][r] = ox[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  stage(0, k_begin);
  __syncthreads();
  int buf = 0;
  for (int k0 = k_begin; k0 < k_end; k0 += 32, buf ^= 1) {
    if (k0 + 32 < k_end) stage(buf ^ 1, k0 + 32);      // the other buffer was released by the barrier at the end of the last tile
    const half_t* sK = smem + buf * TILE_H;
    const half_t* sV = sK + 2 * K_TILE;
    f32x16 sh, sx;
#pragma unroll
    for (int r = 0; r < 16; ++r) sh[r] = sx[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const half8 kh = *reinterpret_cast<const half8*>(sK + fr * KROW + s * 16 + fh * 8);
      const half8 kl = *reinterpret_cast<const half8*>(sK + K_TILE + fr * KROW + s * 16 + fh * 8);
      sh = mfma32(kh, qf[s].hi, sh);
      sx = mfma32(kl, qf[s].hi, sx);
      sx = mfma32(kh, qf[s].lo, sx);
    }
    float sv[16];
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sv[r] = fmaf(sx[r], SPLIT_INV, sh[r]);
      tmax = fmaxf(tmax, sv[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = exp2f(m_run - m_new);
    float psum = 0.f;
    float pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pv[r] = exp2f(sv[r] - m_new);
      psum += pv[r];
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        oh[t][r] *= alpha;
        ox[t][r] *= alpha;
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Split8 pf;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pf.hi[j] = (half_t)pv[8 * ks + j];
        pf.lo[j] = split_lo(pv[8 * ks + j], pf.hi[j]);
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const half_t* vr = sV + (t * 32 + fr) * VROW + 16 * ks + 4 * fh;      // keys 16 ks + 4 fh + (0..3) and + 8
        const half4 a0 = *reinterpret_cast<const half4*>(vr), a1 = *reinterpret_cast<const half4*>(vr + 8);
        const half4 b0 = *reinterpret_cast<const half4*>(vr + V_TILE), b1 = *reinterpret_cast<const half4*>(vr + V_TILE + 8);
        const half8 vh = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const half8 vl = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        oh[t] = mfma32(vh, pf.hi, oh[t]);
        ox[t] = mfma32(vl, pf.hi, ox[t]);
        ox[t] = mfma32(vh, pf.lo, ox[t]);
      }
    }
    __syncthreads();                                    // tile consumed by every wave; the next one is staged and visible
  }

  const float inv = 1.f / l_run;
  half_t* orow = p.o + qrow * p.ldo + head * HD;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = t * 32 + 8 * g + 4 * fh;
      if (d < HD) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(ox[t][4 * g + e], SPLIT_INV, oh[t][4 * g + e]) * inv;
        store_h4(orow + d, p.o_lo_off, v);
      }
    }
  }
}
}  // namespace

hipError_t precise_attn_launch(const PreciseAttnParams& p, hipStream_t stream) {
  if (p.GQ % 32 || p.GK % 32 || p.num_groups <= 0 || p.o_lo_off == 0) return hipErrorInvalidValue;
  if ((p.ldq & 3) || (p.ldk & 3) || (p.ldvT & 3) || (p.ldo & 3)) return hipErrorInvalidValue;
  if (p.wq < 32 && (32 % p.wq)) return hipErrorInvalidValue;
  if (p.wq >= 32 && (p.wq % 32 || p.wk % 32)) return hipErrorInvalidValue;
  if (p.wq < 32 && ((32 / p.wq) * p.wk) % 32) return hipErrorInvalidValue;
  const int total = p.num_groups * p.heads * (p.GQ / 32);
  const dim3 grid((total + 3) / 4), block(256);
  if (p.wq % 128 == 0 && p.GQ % 128 == 0) precise_attn_shared_kernel<<<grid, block, 0, stream>>>(p);      // 4 query tiles per window share K / V
  else if (p.wq < 32) precise_attn_kernel<true><<<grid, block, 0, stream>>>(p);
  else precise_attn_kernel<false><<<grid, block, 0, stream>>>(p);
  return hipGetLastError();
}
