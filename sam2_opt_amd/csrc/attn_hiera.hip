// Hiera multi-head attention for head_dim 72 (windowed, pooled-query and global blocks).
// Reference: MultiScaleAttention.forward, /root/reference/sam2/sam2/modeling/backbones/hieradet.py:56-81.
//
// One wave = one tile of 32 queries of one (group, head); a group is a window (or a pack of
// small windows with a block-diagonal mask).  Flash-style online softmax, swapped products:
//   S^T[key][q] = K . Q^T      (A = K fragment from LDS, B = Q fragment in registers)
//   O^T[d][q]  += V^T . P^T    (A = V^T fragment from LDS, B = P^T straight from the S^T
//                               accumulator: its rows are the keys the second product sums over)
// so softmax statistics are per lane (column = query) and P never touches LDS.
// head_dim 72 is padded in LDS/registers only: QK^T sums over 80 (5 MFMA k-steps, pad = 0),
// PV produces 96 rows (3 MFMA row tiles) of which 72 are stored.
#include "attn.h"

namespace {
constexpr int HD = 72;
constexpr int KROW = 88;              // K tile row stride in halfs (176 B: conflict-free ds_read_b128)
constexpr int VROW = 36;              // V^T tile row stride in halfs (72 B: conflict-free ds_read_b64)
constexpr int K_TILE = 32 * KROW;     // halfs
constexpr int V_TILE = 96 * VROW;     // halfs (rows 72..95 are never written; their products are discarded)
constexpr int REGION = K_TILE + V_TILE;

template <bool SHARE>
__global__ __launch_bounds__(256) void hiera_attn_kernel(const HieraAttnParams p) {
  __shared__ __attribute__((aligned(16))) half_t smem[(SHARE ? 1 : 4) * REGION];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int qtiles = p.GQ / 32;
  const int total = p.num_groups * p.heads * qtiles;
  int task = blockIdx.x * 4 + wave;
  const bool live = task < total;
  if (!live) task = total - 1;          // keep barrier participation uniform
  const int qt = task % qtiles;
  const int gh = task / qtiles;
  const int head = gh % p.heads, grp = gh / p.heads;

  half_t* sK = smem + (SHARE ? 0 : wave * REGION);
  half_t* sV = sK + K_TILE;
  // zero the head-dim pad (cols 72..87) of the K tile once; staging never touches it
  {
    const int nthr = SHARE ? 256 : 64, t = SHARE ? tid : lane;
    for (int i = t; i < 32 * 16; i += nthr) sK[(i >> 4) * KROW + HD + (i & 15)] = (half_t)0.f;
  }

  // ---- Q fragments (B operand): lane holds Q[q = fr][d = 16 s + 8 fh + j]
  const size_t qrow = (size_t)grp * p.GQ + qt * 32 + fr;
  half8 qf[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    if (s == 4 && fh == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = (half_t)0.f;
    } else {
      qf[s] = *reinterpret_cast<const half8*>(p.q + qrow * p.ldq + head * HD + s * 16 + fh * 8);
    }
  }

  const int q_in_grp = qt * 32 + fr;                 // this lane's query index inside the group
  const int nwin = p.wq >= 32 ? 1 : 32 / p.wq;
  const int kv_start = ((qt * 32) / p.wq) * p.wk;    // first key (inside group) this q-tile can see
  const int kv_tiles = (nwin * p.wk) / 32;
  const int q_win = q_in_grp / p.wq;

  f32x16 o[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const half_t* kbase = p.k + ((size_t)grp * p.GK) * p.ldk + head * HD;
  const half_t* vbase = p.vT + (size_t)head * HD * p.ldvT + (size_t)grp * p.GK;

  for (int kt = 0; kt < kv_tiles; ++kt) {
    const int k0 = kv_start + kt * 32;
    __syncthreads();                                  // previous tile fully consumed
    {
      const int nthr = SHARE ? 256 : 64, t = SHARE ? tid : lane;
      // K tile: 32 rows x 9 chunks of 8 halfs
      for (int c = t; c < 32 * 9; c += nthr) {
        const int row = c / 9, ch = c % 9;
        const half8 v = *reinterpret_cast<const half8*>(kbase + (size_t)(k0 + row) * p.ldk + ch * 8);
        *reinterpret_cast<half8*>(sK + row * KROW + ch * 8) = v;
      }
      // V^T tile: 72 rows x 8 chunks of 4 halfs (8-B LDS writes: rows are 72 B apart)
      for (int c = t; c < HD * 8; c += nthr) {
        const int row = c >> 3, ch = c & 7;
        const half4 v = *reinterpret_cast<const half4*>(vbase + (size_t)row * p.ldvT + k0 + ch * 4);
        *reinterpret_cast<half4*>(sV + row * VROW + ch * 4) = v;
      }
    }
    __syncthreads();

    // ---- S^T = K Q^T  (32 keys x 32 queries)
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      const half8 kf = *reinterpret_cast<const half8*>(sK + fr * KROW + ks * 16 + fh * 8);
      s = mfma32(kf, qf[ks], s);
    }
    // ---- online softmax over the key axis (registers + the other lane half)
    float tmax = -1e30f;
    bool valid[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kj = k0 + acc_row(r, lane);
      valid[r] = (kj / p.wk) == q_win;
      s[r] *= p.scale_log2e;
      if (valid[r]) tmax = fmaxf(tmax, s[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = exp2f(m_run - m_new);
    float psum = 0.f;
    half8 pf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = valid[r] ? exp2f(s[r] - m_new) : 0.f;
      psum += pv;
      pf[r >> 3][r & 7] = (half_t)pv;
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
    // ---- O^T += V^T P^T
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const half_t* vr = sV + (t * 32 + fr) * VROW + ks * 16 + fh * 4;
        const half4 lo = *reinterpret_cast<const half4*>(vr);
        const half4 hi = *reinterpret_cast<const half4*>(vr + 8);
        const half8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[t] = mfma32(vf, pf[ks], o[t]);
      }
    }
  }

  if (live) {
    const float inv = 1.f / l_run;
    half_t* orow = p.o + qrow * p.ldo + head * HD;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = t * 32 + 8 * g + 4 * fh;
        if (d < HD) {
          const half4 h = {(half_t)(o[t][4 * g] * inv), (half_t)(o[t][4 * g + 1] * inv),
                           (half_t)(o[t][4 * g + 2] * inv), (half_t)(o[t][4 * g + 3] * inv)};
          *reinterpret_cast<half4*>(orow + d) = h;
        }
      }
    }
  }
}
}  // namespace

hipError_t hiera_attn_launch(const HieraAttnParams& p, hipStream_t stream) {
  if (p.GQ % 32 || p.GK % 32 || p.num_groups <= 0) return hipErrorInvalidValue;
  if ((p.ldq & 7) || (p.ldk & 7) || (p.ldvT & 3) || (p.ldo & 3)) return hipErrorInvalidValue;
  if (p.wq < 32 && (32 % p.wq)) return hipErrorInvalidValue;
  if (p.wq >= 32 && (p.wq % 32 || p.wk % 32)) return hipErrorInvalidValue;
  if (p.wq < 32 && ((32 / p.wq) * p.wk) % 32) return hipErrorInvalidValue;
  const int qtiles = p.GQ / 32;
  const int total = p.num_groups * p.heads * qtiles;
  const int blocks = (total + 3) / 4;
  if (qtiles % 4 == 0)
    hiera_attn_kernel<true><<<dim3(blocks), dim3(256), 0, stream>>>(p);
  else
    hiera_attn_kernel<false><<<dim3(blocks), dim3(256), 0, stream>>>(p);
  return hipGetLastError();
}
