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

template <bool SHARE, bool MASK>
__global__ __launch_bounds__(256) void hiera_attn_kernel(const HieraAttnParams p) {
  __shared__ __attribute__((aligned(16))) half_t smem[(SHARE ? 1 : 4) * REGION];
  constexpr int NTHR = SHARE ? 256 : 64;
  constexpr int K_IT = (32 * 9 + NTHR - 1) / NTHR;      // 16-B chunks of the K tile per thread
  constexpr int V_IT = (HD * 8 + NTHR - 1) / NTHR;      // 8-B chunks of the V^T tile per thread
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int st = SHARE ? tid : lane;
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
  for (int i = st; i < 32 * 16; i += NTHR) sK[(i >> 4) * KROW + HD + (i & 15)] = (half_t)0.f;

  // ---- Q fragments (B operand): lane holds Q[q = fr][d = 16 s + 8 fh + j]; Q is pre-scaled by 72^-0.5 * log2(e)
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

  // register-staged tiles: the loads of tile t+1 are issued before the MFMAs of tile t
  half8 rk[K_IT];
  half4 rv[V_IT];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < K_IT; ++i) {
      const int c = st + i * NTHR;
      if (c < 32 * 9) rk[i] = *reinterpret_cast<const half8*>(kbase + (size_t)(k0 + c / 9) * p.ldk + (c % 9) * 8);
    }
#pragma unroll
    for (int i = 0; i < V_IT; ++i) {
      const int c = st + i * NTHR;
      if (c < HD * 8) rv[i] = *reinterpret_cast<const half4*>(vbase + (size_t)(c >> 3) * p.ldvT + k0 + (c & 7) * 4);
    }
  };
  auto swrite = [&]() {
#pragma unroll
    for (int i = 0; i < K_IT; ++i) {
      const int c = st + i * NTHR;
      if (c < 32 * 9) *reinterpret_cast<half8*>(sK + (c / 9) * KROW + (c % 9) * 8) = rk[i];
    }
#pragma unroll
    for (int i = 0; i < V_IT; ++i) {
      const int c = st + i * NTHR;
      if (c < HD * 8) *reinterpret_cast<half4*>(sV + (c >> 3) * VROW + (c & 7) * 4) = rv[i];
    }
  };

  gload(kv_start);
  swrite();
  __syncthreads();
#pragma nounroll
  for (int kt = 0; kt < kv_tiles; ++kt) {
    const int k0 = kv_start + kt * 32;
    if (kt + 1 < kv_tiles) gload(k0 + 32);
    // ---- S^T = K Q^T  (32 keys x 32 queries), exp2 domain
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      const half8 kf = *reinterpret_cast<const half8*>(sK + fr * KROW + ks * 16 + fh * 8);
      s = mfma32(kf, qf[ks], s);
    }
    float tmax = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (MASK && ((k0 + acc_row(r, lane)) / p.wk) != q_win) s[r] = -1e30f;
      tmax = fmaxf(tmax, s[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    float psum = 0.f;
    half8 pf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float pv = __builtin_amdgcn_exp2f(s[r] - m_new);
      if (MASK && s[r] <= -1e30f) pv = 0.f;          // fully masked tile for this query: m_new may still be the sentinel
      psum += pv;
      pf[r >> 3][r & 7] = (half_t)pv;
    }
    psum += __shfl_xor(psum, 32, 64);
    if (__any(m_new > m_run)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      m_run = m_new;
    }
    l_run += psum;
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
    __syncthreads();                                  // tile fully consumed
    if (kt + 1 < kv_tiles) swrite();
    __syncthreads();
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
  const bool mask = !(p.wq >= p.GQ && p.wk >= p.GK);
  if (qtiles % 4 == 0) {
    if (mask) hiera_attn_kernel<true, true><<<dim3(blocks), dim3(256), 0, stream>>>(p);
    else hiera_attn_kernel<true, false><<<dim3(blocks), dim3(256), 0, stream>>>(p);
  } else {
    if (mask) hiera_attn_kernel<false, true><<<dim3(blocks), dim3(256), 0, stream>>>(p);
    else hiera_attn_kernel<false, false><<<dim3(blocks), dim3(256), 0, stream>>>(p);
  }
  return hipGetLastError();
}
