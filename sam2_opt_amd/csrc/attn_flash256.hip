// Single-head, head_dim 256 flash attention for the memory attention (self- and cross-attention
// of RoPEAttention, /root/reference/sam2/sam2/modeling/sam/transformer.py:345-424: 1 head, d=256,
// q = 4096 tokens, kv = 4096 (self) or L*4096+P <= 28736 (memory bank + object pointers)).
//
// 4096 queries are only 128 wave tiles, so the KV axis is split across workgroups (grid.y) and a
// combine pass merges the partial (m, l, O).  Workgroup = 4 waves = 128 queries; the 4 waves share
// each 32-key K tile and V^T tile through LDS.  Per wave: S^T = K.Q^T (16 MFMA k-steps over d),
// in-register online softmax (column = query on the lane), O^T += V^T.P^T with P^T taken straight
// from the S^T accumulator (8 row tiles x 2 k-steps).  The next tile's global loads are issued
// before the MFMAs of the current one and written to LDS after them.
#include "attn.h"

namespace {
constexpr int D = 256;
constexpr int KROW = 264;            // 528-B rows: conflict-free ds_read_b128 (see gemm.hip note)
constexpr int VROW = 36;             // 72-B rows: conflict-free ds_read_b64
constexpr int K_TILE = 32 * KROW;    // halfs
constexpr int V_TILE = D * VROW;

__global__ __launch_bounds__(256, 1) void flash256_kernel(const Flash256Params p) {
  __shared__ __attribute__((aligned(16))) half_t smem[K_TILE + V_TILE];
  half_t* sK = smem;
  half_t* sV = smem + K_TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int split = blockIdx.y;
  const int ntiles = (p.Nk + 31) / 32;
  const int per = (ntiles + p.splits - 1) / p.splits;
  const int t_lo = split * per, t_hi = min(ntiles, t_lo + per);

  // Q fragments (B operand): Q[q = fr][d = 16 s + 8 fh + j]
  half8 qf[16];
  {
    const half_t* qp = p.q + (size_t)(q0 + fr) * p.ldq + fh * 8;
#pragma unroll
    for (int s = 0; s < 16; ++s) qf[s] = *reinterpret_cast<const half8*>(qp + s * 16);
  }
  f32x16 o[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  // staging: K tile = 32 rows x 32 chunks(16 B) = 1024 chunks -> 4 per thread
  //          V^T tile = 256 rows x 4 chunks(16 B) = 1024 chunks -> 4 per thread
  half8 rk[4], rv[4];
  auto gload = [&](int tile) {
    const int k0 = tile * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256;
      rk[i] = *reinterpret_cast<const half8*>(p.k + (size_t)(k0 + (c >> 5)) * p.ldk + (c & 31) * 8);
      rv[i] = *reinterpret_cast<const half8*>(p.vT + (size_t)(c >> 2) * p.ldvT + k0 + (c & 3) * 8);
    }
  };
  auto swrite = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256;
      *reinterpret_cast<half8*>(sK + (c >> 5) * KROW + (c & 31) * 8) = rk[i];
      half_t* vd = sV + (c >> 2) * VROW + (c & 3) * 8;
      const half4 lo = {rv[i][0], rv[i][1], rv[i][2], rv[i][3]};
      const half4 hi = {rv[i][4], rv[i][5], rv[i][6], rv[i][7]};
      *reinterpret_cast<half4*>(vd) = lo;
      *reinterpret_cast<half4*>(vd + 4) = hi;
    }
  };

  if (t_lo < t_hi) {
    gload(t_lo);
    swrite();
  }
  __syncthreads();
  for (int tile = t_lo; tile < t_hi; ++tile) {
    const int k0 = tile * 32;
    if (tile + 1 < t_hi) gload(tile + 1);
    // ---- S^T = K Q^T
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const half8 kf = *reinterpret_cast<const half8*>(sK + fr * KROW + ks * 16 + fh * 8);
      s = mfma32(kf, qf[ks], s);
    }
    float tmax = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] *= p.scale_log2e;
      if (k0 + acc_row(r, lane) < p.Nk) tmax = fmaxf(tmax, s[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = exp2f(m_run - m_new);
    float psum = 0.f;
    half8 pf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = (k0 + acc_row(r, lane) < p.Nk) ? exp2f(s[r] - m_new) : 0.f;
      psum += pv;
      pf[r >> 3][r & 7] = (half_t)pv;
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
    // ---- O^T += V^T P^T
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const half_t* vr = sV + (t * 32 + fr) * VROW + ks * 16 + fh * 4;
        const half4 lo = *reinterpret_cast<const half4*>(vr);
        const half4 hi = *reinterpret_cast<const half4*>(vr + 8);
        const half8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[t] = mfma32(vf, pf[ks], o[t]);
      }
    }
    __syncthreads();                    // everyone done reading this tile
    if (tile + 1 < t_hi) swrite();
    __syncthreads();
  }

  // ---- partial results
  const int q = q0 + fr;
  float* op = p.o_part + ((size_t)split * p.Nq + q) * D;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = {o[t][4 * g], o[t][4 * g + 1], o[t][4 * g + 2], o[t][4 * g + 3]};
      *reinterpret_cast<f32x4*>(op + t * 32 + 8 * g + 4 * fh) = v;
    }
  if (fh == 0) {
    float* ml = p.ml_part + ((size_t)split * p.Nq + q) * 2;
    ml[0] = m_run;
    ml[1] = l_run;
  }
}

// one workgroup (256 threads = 256 channels) per query
__global__ __launch_bounds__(256) void flash256_combine_kernel(const Flash256Params p) {
  const int q = blockIdx.x, d = threadIdx.x;
  float mstar = -1e30f;
  for (int s = 0; s < p.splits; ++s) mstar = fmaxf(mstar, p.ml_part[((size_t)s * p.Nq + q) * 2]);
  float L = 0.f, acc = 0.f;
  for (int s = 0; s < p.splits; ++s) {
    const float* ml = p.ml_part + ((size_t)s * p.Nq + q) * 2;
    const float w = exp2f(ml[0] - mstar);
    L += w * ml[1];
    acc += w * p.o_part[((size_t)s * p.Nq + q) * D + d];
  }
  p.out[(size_t)q * p.ldout + d] = (half_t)(acc / L);
}
}  // namespace

hipError_t flash256_launch(const Flash256Params& p, hipStream_t stream) {
  if (p.Nq % 128 || p.Nk <= 0 || p.splits <= 0 || (p.ldq & 7) || (p.ldk & 7) || (p.ldvT & 7)) return hipErrorInvalidValue;
  flash256_kernel<<<dim3(p.Nq / 128, p.splits), dim3(256), 0, stream>>>(p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  flash256_combine_kernel<<<dim3(p.Nq), dim3(256), 0, stream>>>(p);
  return hipGetLastError();
}
