// Single-head, head_dim 256 flash attention for the memory attention (self- and cross-attention
// of RoPEAttention, /root/reference/sam2/sam2/modeling/sam/transformer.py:345-424: 1 head, d=256,
// q = 4096 tokens, kv = 4096 (self) or L*4096+P <= 28736 (memory bank + object pointers)).
//
// 4096 queries are only 128 wave tiles, so the KV axis is split across workgroups (grid.y) and a
// combine pass merges the partial (m, l, O).  Workgroup = 4 waves = 128 queries; the 4 waves share
// each 32-key K tile and V^T tile through LDS.  Per wave: S^T = K.Q^T (16 MFMA k-steps over d),
// in-register online softmax (column = query on the lane), O^T += V^T.P^T with P^T taken straight
// from the S^T accumulator (8 row tiles x 2 k-steps).
//
// Q arrives PRE-SCALED by head_dim^-0.5 * log2(e) (folded into the q-projection GEMM epilogue in f32), so
// scores are already in the exp2 domain.  K/V^T tiles are staged by LDS-DMA (global_load_lds_dwordx4) into a
// double buffer: tile t+1 is in flight while tile t is consumed; one barrier per tile.  LDS images are
// linear with the bank-conflict swizzle on the DMA source address:
//   K   [32 keys][32 x 16-B chunks]: slot pc of key row r holds chunk pc ^ (r & 15)   (ds_read_b128, conflict-free)
//   V^T [256 d  ][ 4 x 16-B chunks]: slot pc of row d holds chunk pc ^ ((d >> 2) & 3) (ds_read_b64, 2-way)
// The O accumulator is rescaled only when some query of the wave saw a new running maximum.
#include "attn.h"

namespace {
constexpr int D = 256;
constexpr int K_TILE_B = 32 * 512;       // bytes
constexpr int V_TILE_B = D * 64;         // bytes
constexpr int STAGE_B = K_TILE_B + V_TILE_B;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__global__ __launch_bounds__(256, 2) void flash256_kernel(const Flash256Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int split = blockIdx.y;
  const int ntiles = (p.Nk + 31) / 32;
  const int per = (ntiles + p.splits - 1) / p.splits;
  const int t_lo = split * per, t_hi = min(ntiles, t_lo + per);

  // ---- LDS-DMA sources (per lane), hoisted: 4 K pieces (2 key rows each) + 4 V^T pieces (16 d rows each) per wave
  int k_src[4], v_src[4];          // element offsets (fit 32 bits)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int kp = wave * 4 + j;                       // K piece 0..15
    const int krow = kp * 2 + (lane >> 5), kpc = lane & 31;
    k_src[j] = krow * p.ldk + ((kpc ^ (krow & 15)) << 3);
    const int vp = wave * 4 + j;                       // V^T piece 0..15
    const int vrow = vp * 16 + (lane >> 2), vpc = lane & 3;
    v_src[j] = vrow * p.ldvT + ((vpc ^ ((vrow >> 2) & 3)) << 3);
  }
  auto issue = [&](int stage, int tile) {
    char* sb = smem + stage * STAGE_B;
    const half_t* kb = p.k + (size_t)tile * 32 * p.ldk;
    const half_t* vb = p.vT + tile * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(kb + k_src[j]), (lds_ptr_t)(sb + (wave * 4 + j) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(vb + v_src[j]), (lds_ptr_t)(sb + K_TILE_B + (wave * 4 + j) * 1024), 16, 0, 0);
    }
  };

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

  if (t_lo < t_hi) issue(0, t_lo);
  __syncthreads();
#pragma nounroll
  for (int tile = t_lo; tile < t_hi; ++tile) {
    const int cur = (tile - t_lo) & 1;
    if (tile + 1 < t_hi) issue(cur ^ 1, tile + 1);
    const char* sK = smem + cur * STAGE_B;
    const char* sV = sK + K_TILE_B;
    const int k0 = tile * 32;
    // ---- S^T = K Q^T (already in the exp2 domain)
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const half8 kf = *reinterpret_cast<const half8*>(sK + fr * 512 + (((2 * ks + fh) ^ (fr & 15)) << 4));
      s = mfma32(kf, qf[ks], s);
    }
    const bool tail = k0 + 32 > p.Nk;                  // wave-uniform: only the last tile can be partial
    float tmax = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (tail && k0 + acc_row(r, lane) >= p.Nk) s[r] = -1e30f;
      tmax = fmaxf(tmax, s[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    float psum = 0.f;
    half8 pf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __builtin_amdgcn_exp2f(s[r] - m_new);
      psum += pv;
      pf[r >> 3][r & 7] = (half_t)pv;
    }
    psum += __shfl_xor(psum, 32, 64);
    if (__any(m_new > m_run)) {                        // rescale only when some query got a new maximum
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      m_run = m_new;
    }
    l_run += psum;
    // ---- O^T += V^T P^T
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int row = t * 32 + fr;
      const char* vr = sV + row * 64;
      const int sw = (row >> 2) & 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        // keys 16 ks + 4 fh + {0..3} and + 8: 8-B granules g = 4 ks + fh and g + 2 -> 16-B chunk g >> 1, half g & 1
        const int g0 = 4 * ks + fh, g1 = g0 + 2;
        const half4 lo = *reinterpret_cast<const half4*>(vr + ((((g0 >> 1) ^ sw) << 4) | ((g0 & 1) << 3)));
        const half4 hi = *reinterpret_cast<const half4*>(vr + ((((g1 >> 1) ^ sw) << 4) | ((g1 & 1) << 3)));
        const half8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[t] = mfma32(vf, pf[ks], o[t]);
      }
    }
    __syncthreads();            // retires tile+1 (vmcnt(0)) and frees `cur`
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

// 64 threads per query (4 channels each), 4 queries per workgroup
__global__ __launch_bounds__(256) void flash256_combine_kernel(const Flash256Params p) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6), d = (threadIdx.x & 63) * 4;
  float mstar = -1e30f;
  for (int s = 0; s < p.splits; ++s) mstar = fmaxf(mstar, p.ml_part[((size_t)s * p.Nq + q) * 2]);
  float L = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < p.splits; ++s) {
    const float* ml = p.ml_part + ((size_t)s * p.Nq + q) * 2;
    const float w = exp2f(ml[0] - mstar);
    L += w * ml[1];
    const f32x4 v = *reinterpret_cast<const f32x4*>(p.o_part + ((size_t)s * p.Nq + q) * D + d);
    acc[0] += w * v[0]; acc[1] += w * v[1]; acc[2] += w * v[2]; acc[3] += w * v[3];
  }
  const float inv = 1.f / L;
  const half4 h = {(half_t)(acc[0] * inv), (half_t)(acc[1] * inv), (half_t)(acc[2] * inv), (half_t)(acc[3] * inv)};
  *reinterpret_cast<half4*>(p.out + (size_t)q * p.ldout + d) = h;
}
}  // namespace

hipError_t flash256_init() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&flash256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_B);
}

hipError_t flash256_launch(const Flash256Params& p, hipStream_t stream) {
  if (p.Nq % 128 || p.Nk <= 0 || p.splits <= 0 || (p.ldq & 7) || (p.ldk & 7) || (p.ldvT & 7) || (p.ldout & 3)) return hipErrorInvalidValue;
  flash256_kernel<<<dim3(p.Nq / 128, p.splits), dim3(256), 2 * STAGE_B, stream>>>(p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  flash256_combine_kernel<<<dim3(p.Nq / 4), dim3(256), 0, stream>>>(p);
  return hipGetLastError();
}
