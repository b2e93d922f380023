// Tiny fp32 multi-head attention for the two-way mask decoder
// (sam/transformer.py:222-294: 8 heads, head_dim 32 (self) or 16 (cross), tokens T <= 64 on one side).
// One wave per (batch, head, query); lanes stride over the keys with a private online softmax and
// the 64 partial states are merged with wave shuffles.
#include "attn.h"

namespace {
template <int HD>
__global__ __launch_bounds__(256) void small_attn_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                         const float* __restrict__ v, int ldv, float* __restrict__ out, int ldo,
                                                         int Tq, int Tk, int heads, int batch, size_t q_bs, size_t kv_bs, size_t o_bs) {
  const long task = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (task >= (long)batch * heads * Tq) return;
  const int qi = (int)(task % Tq);
  const int h = (int)((task / Tq) % heads);
  const int b = (int)(task / ((long)Tq * heads));
  const float* qp = q + b * q_bs + (size_t)qi * ldq + h * HD;
  float qr[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) qr[d] = qp[d];
  const float scale = rsqrtf((float)HD);
  float m = -1e30f, l = 0.f, o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
  for (int j = lane; j < Tk; j += 64) {
    const float* kp = k + b * kv_bs + (size_t)j * ldk + h * HD;
    const float* vp = v + b * kv_bs + (size_t)j * ldv + h * HD;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s += qr[d] * kp[d];
    s *= scale;
    const float mn = fmaxf(m, s);
    const float a = __expf(m - mn), pj = __expf(s - mn);
    l = l * a + pj;
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = o[d] * a + pj * vp[d];
    m = mn;
  }
  const float mstar = wave_max(m);
  const float w = __expf(m - mstar);          // lanes without keys: exp(-1e30 - m*) = 0
  const float L = wave_sum(l * w);
  float* op = out + b * o_bs + (size_t)qi * ldo + h * HD;
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    const float r = wave_sum(o[d] * w);
    if (lane == 0) op[d] = r / L;
  }
}

// ---- token -> image attention: few queries (T <= 64), 4096+ keys, head_dim 16.
// One 1024-thread workgroup per (query, head): 16 waves sweep 1/16 of the keys each with float4 loads,
// their online-softmax states are merged through LDS.
__global__ __launch_bounds__(1024) void t2i_attn_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                        const float* __restrict__ v, int ldv, float* __restrict__ out, int ldo,
                                                        int Tq, int Tk, int heads, size_t q_bs, size_t kv_bs, size_t o_bs) {
  constexpr int HD = 16;
  __shared__ float sm[16], sl[16], so[16][HD];
  const int qi = blockIdx.x % Tq, h = blockIdx.x / Tq;
  q += blockIdx.y * q_bs; k += blockIdx.y * kv_bs; v += blockIdx.y * kv_bs; out += blockIdx.y * o_bs;     // batch
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x4 qr[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) qr[d] = *reinterpret_cast<const f32x4*>(q + (size_t)qi * ldq + h * HD + 4 * d);
  float m = -1e30f, l = 0.f, o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
  for (int j = tid; j < Tk; j += 1024) {
    const float* kp = k + (size_t)j * ldk + h * HD;
    const float* vp = v + (size_t)j * ldv + h * HD;
    float s = 0.f;
    f32x4 vv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const f32x4 kk = *reinterpret_cast<const f32x4*>(kp + 4 * d);
      vv[d] = *reinterpret_cast<const f32x4*>(vp + 4 * d);
      s += qr[d][0] * kk[0] + qr[d][1] * kk[1] + qr[d][2] * kk[2] + qr[d][3] * kk[3];
    }
    s *= 0.25f;                                   // 1 / sqrt(16)
    const float mn = fmaxf(m, s);
    const float a = __expf(m - mn), pj = __expf(s - mn);
    l = l * a + pj;
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = o[d] * a + pj * vv[d >> 2][d & 3];
    m = mn;
  }
  const float mw = wave_max(m);
  const float w = __expf(m - mw);
  const float lw = wave_sum(l * w);
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    const float r = wave_sum(o[d] * w);
    if (lane == 0) so[wave][d] = r;
  }
  if (lane == 0) { sm[wave] = mw; sl[wave] = lw; }
  __syncthreads();
  if (tid < HD) {
    float ms = -1e30f;
    for (int i = 0; i < 16; ++i) ms = fmaxf(ms, sm[i]);
    float L = 0.f, acc = 0.f;
    for (int i = 0; i < 16; ++i) {
      const float ww = __expf(sm[i] - ms);
      L += ww * sl[i];
      acc += ww * so[i][tid];
    }
    out[(size_t)qi * ldo + h * HD + tid] = acc / L;
  }
}

// ---- token -> image attention, all queries of a prompt at once (T <= 8, head_dim 16, Tk a multiple of 512).
// The per-(query, head) kernel above reads every key / value row once PER QUERY (8 x 4 MB per prompt from L2, in 64-byte pieces).
// Here a 512-thread workgroup takes one head and 512 keys of one prompt: thread = key (K row -> T scores, V row -> LDS), wave = query
// (soft-max over the 512 scores of its query, in LDS), then thread = (query, channel, key quarter) for the P.V sum.  The key splits are
// merged by t2i_merge_kernel (running max / sum per split, like the flash partials).  part: [batch, heads, S, T, 18] = o[16], m, l.
constexpr int T2I_KEYS = 512, T2I_TMAX = 8;
__global__ __launch_bounds__(T2I_KEYS) void t2i_part_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                            const float* __restrict__ v, int ldv, float* __restrict__ part, int Tq, int heads,
                                                            int splits, size_t q_bs, size_t kv_bs) {
  constexpr int HD = 16;
  __shared__ __attribute__((aligned(16))) float sq[T2I_TMAX][HD];
  __shared__ __attribute__((aligned(16))) float sS[T2I_TMAX][T2I_KEYS];
  __shared__ __attribute__((aligned(16))) float sV[T2I_KEYS * HD];
  const int h = blockIdx.x % heads, sp = blockIdx.x / heads;
  q += blockIdx.y * q_bs; k += blockIdx.y * kv_bs; v += blockIdx.y * kv_bs;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < Tq * HD) sq[tid / HD][tid % HD] = q[(size_t)(tid / HD) * ldq + h * HD + tid % HD] * 0.25f;      // 1 / sqrt(16) folded in
  const size_t key = (size_t)sp * T2I_KEYS + tid;
  f32x4 kk[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    kk[d] = *reinterpret_cast<const f32x4*>(k + key * ldk + h * HD + 4 * d);
    *reinterpret_cast<f32x4*>(sV + tid * HD + 4 * d) = *reinterpret_cast<const f32x4*>(v + key * ldv + h * HD + 4 * d);
  }
  __syncthreads();
  for (int t = 0; t < Tq; ++t) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const f32x4 qq = *reinterpret_cast<const f32x4*>(&sq[t][4 * d]);
      s += qq[0] * kk[d][0] + qq[1] * kk[d][1] + qq[2] * kk[d][2] + qq[3] * kk[d][3];
    }
    sS[t][tid] = s;
  }
  __syncthreads();
  float* pp = part + ((((size_t)blockIdx.y * heads + h) * splits + sp) * Tq + wave) * 18;
  if (wave < Tq) {                                   // wave-uniform: wave w owns query w
    float sv[8], m = -1e30f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { sv[j] = sS[wave][lane + 64 * j]; m = fmaxf(m, sv[j]); }
    m = wave_max(m);
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float e = __expf(sv[j] - m); sS[wave][lane + 64 * j] = e; l += e; }
    l = wave_sum(l);
    if (lane == 0) { pp[16] = m; pp[17] = l; }
  }
  __syncthreads();
  if (wave < Tq) {
    const int d = lane & 15, kq = lane >> 4;         // keys 4 i + kq: the 64 lanes read 64 consecutive floats of sV
    float acc = 0.f;
#pragma unroll 8
    for (int i = 0; i < T2I_KEYS / 4; ++i) acc = fmaf(sS[wave][4 * i + kq], sV[(4 * i + kq) * HD + d], acc);
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    if (lane < HD) pp[lane] = acc;
  }
}

// one thread per (query, head, channel) of a prompt: out = sum_s w_s o_s / sum_s w_s l_s, w_s = exp(m_s - max m)
__global__ void t2i_merge_kernel(const float* __restrict__ part, float* __restrict__ out, int ldo, int Tq, int heads, int splits, size_t o_bs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Tq * heads * 16) return;
  const int d = i & 15, h = (i >> 4) % heads, t = i / (16 * heads);
  const float* pb = part + (((size_t)blockIdx.y * heads + h) * splits * Tq + t) * 18;
  float m = -1e30f;
  for (int s = 0; s < splits; ++s) m = fmaxf(m, pb[(size_t)s * Tq * 18 + 16]);
  float acc = 0.f, L = 0.f;
  for (int s = 0; s < splits; ++s) {
    const float* ps = pb + (size_t)s * Tq * 18;
    const float w = __expf(ps[16] - m);
    acc += w * ps[d];
    L += w * ps[17];
  }
  out[blockIdx.y * o_bs + (size_t)t * ldo + h * 16 + d] = acc / L;
}

// ---- image -> token attention: 4096 queries, few keys (Tk <= 64), head_dim 16: one thread per (query, head),
// keys/values of the head broadcast from LDS.
__global__ __launch_bounds__(256) void i2t_attn_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                       const float* __restrict__ v, int ldv, float* __restrict__ out, int ldo,
                                                       int Tq, int Tk, int heads, size_t q_bs, size_t kv_bs, size_t o_bs, half_t* out16, size_t lo_off) {
  constexpr int HD = 16;
  __shared__ float sk[64 * 128], sv[64 * 128];           // [Tk][heads*HD], heads*HD <= 128
  const int C = heads * HD;
  q += blockIdx.y * q_bs; k += blockIdx.y * kv_bs; v += blockIdx.y * kv_bs;     // batch
  if (out16) out16 += blockIdx.y * o_bs; else out += blockIdx.y * o_bs;
  for (int i = threadIdx.x; i < Tk * C; i += 256) {
    sk[i] = k[(size_t)(i / C) * ldk + i % C];
    sv[i] = v[(size_t)(i / C) * ldv + i % C];
  }
  __syncthreads();
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Tq * heads) return;
  const int h = idx % heads, qi = idx / heads;
  f32x4 qr[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) qr[d] = *reinterpret_cast<const f32x4*>(q + (size_t)qi * ldq + h * HD + 4 * d);
  float m = -1e30f, l = 0.f, o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
  for (int j = 0; j < Tk; ++j) {
    const float* kp = sk + j * C + h * HD;
    const float* vp = sv + j * C + h * HD;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s += qr[d >> 2][d & 3] * kp[d];
    s *= 0.25f;
    const float mn = fmaxf(m, s);
    const float a = __expf(m - mn), pj = __expf(s - mn);
    l = l * a + pj;
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = o[d] * a + pj * vp[d];
    m = mn;
  }
  const float inv = 1.f / l;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const f32x4 r = {o[4 * d] * inv, o[4 * d + 1] * inv, o[4 * d + 2] * inv, o[4 * d + 3] * inv};
    if (out16) store_h4(out16 + (size_t)qi * ldo + h * HD + 4 * d, lo_off, r);         // the MFMA operand of the out-projection, directly
    else *reinterpret_cast<f32x4*>(out + (size_t)qi * ldo + h * HD + 4 * d) = r;
  }
}
}  // namespace

hipError_t small_attn_launch(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo,
                             int Tq, int Tk, int heads, int hd, int batch, size_t q_bstride, size_t kv_bstride,
                             size_t o_bstride, hipStream_t stream, float* scratch, size_t scratch_floats, half_t* out16, size_t out16_lo_off) {
  if (out16 && !(hd == 16 && batch <= 65535 && !(ldq & 3) && !(ldk & 3) && !(ldv & 3) && !(ldo & 3) && Tk <= 64 && heads * 16 <= 128 && Tq >= 1024))
    return hipErrorInvalidValue;                          // f16 output: the image -> token kernel only
  const long tasks = (long)batch * heads * Tq;
  const dim3 grid((unsigned)((tasks + 3) / 4)), block(256);
  const bool al = !(ldq & 3) && !(ldk & 3) && !(ldv & 3) && !(ldo & 3);
  if (hd == 16 && batch <= 65535 && al && Tq <= T2I_TMAX && Tk >= 1024 && Tk % T2I_KEYS == 0 && scratch &&
      (size_t)batch * heads * (Tk / T2I_KEYS) * Tq * 18 <= scratch_floats) {
    const int splits = Tk / T2I_KEYS;
    t2i_part_kernel<<<dim3(heads * splits, batch), dim3(T2I_KEYS), 0, stream>>>(q, ldq, k, ldk, v, ldv, scratch, Tq, heads, splits, q_bstride, kv_bstride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int outs = Tq * heads * 16;
    t2i_merge_kernel<<<dim3((outs + 255) / 256, batch), dim3(256), 0, stream>>>(scratch, out, ldo, Tq, heads, splits, o_bstride);
    return hipGetLastError();
  }
  if (hd == 16 && batch <= 65535 && al && Tq <= 64 && Tk >= 1024) {
    t2i_attn_kernel<<<dim3(Tq * heads, batch), dim3(1024), 0, stream>>>(q, ldq, k, ldk, v, ldv, out, ldo, Tq, Tk, heads, q_bstride,
                                                                        kv_bstride, o_bstride);
    return hipGetLastError();
  }
  if (hd == 16 && batch <= 65535 && al && Tk <= 64 && heads * 16 <= 128 && Tq >= 1024) {
    i2t_attn_kernel<<<dim3((Tq * heads + 255) / 256, batch), dim3(256), 0, stream>>>(q, ldq, k, ldk, v, ldv, out, ldo, Tq, Tk, heads,
                                                                                     q_bstride, kv_bstride, o_bstride, out16, out16_lo_off);
    return hipGetLastError();
  }
  if (hd == 16)
    small_attn_kernel<16><<<grid, block, 0, stream>>>(q, ldq, k, ldk, v, ldv, out, ldo, Tq, Tk, heads, batch, q_bstride, kv_bstride, o_bstride);
  else if (hd == 32)
    small_attn_kernel<32><<<grid, block, 0, stream>>>(q, ldq, k, ldk, v, ldv, out, ldo, Tq, Tk, heads, batch, q_bstride, kv_bstride, o_bstride);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}
