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
}  // namespace

hipError_t small_attn_launch(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo,
                             int Tq, int Tk, int heads, int hd, int batch, size_t q_bstride, size_t kv_bstride,
                             size_t o_bstride, hipStream_t stream) {
  const long tasks = (long)batch * heads * Tq;
  const dim3 grid((unsigned)((tasks + 3) / 4)), block(256);
  if (hd == 16)
    small_attn_kernel<16><<<grid, block, 0, stream>>>(q, ldq, k, ldk, v, ldv, out, ldo, Tq, Tk, heads, batch, q_bstride, kv_bstride, o_bstride);
  else if (hd == 32)
    small_attn_kernel<32><<<grid, block, 0, stream>>>(q, ldq, k, ldk, v, ldv, out, ldo, Tq, Tk, heads, batch, q_bstride, kv_bstride, o_bstride);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}
