// Accumulator-stationary GEMM for long K and N = 576: the fc2 (K = 2304) and output-projection (K = 576) linears of Hiera
// stage 3 (/root/reference/sam2/sam2/modeling/backbones/hieradet.py:79-81 and :123-129),  x32 = res + X . W^T + b.
//

// (576 = 18 tiles) and a wave keeps the whole output row block - Y^T [576 channels x 32 tokens] = 288 accumulator
// registers - for its whole life, while K streams past in chunks of 32:
//   * W chunk [576 x 32 k] = 36 fragment tiles (32 rows x 32 B) = 36 KiB, packed at weight-load time in exactly that
//     order, travels through a 4-slot LDS ring by LDS-DMA (1-KiB pieces = 8 full cache lines, 9 per wave per chunk);
//   * X is read straight from global memory into B-operand registers (lane = token row, 16 B per k-step); the four k-steps
//     of a 64-k pair are issued together, so the four 32-B sectors of each token's 128-B line arrive from one L2 request
//     (the other three hit in L1), one pair ahead of its use;
//   * per chunk 36 MFMAs (Y^T tile t += W(t, ks) . X^T(ks)), W fragments read in double-buffered batches;
//   * nothing is written until the end: + bias + f32 residual, 16-B stores (4 consecutive channels per lane).
// One workgroup = 4 waves = 128 tokens (one wave per SIMD: the accumulators leave no room for a second); every workgroup
// streams all of W from L2 (2.65 MB for fc2) - 31 B/clk per CU at the MFMA rate, at the edge of what L2 delivers.
#include "gemm_ks.h"
#include <cstdlib>

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

static __device__ __forceinline__ const char* sgpr_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

constexpr int NW = 8;                    // waves per workgroup = 8 token blocks of 32 (two waves per SIMD)
constexpr int OTH = 9;                   // output tiles of 32 channels per workgroup: one HALF of N = 576 (144 accumulator registers)
constexpr int NPR = 2 * 2 * OTH;         // 36 fragment pieces per pair of 32-k chunks (64 k = one cache line of every X row)
constexpr int PPW = (NPR + NW - 1) / NW; // 5 pieces per wave per pair ...
constexpr int NPP = PPW * NW;            // ... of a 40-piece slot: 4 zero pieces pad the packed image (uniform vmcnt counts)
constexpr int STAGE_B = NPP * 1024;
constexpr int NST = 3;                   // ring slots: two pairs in flight while one is consumed (120 KiB)
constexpr int FB = 4;

// ABL != 0: timing ablations (wrong results): 1 no LDS-DMA in the loop, 2 no vmcnt wait / barrier, 3 no X loads in the loop,
// 4 no MFMA (fragments only consumed), 5 no LDS fragment reads in the loop
template <int ABL>
__global__ __launch_bounds__(512, 1) void gemm_ks_kernel(const GemmKsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int tok0 = (blockIdx.x * NW + wave) * 32;
  const int half = blockIdx.y;           // channels [288 half, 288 half + 288)
  const int npair = p.K >> 6;

  const char* wp = reinterpret_cast<const char*>(p.wpack) + (size_t)half * npair * STAGE_B;
  const unsigned lane_off = (unsigned)lane * 16u;
  auto issue = [&](int pr) {             // pair min(pr, npair-1) -> ring slot pr % NST (past the end: a slot nobody reads)
    char* sb = smem + (pr % NST) * STAGE_B;
    const char* cb = wp + (size_t)min(pr, npair - 1) * STAGE_B;
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
      const int q = wave + NW * k;
      const char* src = sgpr_ptr(cb + q * 1024);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + lane_off), (lds_ptr_t)(sb + q * 1024), 16, 0, 0);
    }
  };
  // X fragments of pair pr (k = 64 pr .. 64 pr + 63): fragment 2 c + ks <-> chunk c, k-step ks.  The four 16-B loads of a
  // lane cover one 128-B line of its token row (one L2 request, three L1 hits).
  const half_t* xrow = p.x16 + (size_t)min(tok0 + fr, p.M - 1) * p.ldx + fh * 8;
  struct XP { half8 f[4]; };
  auto load_x = [&](int pr) {
    const half_t* xp = xrow + (size_t)min(pr, npair - 1) * 64;
    XP x;
#pragma unroll
    for (int j = 0; j < 4; ++j) x.f[j] = *reinterpret_cast<const half8*>(xp + 16 * j);
    return x;
  };
  const int rd_w = fr * 32 + ((fh ^ ((fr >> 3) & 1)) << 4);

  f32x16 acc[OTH];
#pragma unroll
  for (int t = 0; t < OTH; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // vector-memory order per pair and wave: [X fragments (4)] [pieces (5)]; two pairs ahead of the one being consumed
  XP xc = load_x(0);
  issue(0);
  XP x1 = load_x(1);
  issue(1);
#pragma nounroll
  for (int pr = 0; pr < npair; ++pr) {
    if (ABL != 2) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + PPW) : "memory");   // pair pr landed; pair pr+1 (the 9 youngest) stays in flight
      __builtin_amdgcn_s_barrier();                       // ... for every wave; everyone is past pair pr-1 -> its slot is free
    }
    const XP x2 = (ABL == 3) ? x1 : load_x(pr + 2);
    if (ABL != 1) issue(pr + 2);
    const char* sW = smem + (pr % NST) * STAGE_B + rd_w;
    // 36 MFMAs: fragment f = c * 18 + 2 tt + ks = LDS piece f; batches of FB, double-buffered
    constexpr int NB = NPR / FB;
    half8 cur[FB], nxt[FB];
#pragma unroll
    for (int j = 0; j < FB; ++j) cur[j] = *reinterpret_cast<const half8*>(sW + j * 1024);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int j = 0; j < FB; ++j)
        if (b + 1 < NB) { if (ABL != 5) nxt[j] = *reinterpret_cast<const half8*>(sW + ((b + 1) * FB + j) * 1024); else nxt[j] = cur[j]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < FB; ++j) {
        const int f = b * FB + j, c = f / (2 * OTH), g = f % (2 * OTH);
        if (ABL != 4) acc[g >> 1] = mfma32(cur[j], xc.f[2 * c + (g & 1)], acc[g >> 1]);
        else asm volatile("" ::"v"(cur[j]), "v"(xc.f[2 * c + (g & 1)]));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < FB; ++j) cur[j] = nxt[j];
    }
    xc = x1;
    x1 = x2;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // trailing duplicate pieces land before the LDS is released

  // ---- out = acc + bias (+ residual); accumulator row = channel 288 half + 32 t + 8 g + 4 fh + e, column = token fr
  const int tok = tok0 + fr;
  if (tok < p.M) {
    float* orow = p.out32 + (size_t)tok * p.ld32;
    const bool has_res = p.res != nullptr;                  // wave-uniform
    const float* rrow = p.res + (size_t)tok * p.ldres;
#pragma unroll
    for (int t = 0; t < OTH; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = 32 * (OTH * half + t) + 8 * g + 4 * fh;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + c0);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = acc[t][4 * g + e] + bv[e];
        if (has_res) {
          const f32x4 rv = *reinterpret_cast<const f32x4*>(rrow + c0);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += rv[e];
        }
        *reinterpret_cast<f32x4*>(orow + c0) = o;
      }
  }
}

// one thread per 16-B unit of the packed image [channel half][pair of chunks][40 pieces]: piece c * 18 + 2 tt + ks holds rows
// n = 288 half + 32 tt + row, k = 64 pr + 32 c + 16 ks + 8 h .. + 7 at byte 32 row + 16 (h ^ ((row >> 3) & 1)); pieces 36..39 zero
__global__ void gemm_ks_pack_kernel(const half_t* __restrict__ w, int K, int ldw, half_t* __restrict__ out, long units) {
  const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= units) return;
  const long piece = u >> 6;
  const int l = (int)(u & 63), row = l >> 1, h = (l & 1) ^ ((row >> 3) & 1);
  const int npair = K >> 6;
  const int q = (int)(piece % NPP);
  const long hp = piece / NPP;
  const int pr = (int)(hp % npair), half = (int)(hp / npair);
  half8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (half_t)0.f;
  if (q < NPR) {
    const int c = q / (2 * OTH), g = q % (2 * OTH), tt = g >> 1, ks = g & 1;
    v = *reinterpret_cast<const half8*>(w + (size_t)(288 * half + 32 * tt + row) * ldw + 64 * pr + 32 * c + 16 * ks + 8 * h);
  }
  *reinterpret_cast<half8*>(out + u * 8) = v;
}
}  // namespace

hipError_t gemm_ks_init() {
  hipError_t e = hipSuccess;
  const void* fns[6] = {(const void*)&gemm_ks_kernel<0>, (const void*)&gemm_ks_kernel<1>, (const void*)&gemm_ks_kernel<2>,
                        (const void*)&gemm_ks_kernel<3>, (const void*)&gemm_ks_kernel<4>, (const void*)&gemm_ks_kernel<5>};
  for (int i = 0; i < 6 && e == hipSuccess; ++i) e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE_B);
  return e;
}
bool gemm_ks_supported(int N, int K) { return N == 2 * 32 * OTH && K >= 128 && (K & 63) == 0; }
size_t gemm_ks_pack_bytes(int N, int K) { return gemm_ks_supported(N, K) ? (size_t)2 * (K / 64) * STAGE_B : 0; }

hipError_t gemm_ks_pack(const half_t* w, int N, int K, int ldw, half_t* wpack, hipStream_t s) {
  const long units = (long)(gemm_ks_pack_bytes(N, K) / 16);
  if (units == 0) return hipErrorInvalidValue;
  gemm_ks_pack_kernel<<<dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s>>>(w, K, ldw, wpack, units);
  return hipGetLastError();
}

hipError_t gemm_ks_launch(const GemmKsParams& p, hipStream_t s) {
  if (p.M <= 0) return hipSuccess;
  if (!gemm_ks_supported(2 * 32 * OTH, p.K) || (p.ldx & 7) || (p.ld32 & 3) || (p.res && (p.ldres & 3)) || !p.bias || !p.out32) return hipErrorInvalidValue;
  static const int abl = getenv("SAM2MI_KS_ABL") ? atoi(getenv("SAM2MI_KS_ABL")) : 0;      // tuning aid
  const dim3 grid((p.M + 32 * NW - 1) / (32 * NW), 2), block(64 * NW);
  switch (abl) {
    case 1: gemm_ks_kernel<1><<<grid, block, NST * STAGE_B, s>>>(p); break;
    case 2: gemm_ks_kernel<2><<<grid, block, NST * STAGE_B, s>>>(p); break;
    case 3: gemm_ks_kernel<3><<<grid, block, NST * STAGE_B, s>>>(p); break;
    case 4: gemm_ks_kernel<4><<<grid, block, NST * STAGE_B, s>>>(p); break;
    case 5: gemm_ks_kernel<5><<<grid, block, NST * STAGE_B, s>>>(p); break;
    default: gemm_ks_kernel<0><<<grid, block, NST * STAGE_B, s>>>(p);
  }
  return hipGetLastError();
}
