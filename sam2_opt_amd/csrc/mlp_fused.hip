// Fused Hiera MLP for gfx950:  x += fc2( GELU( fc1( LN2(x) ) ) )  in ONE kernel, the 4C-wide hidden activation never
// leaves the CU.  Reference: MultiScaleBlock.forward, /root/reference/sam2/sam2/modeling/backbones/hieradet.py:163-165
// (x = x + drop_path(self.mlp(self.norm2(x)))), MLP = Linear(C, 4C) -> GELU(erf) -> Linear(4C, C) (:123-129).
//
// Why: as two GEMMs the hidden tensor [M, 4C] f16 is written and re-read through HBM/L2 (604 MB each way per launch in
// stage 1) and both GEMMs re-fetch their operand tiles from L2 N/128 resp. M/128 times; stages 1-2 (C = 144 / 288) ran
// at 200-400 TFLOP/s for that reason.  Here the structure is that of a flash-attention kernel without the softmax state:
//     "Q" = a wave's 32 (or 64) token rows of LN2(x), register resident as MFMA B operands      (C/16 fragments)
//     "K" = W1 chunk  [32 hidden units, C]   ->  S^T = W1_chunk . X^T                       (C/16 MFMA k-steps)
//     "P" = GELU(S^T + b1) as f16, taken straight from the S^T accumulator as the next B operand
//     "V^T" = W2 chunk [C out channels, 32 hidden units]  ->  Y^T += W2_chunk . P           (C/32 tiles x 2 k-steps)
// W1 rows are read PERMUTED (hidden unit pi23(i) on MFMA row i, bits 2 and 3 swapped) so that the accumulator registers a
// lane owns are exactly hidden units 16 ks + 8 fh + 0..7 in B-operand order (same trick as flash256_v3, attn_flash256.hip).
// Weights travel through a 4-slot LDS ring by LDS-DMA in 1-KiB pieces that ARE fragment tiles (32 rows x 32 B), from an image
// PACKED at weight-load time in exactly that order (mlp_fused_pack): every piece is 8 full cache lines, its source is
// wave-uniform base + 16 * lane, and a fragment read is slot + piece * 1024 + lane constant (immediate offsets).  The two
// 16-B halves of a row are swapped on rows with bit 3 set (in the packed image), which makes every 16-lane ds_read_b128
// group conflict-free.
// One workgroup = 4 waves (one per SIMD, up to 512 registers) = 128 * TN tokens; all workgroups stream the same weights
// (L2 resident): L2->LDS traffic per token is 16 C^2 / (128 TN) bytes instead of 2 x (A + W tiles) per 128x128 GEMM tile.
// b1 is added in front of the GELU (LDS table); b2 and the f32 residual are added when Y^T is written back (16-B stores:
// 4 consecutive channels per lane).  The chunk loop is software-pipelined (see mlp_fused_kernel).
#include "mlp_fused.h"
#include <cstdlib>

#ifndef VSLICE
#define VSLICE 10
#endif

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

static __device__ __forceinline__ int pi23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

// Keeps a wave-uniform pointer in an SGPR pair and opaque to loop strength reduction, so that base + (32-bit lane offset)
// selects the scalar-base form of global_load_lds (otherwise the loop carries one 64-bit VGPR address per piece).
static __device__ __forceinline__ const char* sgpr_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

template <int N>
static __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int C, int TN, int NST>
struct MlpCfg {
  static constexpr int NW = 4;                      // one wave per SIMD
  static constexpr int KS = C / 16;                 // fc1 k-steps
  static constexpr int OT = (C + 31) / 32;          // output-channel tiles
  static constexpr int H4 = 4 * C;
  static constexpr int NCH = H4 / 32;               // chunks of 32 hidden units = ring stages to stream
  static constexpr int W1P = KS, W2P = OT * 2, NP = W1P + W2P;
  static constexpr int PPW = (NP + NW - 1) / NW;    // pieces per wave per stage (uniform: counted vmcnt)
  static constexpr int NPP = PPW * NW;              // pieces per chunk in the packed weight image (zero pieces pad NP up to a multiple of 4)
  static constexpr int STAGE_B = NPP * 1024;
  static constexpr int LDS_B = NST * STAGE_B + H4 * 4;
  static_assert(C % 16 == 0 && NST >= 3, "shape");
  static_assert((NST - 1) * PPW <= 63, "vmcnt range");
};

// erf-GELU without the |x| clamp of gelu_erf_fast (common.h): for finite x the exponential underflows long before
// |x| * q can overflow, and the operands here are finite (LayerNorm output times weights).  relu(x) = med3(x, 0, inf):
// fmaxf would cost a second v_max for sNaN quieting.  Written per element; the compiler packs pairs into v_pk_* ops.
static __device__ __forceinline__ float gelu_nc(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.23164189f, 1.0f));
  const float m = x * 0.84932180f;
  const float e = __builtin_amdgcn_exp2f(-(m * m));
  float poly = fmaf(0.5307027145f, t, -0.7265760135f);
  poly = fmaf(poly, t, 0.7107068705f);
  poly = fmaf(poly, t, -0.142248368f);
  poly = fmaf(poly, t, 0.127414796f);
  return fmaf(-ax, poly * t * e, __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_inff()));
}

// Software pipeline over the 4C / 32 hidden chunks (one wave per SIMD, the flash256_v3 scheme):
//   iteration j:   S_{j+1} = W1_{j+1} X^T  [MFMA]   beside   P_j = GELU(S_j + b1_j)  [VALU]   beside the LDS-DMA of chunk j+3
//                  Y^T += W2_j P_j          [MFMA]
// All fragments of both products are read from LDS at the top of the iteration (one wave per SIMD has the registers), the
// first phase is laid out with sched_group_barrier as (1 MFMA, 1 DMA piece, a slice of the VALU work) groups.
// C = 144 with ONE token tile per wave (TN = 1, 3-slot ring): 80 + 36 accumulator / operand registers instead of 160 + 72, so the kernel
// fits 256 registers and 62 KB of LDS and TWO workgroups share a CU - two waves per SIMD with independent barriers, i.e. the GELU of one
// beside the MFMA chains of the other (with one wave per SIMD the two phases of an iteration can only overlap inside the wave's own
// instruction stream).  The weights stream twice as often (128 tokens per workgroup), which the L2 -> LDS path has room for at C = 144.
template <int C, int TN, int NST>
__global__ __launch_bounds__(256, (C == 144 && TN == 1) ? 2 : 1) void mlp_fused_kernel(const MlpFusedParams p) {
  using K = MlpCfg<C, TN, NST>;
  constexpr int NW = K::NW, KS = K::KS, OT = K::OT, H4 = K::H4, NCH = K::NCH, W1P = K::W1P, NP = K::NP, PPW = K::PPW, STAGE_B = K::STAGE_B;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bias_lds = reinterpret_cast<float*>(smem + NST * STAGE_B);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int tok0 = (blockIdx.x * NW + wave) * (32 * TN);

  // X fragments (B operand): X[token fr][16 s + 8 fh + j]; rows past M are clamped (their results are not stored)
  half8 xf[TN][KS];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const size_t row = (size_t)min(tok0 + 32 * tn + fr, p.M - 1);
    if (p.ln_eps > 0.f) {                    // wave-uniform: LN2 fused into the load - X = normalised rows of the residual stream
      ln_row_fragments<KS>(p.x32 + row * p.ld32 + fh * 8, p.ln_eps, xf[tn]);
    } else {
      const half_t* xp = p.x16 + row * p.ldx + fh * 8;
#pragma unroll
      for (int s = 0; s < KS; ++s) xf[tn][s] = *reinterpret_cast<const half8*>(xp + s * 16);
    }
  }
  for (int i = tid; i < H4; i += 256) bias_lds[i] = p.b1[i];
  __syncthreads();                           // bias table visible (also retires the X loads)

  // ---- LDS-DMA from the PACKED weight image (mlp_fused_pack): chunk j = NPP consecutive 1-KiB pieces, each one fragment
  // tile (32 rows x 32 B, halves swapped on rows with bit 3 set) exactly as it sits in LDS.  Every piece is 8 full cache
  // lines (strided 32-B row segments of the nn.Linear layout would cost 32 L2 requests per KiB: measured 7.5 B/clk/CU).
  // Ring step i loads chunk min(i, NCH-1) into slot i % NST: past the end the last chunk is loaded again into a free slot
  // (read only by the discarded S chain of the final iteration), so every iteration issues exactly PPW pieces per wave.
  const char* wp = reinterpret_cast<const char*>(p.wpack);
  const unsigned lane_off = (unsigned)lane * 16u;
  auto issue = [&](int i) {
    char* sb = smem + (i % NST) * STAGE_B;
    const char* cb = wp + (size_t)min(i, NCH - 1) * STAGE_B;
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
      const int q = wave + NW * k;           // wave-uniform
      const char* src = sgpr_ptr(cb + q * 1024);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + lane_off), (lds_ptr_t)(sb + q * 1024), 16, 0, 0);
    }
  };
  // fragment read addresses (bytes inside a piece)
  const int r1 = pi23(fr);
  const int rd_w1 = r1 * 32 + ((fh ^ ((r1 >> 3) & 1)) << 4);
  const int rd_w2 = W1P * 1024 + fr * 32 + ((fh ^ ((fr >> 3) & 1)) << 4);
  struct W1F { half8 f[KS]; };
  struct W2F { half8 f[2 * OT]; };
  auto read_w1 = [&](int i) {
    const char* sb = smem + (i % NST) * STAGE_B + rd_w1;
    W1F w;
#pragma unroll
    for (int s = 0; s < KS; ++s) w.f[s] = *reinterpret_cast<const half8*>(sb + s * 1024);
    return w;
  };
  auto read_w2 = [&](int i) {
    const char* sb = smem + (i % NST) * STAGE_B + rd_w2;
    W2F w;
#pragma unroll
    for (int f = 0; f < 2 * OT; ++f) w.f[f] = *reinterpret_cast<const half8*>(sb + f * 1024);
    return w;
  };
  auto chain = [&](const W1F& w, f32x16 (&sa)[TN]) {           // S^T = W1_chunk X^T (first MFMA starts from the constant 0)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      f32x16 z;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = 0.f;
      sa[tn] = mfma32(w.f[0], xf[tn][0], z);
    }
#pragma unroll
    for (int s = 1; s < KS; ++s)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) sa[tn] = mfma32(w.f[s], xf[tn][s], sa[tn]);
  };

  f32x16 acc[TN][OT];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tn][t][r] = 0.f;

#pragma unroll
  for (int i = 0; i < NST - 1; ++i) issue(i);
  wait_vm<(NST - 2) * PPW>();
  __builtin_amdgcn_s_barrier();
  f32x16 scur[TN];
  {
    const W1F w = read_w1(0);
    __builtin_amdgcn_sched_barrier(0);
    chain(w, scur);
  }

#pragma nounroll
  for (int j = 0; j < NCH; ++j) {
    // chunk j+1 landed (NST-3 younger ones may stay in flight); every wave is past chunk j-1 -> slot (j-1) % NST is free
    wait_vm<(NST - 3) * PPW>();
    __builtin_amdgcn_s_barrier();
    const W1F w1 = read_w1(j + 1);
    const W2F w2 = read_w2(j);
    // b1 of this chunk: accumulator register 8 ks + e <-> hidden unit 32 j + 16 ks + 8 fh + e
    f32x4 bv[4];
    {
      const float* bl = bias_lds + 32 * j + 8 * fh;
      bv[0] = *reinterpret_cast<const f32x4*>(bl); bv[1] = *reinterpret_cast<const f32x4*>(bl + 4);
      bv[2] = *reinterpret_cast<const f32x4*>(bl + 16); bv[3] = *reinterpret_cast<const f32x4*>(bl + 20);
    }
    __builtin_amdgcn_sched_barrier(0);
    issue(j + NST - 1);
    f32x16 snext[TN];
    chain(w1, snext);
    half8 pf[TN][2];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) pf[tn][r >> 3][r & 7] = (half_t)gelu_nc(scur[tn][r] + bv[r >> 2][r & 3]);
#pragma unroll
    for (int g = 0; g < KS * TN; ++g) {                  // phase schedule: 1 MFMA, 1 LDS-DMA piece (while there are any), VALU slice
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (g < PPW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, VSLICE, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- Y^T += W2_chunk P
#pragma unroll
    for (int f = 0; f < 2 * OT; ++f)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tn][f >> 1] = mfma32(w2.f[f], pf[tn][f & 1], acc[tn][f >> 1]);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) scur[tn] = snext[tn];
  }
  wait_vm<0>();                                          // the trailing (duplicate) DMA pieces must land before the LDS is released

  // ---- x += Y + b2   (accumulator row = channel 32 t + 8 g + 4 fh + 0..3, column = token fr)
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int tok = tok0 + 32 * tn + fr;
    if (tok < p.M) {
      float* xr = p.x32 + (size_t)tok * p.ld32;
#pragma unroll
      for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int c0 = 32 * t + 8 * g4 + 4 * fh;
          if (C % 32 == 0 || c0 < C) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(xr + c0);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.b2 + c0);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rv[e] + bv[e] + acc[tn][t][4 * g4 + e];
            *reinterpret_cast<f32x4*>(xr + c0) = o;
          }
        }
    }
  }
}

// one thread per 16-B unit of the packed image
template <int C, int TN, int NST>
__global__ void mlp_pack_kernel(const half_t* __restrict__ w1, const half_t* __restrict__ w2, half_t* __restrict__ out) {
  using K = MlpCfg<C, TN, NST>;
  const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= (long)K::NCH * K::NPP * 64) return;
  const int j = (int)(u / (K::NPP * 64)), rem = (int)(u % (K::NPP * 64));
  const int q = rem >> 6, l = rem & 63, row = l >> 1, h = (l & 1) ^ ((row >> 3) & 1);
  half8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (half_t)0.f;
  if (q < K::W1P) v = *reinterpret_cast<const half8*>(w1 + (size_t)(32 * j + row) * C + 16 * q + 8 * h);
  else if (q < K::NP) {
    const int r = q - K::W1P, t = r >> 1, ks = r & 1, c = 32 * t + row;
    if (c < C) v = *reinterpret_cast<const half8*>(w2 + (size_t)c * K::H4 + 32 * j + 16 * ks + 8 * h);
  }
  *reinterpret_cast<half8*>(out + (size_t)u * 8) = v;
}

template <int C, int TN, int NST>
hipError_t launch_cfg(const MlpFusedParams& p, hipStream_t s) {
  using K = MlpCfg<C, TN, NST>;
  const int wg_tok = 32 * TN * K::NW;
  const int grid = (p.M + wg_tok - 1) / wg_tok;
  mlp_fused_kernel<C, TN, NST><<<dim3(grid), dim3(64 * K::NW), K::LDS_B, s>>>(p);
  return hipGetLastError();
}
template <int C, int TN, int NST>
hipError_t attr_cfg() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_kernel<C, TN, NST>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, MlpCfg<C, TN, NST>::LDS_B);
}
}  // namespace

// Configurations: C = 144: 32 tokens per wave (128 per workgroup), 3-slot ring (62 KiB), 246 registers: two workgroups per CU (the
// default since round 3; before: 64 tokens per wave, 256 per workgroup, 4-slot ring of 82 KiB, 400 registers, one workgroup per CU);
// 19 + 1 padding pieces per chunk;
// C = 288: 32 tokens per wave (128 per workgroup), 36 pieces per chunk, 3-slot ring (113 KiB).
// C = 576 does not fit: 32 tokens need 144 (X) + 288 (Y^T) of the 512 registers and the compiler spills X into scratch
// (reloaded inside the MFMA chains behind vmcnt(0), which also drains the DMA ring) - stage 3 stays on the two-GEMM path.
hipError_t mlp_fused_init() {
  hipError_t e[4] = {attr_cfg<144, 2, 4>(), attr_cfg<288, 1, 4>(), attr_cfg<288, 1, 3>(), attr_cfg<144, 1, 3>()};
  for (int i = 0; i < 4; ++i)
    if (e[i] != hipSuccess) return e[i];
  return hipSuccess;
}

bool mlp_fused_supported(int C) { return C == 144 || C == 288; }

size_t mlp_fused_pack_bytes(int C) {
  if (C == 144) return (size_t)MlpCfg<144, 2, 4>::NCH * MlpCfg<144, 2, 4>::STAGE_B;
  if (C == 288) return (size_t)MlpCfg<288, 1, 4>::NCH * MlpCfg<288, 1, 4>::STAGE_B;
  return 0;
}

hipError_t mlp_fused_pack(const half_t* w1, const half_t* w2, int C, half_t* wpack, hipStream_t s) {
  const long units = (long)(mlp_fused_pack_bytes(C) / 16);
  if (units == 0) return hipErrorInvalidValue;
  const dim3 grid((unsigned)((units + 255) / 256)), block(256);
  if (C == 144) mlp_pack_kernel<144, 2, 4><<<grid, block, 0, s>>>(w1, w2, wpack);
  else mlp_pack_kernel<288, 1, 4><<<grid, block, 0, s>>>(w1, w2, wpack);
  return hipGetLastError();
}

// C = 288: a 3-slot ring (113 KB) runs the kernel at the same speed as 4 slots (149 KB) and leaves LDS for a small workgroup of the
// tracking stream beside it (+0.7 % end to end); SAM2MI_MLP_RING4=1 restores 4 slots
// C = 144: two workgroups per CU (<144, 1, 3>), 462 -> 346 us per stage-1 launch (round 3); SAM2MI_MLP144_2WG=0 restores <144, 2, 4>
static bool two_per_cu() { static const bool v = getenv("SAM2MI_MLP144_2WG") ? atoi(getenv("SAM2MI_MLP144_2WG")) != 0 : true; return v; }
static bool ring4() { static const bool v = getenv("SAM2MI_MLP_RING4") != nullptr; return v; }

const char* mlp_fused_kernel_name(int C) {       // as rocprofv3 prints it
  if (C == 144) return two_per_cu() ? "mlp_fused_kernel<144, 1, 3>" : "mlp_fused_kernel<144, 2, 4>";
  return ring4() ? "mlp_fused_kernel<288, 1, 4>" : "mlp_fused_kernel<288, 1, 3>";
}

hipError_t mlp_fused_launch(const MlpFusedParams& p, int C, hipStream_t s) {
  if (p.M <= 0) return hipSuccess;
  if ((p.ldx & 7) || (p.ld32 & 3)) return hipErrorInvalidValue;
  switch (C) {
    case 144: return two_per_cu() ? launch_cfg<144, 1, 3>(p, s) : launch_cfg<144, 2, 4>(p, s);
    case 288: return ring4() ? launch_cfg<288, 1, 4>(p, s) : launch_cfg<288, 1, 3>(p, s);
    default: return hipErrorInvalidValue;
  }
}
