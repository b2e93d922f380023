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
// Weights travel through an LDS ring by LDS-DMA in 1-KiB pieces that ARE fragment tiles (32 rows x 32 B): the per-lane
// source offset is one constant per operand, the piece base is wave-uniform, and a fragment read is
// stage + piece * 1024 + lane constant (immediate offsets).  The two 16-B halves of a row are swapped on rows with bit 3
// set (on the DMA source address), which makes every 16-lane ds_read_b128 group conflict-free.
// One workgroup = 4 waves (one per SIMD, up to 512 registers) = 128 * TN tokens; all workgroups stream the same weights
// (L2 resident): L2->LDS traffic per token is 16 C^2 / (128 TN) bytes instead of 2 x (A + W tiles) per 128x128 GEMM tile.
// b1 is the initial value of the S^T accumulator; b2 and the f32 residual are added when Y^T is written back (16-B
// stores: 4 consecutive channels per lane).
#include "mlp_fused.h"
#include <cstdlib>

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

static __device__ __forceinline__ int pi23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

// Keeps a wave-uniform pointer in an SGPR pair and opaque to loop strength reduction, so that base + (32-bit lane offset)
// selects the scalar-base form of global_load_lds (otherwise the loop carries one 64-bit VGPR address per piece).
static __device__ __forceinline__ const char* sgpr_ptr(const char* p) {
  unsigned long long v = reinterpret_cast<unsigned long long>(p);
  asm volatile("" : "+s"(v));
  return reinterpret_cast<const char*>(v);
}

template <int N>
static __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int C, int TN, int G, int NST>
struct MlpCfg {
  static constexpr int KS = C / 16;                 // fc1 k-steps
  static constexpr int OT = (C + 31) / 32;          // output-channel tiles
  static constexpr int H4 = 4 * C;
  static constexpr int NSTG = H4 / (32 * G);        // ring stages to stream (G chunks of 32 hidden units each)
  static constexpr int W1P = G * KS, W2P = G * OT * 2, NP = W1P + W2P;
  static constexpr int PW1 = (W1P + 3) / 4, PW2 = (W2P + 3) / 4, PPW = PW1 + PW2;   // pieces per wave per stage (uniform: counted vmcnt)
  static constexpr int STAGE_B = (NP + (((W1P | W2P) & 3) ? 1 : 0)) * 1024;          // + one dump piece for the padding loads
  static constexpr int LDS_B = NST * STAGE_B + H4 * 4;
  static_assert(C % 16 == 0 && H4 % (32 * G) == 0, "shape");
  static_assert((NST - 1) * PPW <= 63, "vmcnt range");
};

template <int C, int TN, int G, int NST, int FB>
__global__ __launch_bounds__(256, 1) void mlp_fused_kernel(const MlpFusedParams p) {
  using K = MlpCfg<C, TN, G, NST>;
  constexpr int KS = K::KS, OT = K::OT, H4 = K::H4, NSTG = K::NSTG, W1P = K::W1P, W2P = K::W2P, NP = K::NP, PW1 = K::PW1, PW2 = K::PW2, PPW = K::PPW, STAGE_B = K::STAGE_B;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bias_lds = reinterpret_cast<float*>(smem + NST * STAGE_B);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int tok0 = (blockIdx.x * 4 + wave) * (32 * TN);

  // X fragments (B operand): X[token fr][16 s + 8 fh + j]; rows past M are clamped (their results are not stored)
  half8 xf[TN][KS];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const half_t* xp = p.x16 + (size_t)min(tok0 + 32 * tn + fr, p.M - 1) * p.ldx + fh * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) xf[tn][s] = *reinterpret_cast<const half8*>(xp + s * 16);
  }
  for (int i = tid; i < H4; i += 256) bias_lds[i] = p.b1[i];
  __syncthreads();                           // bias table visible (also retires the X loads)

  // ---- LDS-DMA: piece q of a stage = fragment tile (32 rows x 32 B); lane l carries row l >> 1, 16-B half (l & 1) ^ swz(row)
  const int prow = lane >> 1, phalf = (lane & 1) ^ ((prow >> 3) & 1);
  const unsigned off_w1 = (unsigned)(prow * C + 8 * phalf) * 2u;                               // bytes
  const unsigned off_w2 = (unsigned)(prow * H4 + 8 * phalf) * 2u;
  const unsigned off_w2_last = (unsigned)((min(32 * (OT - 1) + prow, C - 1) - 32 * (OT - 1)) * H4 + 8 * phalf) * 2u;
  const char* w1b = reinterpret_cast<const char*>(p.w1);
  const char* w2b = reinterpret_cast<const char*>(p.w2);
  // Step i of a stage is the same KIND of piece on all four waves (W1 pieces first, each kind padded to a multiple of 4 with
  // dumped re-loads of its first piece), so the operand base is a compile-time choice and only scalar coordinates vary.
  auto issue = [&](int st) {                 // stage st -> ring slot st % NST
    char* sb = smem + (st % NST) * STAGE_B;
    const int hid0 = st * (32 * G);
#pragma unroll
    for (int i = 0; i < PW1; ++i) {
      const int q = wave + 4 * i;            // wave-uniform
      const bool pad = (W1P & 3) != 0 && q >= W1P;
      const int qe = pad ? 0 : q;
      const int g = qe / KS, ks = qe - g * KS;
      const char* src = sgpr_ptr(w1b + ((size_t)(hid0 + 32 * g) * C + 16 * ks) * 2);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + off_w1), (lds_ptr_t)(sb + (pad ? NP : q) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < PW2; ++i) {
      const int q = wave + 4 * i;
      const bool pad = (W2P & 3) != 0 && q >= W2P;
      const int qe = pad ? 0 : q;
      const int g = qe / (2 * OT), r = qe - g * (2 * OT), t = r >> 1, ks = r & 1;
      const char* src = sgpr_ptr(w2b + ((size_t)(32 * t) * H4 + hid0 + 32 * g + 16 * ks) * 2);
      const unsigned off = (C % 32 != 0 && t == OT - 1) ? off_w2_last : off_w2;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + off), (lds_ptr_t)(sb + (pad ? NP : W1P + q) * 1024), 16, 0, 0);
    }
  };
  // fragment read addresses (bytes inside a piece)
  const int r1 = pi23(fr);
  const int rd_w1 = r1 * 32 + ((fh ^ ((r1 >> 3) & 1)) << 4);
  const int rd_w2 = fr * 32 + ((fh ^ ((fr >> 3) & 1)) << 4);

  f32x16 acc[TN][OT];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tn][t][r] = 0.f;

#pragma unroll
  for (int st = 0; st < NST - 1; ++st)
    if (st < NSTG) issue(st);

#pragma nounroll
  for (int st = 0; st < NSTG; ++st) {
    // stage st landed (the NST-2 younger stages may stay in flight), everyone is done with the slot refilled next
    {
      const int later = min(NST - 2, NSTG - 1 - st);
      if (NST >= 4 && later == 2) wait_vm<2 * PPW>();
      else if (NST >= 3 && later == 1) wait_vm<PPW>();
      else wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      if (st + NST - 1 < NSTG) issue(st + NST - 1);
    }
    const char* sb = smem + (st % NST) * STAGE_B;
    const char* sW1 = sb + rd_w1;
    const char* sW2 = sb + W1P * 1024 + rd_w2;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      // ---- S^T = W1_chunk X^T + b1   (bias = initial accumulator; registers 8 ks + j <-> hidden 16 ks + 8 fh + j)
      f32x16 sacc[TN];
      {
        const float* bl = bias_lds + st * (32 * G) + 32 * g + 8 * fh;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bl), b1 = *reinterpret_cast<const f32x4*>(bl + 4);
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(bl + 16), b3 = *reinterpret_cast<const f32x4*>(bl + 20);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int j = 0; j < 4; ++j) { sacc[tn][j] = b0[j]; sacc[tn][4 + j] = b1[j]; sacc[tn][8 + j] = b2[j]; sacc[tn][12 + j] = b3[j]; }
      }
      {
        constexpr int NB = (KS + FB - 1) / FB;
        half8 cur[FB], nxt[FB];
#pragma unroll
        for (int j = 0; j < FB; ++j)
          if (j < KS) cur[j] = *reinterpret_cast<const half8*>(sW1 + (g * KS + j) * 1024);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
          for (int j = 0; j < FB; ++j)
            if ((b + 1) * FB + j < KS) nxt[j] = *reinterpret_cast<const half8*>(sW1 + (g * KS + (b + 1) * FB + j) * 1024);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < FB; ++j)
            if (b * FB + j < KS) {
#pragma unroll
              for (int tn = 0; tn < TN; ++tn) sacc[tn] = mfma32(cur[j], xf[tn][b * FB + j], sacc[tn]);
            }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < FB; ++j) cur[j] = nxt[j];
        }
      }
      // ---- P = GELU(S^T) as f16 B operands
      half8 pf[TN][2];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) pf[tn][r >> 3][r & 7] = (half_t)gelu_erf_fast(sacc[tn][r]);
      // ---- Y^T += W2_chunk P
      {
        constexpr int NF = 2 * OT, NB = (NF + FB - 1) / FB;
        const char* sW2g = sW2 + g * (2 * OT) * 1024;
        half8 cur[FB], nxt[FB];
#pragma unroll
        for (int j = 0; j < FB; ++j)
          if (j < NF) cur[j] = *reinterpret_cast<const half8*>(sW2g + j * 1024);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
          for (int j = 0; j < FB; ++j)
            if ((b + 1) * FB + j < NF) nxt[j] = *reinterpret_cast<const half8*>(sW2g + ((b + 1) * FB + j) * 1024);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < FB; ++j)
            if (b * FB + j < NF) {
              const int f = b * FB + j;
#pragma unroll
              for (int tn = 0; tn < TN; ++tn) acc[tn][f >> 1] = mfma32(cur[j], pf[tn][f & 1], acc[tn][f >> 1]);
            }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < FB; ++j) cur[j] = nxt[j];
        }
      }
    }
  }

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

template <int C, int TN, int G, int NST, int FB>
hipError_t launch_cfg(const MlpFusedParams& p, hipStream_t s) {
  using K = MlpCfg<C, TN, G, NST>;
  const int wg_tok = 128 * TN;
  const int grid = (p.M + wg_tok - 1) / wg_tok;
  mlp_fused_kernel<C, TN, G, NST, FB><<<dim3(grid), dim3(256), K::LDS_B, s>>>(p);
  return hipGetLastError();
}
template <int C, int TN, int G, int NST, int FB>
hipError_t attr_cfg() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_kernel<C, TN, G, NST, FB>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, MlpCfg<C, TN, G, NST>::LDS_B);
}
}  // namespace

// Configurations: C = 144: 64 tokens per wave, 2 chunks per stage (38 + 2 padding pieces), 3-stage ring (117 KiB);
// C = 288: 32 tokens per wave, 1 chunk per stage (36 + 4 padding pieces), 4-stage ring (148 KiB).
// C = 576 does not fit: 32 tokens need 144 (X) + 288 (Y^T) of the 512 registers and the compiler spills X into scratch
// (reloaded inside the MFMA chains behind vmcnt(0), which also drains the DMA ring) - stage 3 stays on the two-GEMM path.
hipError_t mlp_fused_init() {
  hipError_t e[2] = {attr_cfg<144, 2, 2, 3, 5>(), attr_cfg<288, 1, 1, 4, 6>()};
  for (int i = 0; i < 2; ++i)
    if (e[i] != hipSuccess) return e[i];
  return hipSuccess;
}

bool mlp_fused_supported(int C) { return C == 144 || C == 288; }

hipError_t mlp_fused_launch(const MlpFusedParams& p, int C, hipStream_t s) {
  if (p.M <= 0) return hipSuccess;
  if ((p.ldx & 7) || (p.ld32 & 3)) return hipErrorInvalidValue;
  switch (C) {
    case 144: return launch_cfg<144, 2, 2, 3, 5>(p, s);
    case 288: return launch_cfg<288, 1, 1, 4, 6>(p, s);
    default: return hipErrorInvalidValue;
  }
}
