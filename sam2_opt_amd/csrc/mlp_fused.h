// Fused Hiera MLP (mlp_fused.hip):  x32 += fc2(GELU(fc1(x16))) with x16 = LN2(x32) as f16, one kernel, hidden stays on chip.
#pragma once
#include "common.h"

struct MlpFusedParams {
  const half_t* x16; int ldx;   // [M, C] f16 (LayerNorm output), ldx % 8 == 0
  const half_t* wpack;          // fc1 + fc2 weights in the kernel's piece order (mlp_fused_pack)
  const float* b1;              // [4C]
  const float* b2;              // [C]
  float* x32; int ld32;         // residual stream [M, C] f32, updated in place; ld32 % 4 == 0
  int M;
  float ln_eps;                 // > 0: LayerNorm fused into the operand load - x16 is ignored, X = normalised rows of x32 (the
                                // LN2 input IS the residual stream), and wpack / b1 must carry the affine part (see gemm_xs.h)
};
bool mlp_fused_supported(int C);                     // C in {144, 288}
// Re-orders fc1 [4C, C] and fc2 [C, 4C] (f16, nn.Linear layout) into the LDS-DMA piece stream of the kernel; once per block.
size_t mlp_fused_pack_bytes(int C);
hipError_t mlp_fused_pack(const half_t* w1, const half_t* w2, int C, half_t* wpack, hipStream_t s);
hipError_t mlp_fused_launch(const MlpFusedParams& p, int C, hipStream_t s);
const char* mlp_fused_kernel_name(int C);            // kernel instantiation launched for C, as rocprofv3 prints it
hipError_t mlp_fused_init();                         // dynamic-LDS attributes, once
