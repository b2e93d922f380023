// Fused Hiera MLP (mlp_fused.hip):  x32 += fc2(GELU(fc1(x16))) with x16 = LN2(x32) as f16, one kernel, hidden stays on chip.
#pragma once
#include "common.h"

struct MlpFusedParams {
  const half_t* x16; int ldx;   // [M, C] f16 (LayerNorm output), ldx % 8 == 0
  const half_t* w1;             // fc1 weight [4C, C] f16 (nn.Linear layout)
  const float* b1;              // [4C]
  const half_t* w2;             // fc2 weight [C, 4C] f16
  const float* b2;              // [C]
  float* x32; int ld32;         // residual stream [M, C] f32, updated in place; ld32 % 4 == 0
  int M;
};
bool mlp_fused_supported(int C);                     // C in {144, 288}
hipError_t mlp_fused_launch(const MlpFusedParams& p, int C, hipStream_t s);
hipError_t mlp_fused_init();                         // dynamic-LDS attributes, once
