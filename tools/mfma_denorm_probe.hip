// Hardware probe (gfx950): does v_mfma_f32_32x32x16_f16 honour f16 SUBNORMAL inputs, and does the f32 -> f16 conversion
// produce them?  Decides whether the lo part of a split operand may be stored unscaled (lo = f16(v - f16(v)), mostly
// subnormal for |v| < 0.25) and accumulated into the SAME accumulator as the hi product, or needs the 2^11 scale and
// an accumulator of its own.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_denorm_probe.hip -o gpurun_out/mfma_denorm_probe && gpurun_out/mfma_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const float* av, const float* bv, float* out, _Float16* cvt) {
  const int lane = threadIdx.x;
  half8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)av[0];       // every A element = av[0]
    b[e] = (_Float16)bv[0];
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (lane == 0) {
    out[0] = c[0];                // = 16 * a * b when nothing is flushed
    cvt[0] = a[0];
    cvt[1] = b[0];
  }
}

int main() {
  float *av, *bv, *out;
  _Float16* cvt;
  hipMalloc(&av, 4); hipMalloc(&bv, 4); hipMalloc(&out, 4); hipMalloc(&cvt, 4);
  const float as[] = {1.0f, 3.0e-5f, 1.0e-6f, 6.0e-8f, 2.4e-4f};
  const float bs[] = {1.0f, 1.0f, 1.0f, 1.0f, 3.0e-5f};
  int bad = 0;
  for (int i = 0; i < 5; ++i) {
    hipMemcpy(av, &as[i], 4, hipMemcpyHostToDevice);
    hipMemcpy(bv, &bs[i], 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(av, bv, out, cvt);
    float o;
    _Float16 c2[2];
    hipMemcpy(&o, out, 4, hipMemcpyDeviceToHost);
    hipMemcpy(c2, cvt, 4, hipMemcpyDeviceToHost);
    const double want = 16.0 * (double)(float)c2[0] * (double)(float)c2[1];
    printf("a=%.3e (f16 %.6e) b=%.3e (f16 %.6e): mfma %.6e  expected %.6e  %s\n", as[i], (double)(float)c2[0], bs[i], (double)(float)c2[1], o, want,
           (fabs(o - want) <= 1e-6 * fabs(want) && want != 0.0) ? "HONOURED" : "FLUSHED/DIFFERENT");
    if (!(fabs(o - want) <= 1e-6 * fabs(want)) || want == 0.0) ++bad;
  }
  printf("subnormal f16 MFMA inputs: %s\n", bad ? "NOT fully honoured" : "honoured");
  return 0;
}
