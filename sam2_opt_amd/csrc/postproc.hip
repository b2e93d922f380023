// Mask post-processing: small-hole filling.
//
// Behaviour of fill_holes_in_mask_scores (/root/reference/sam2/sam2/utils/misc.py:312-338) with
// get_connected_components (:47-62, 8-connectivity; the reference computes it with its CUDA extension
// csrc/connected_components.cu and silently skips the step when that is missing):
//     holes = background (score <= 0) connected components with area <= max_area;  score[hole] = 0.1
//
// Only "is my component small" is needed, never the labels, so there is no union-find and no global pass here:
// every background pixel runs a BOUNDED breadth-first fill of its own component and stops as soon as it has seen
// max_area + 1 pixels.  A pixel of a large region gives up after a handful of steps, a pixel of a small hole
// enumerates the whole hole - exact, embarrassingly parallel, one launch, no atomics.  The visited list (at most
// max_area + 1 pixel indices per thread) lives in LDS.
#include "kernels.h"

namespace {
constexpr int FH_THREADS = 256;

__global__ __launch_bounds__(FH_THREADS) void fill_holes_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                              int max_area, float fill_value) {
  extern __shared__ int lists[];                       // [FH_THREADS][max_area + 1]
  const int cap = max_area + 1;
  const size_t plane = (size_t)H * W;
  const float* img = in + (size_t)blockIdx.y * plane;
  const int p = blockIdx.x * FH_THREADS + threadIdx.x;
  if (p >= H * W) return;
  const float v = img[p];
  float r = v;
  if (v <= 0.f) {
    int* L = lists + threadIdx.x * cap;
    int head = 0, tail = 1;
    L[0] = p;
    bool small = true;
    while (head < tail && small) {
      const int c = L[head++];
      const int cy = c / W, cx = c - cy * W;
#pragma unroll 1
      for (int k = 0; k < 8 && small; ++k) {
        const int dy = (k < 3) ? -1 : (k < 5 ? 0 : 1);
        const int dx = (k == 0 || k == 3 || k == 5) ? -1 : ((k == 1 || k == 6) ? 0 : 1);
        const int ny = cy + dy, nx = cx + dx;
        if (ny < 0 || ny >= H || nx < 0 || nx >= W) continue;
        const int q = ny * W + nx;
        if (img[q] > 0.f) continue;
        bool seen = false;
        for (int j = 0; j < tail; ++j) seen |= (L[j] == q);
        if (seen) continue;
        L[tail++] = q;                                 // tail <= max_area here, the list holds max_area + 1
        if (tail > max_area) small = false;            // the component has more than max_area pixels
      }
    }
    if (small) r = fill_value;
  }
  out[(size_t)blockIdx.y * plane + p] = r;
}
}  // namespace

hipError_t fill_holes_launch(const float* in, float* out, int N, int H, int W, int max_area, hipStream_t s) {
  if (N <= 0 || H <= 0 || W <= 0 || max_area < 1 || max_area > FILL_HOLES_MAX_AREA || in == out) return hipErrorInvalidValue;
  const size_t lds = (size_t)FH_THREADS * (max_area + 1) * sizeof(int);
  fill_holes_kernel<<<dim3((H * W + FH_THREADS - 1) / FH_THREADS, N), dim3(FH_THREADS), lds, s>>>(in, out, H, W, max_area, 0.1f);
  return hipGetLastError();
}
