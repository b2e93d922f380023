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

// ===================================================================== mask prompts (add_new_mask, correction clicks)
namespace {
// F.interpolate(x * a + b, scale 1/4, mode="bilinear", antialias=True, align_corners=False) for a square S x S input
// (sam2_base_official.py:505-511): separable triangle filter of support 4 input pixels each side, weights renormalised at
// the borders (ATen _compute_indices_weights_aa: center = 4 (i + 0.5), taps [int(c - 4 + 0.5), int(c + 4 + 0.5)) clipped).
__global__ void aa_down4_kernel(const float* __restrict__ in, int S, float a, float b, float* __restrict__ out) {
  const int So = S / 4;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= So * So) return;
  const int oy = i / So, ox = i % So;
  float wy[8], wx[8];
  int y0, x0, ny, nx;
  auto taps = [&](int o, float* w, int& lo, int& n) {
    const float c = 4.f * (o + 0.5f);
    lo = max(0, (int)(c - 4.f + 0.5f));
    const int hi = min(S, (int)(c + 4.f + 0.5f));
    n = hi - lo;
    float tot = 0.f;
    for (int j = 0; j < 8; ++j) {
      const float x = (j + lo - c + 0.5f) * 0.25f;
      w[j] = j < n ? fmaxf(0.f, 1.f - fabsf(x)) : 0.f;
      tot += w[j];
    }
    for (int j = 0; j < 8; ++j) w[j] /= tot;
  };
  taps(oy, wy, y0, ny);
  taps(ox, wx, x0, nx);
  float acc = 0.f;
  for (int j = 0; j < ny; ++j) {
    float row = 0.f;
    for (int k = 0; k < nx; ++k) row += wx[k] * (in[(size_t)(y0 + j) * S + x0 + k] * a + b);
    acc += wy[j] * row;
  }
  out[i] = acc;
}

// SAM2Base.mask_downsample: Conv2d(1, 1, kernel 4, stride 4) (sam2_base_official.py:200-203) on a [S, S] mask -> [S/4, S/4];
// also raises *any_pos when some input pixel is > 0 (is_obj_appearing of _use_mask_as_output, :527-529)
__global__ void conv4x4s4_kernel(const float* __restrict__ in, int S, const float* __restrict__ w, const float* __restrict__ bias,
                                 float* __restrict__ out, int* any_pos) {
  const int So = S / 4;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= So * So) return;
  const int oy = i / So, ox = i % So;
  float acc = bias[0];
  bool pos = false;
  for (int ky = 0; ky < 4; ++ky)
    for (int kx = 0; kx < 4; ++kx) {
      const float v = in[(size_t)(4 * oy + ky) * S + 4 * ox + kx];
      acc += w[ky * 4 + kx] * v;
      pos |= v > 0.f;
    }
  out[i] = acc;
  if (__any(pos) && (threadIdx.x & 63) == 0) atomicOr(any_pos, 1);
}

// PromptEncoder._embed_masks (sam/prompt_encoder.py:168-171): mask_downscaling = Conv2d(1,4,2,s2) LN2d GELU Conv2d(4,16,2,s2)
// LN2d GELU Conv2d(16,256,1) on a [256,256] mask prompt -> dense embedding, token-major [4096, 256].
// One 256-thread workgroup per output token: the 4x4 input patch and the two small stages in LDS, one thread per channel.
__global__ __launch_bounds__(256) void mask_embed_kernel(const float* __restrict__ mask, MaskEmbedW W, float* __restrict__ out) {
  __shared__ float s1[4][4];      // stage 1: [2x2 positions][4 channels] after LN + GELU
  __shared__ float s2[16];        // stage 2: 16 channels after LN + GELU
  const int tok = blockIdx.x, ty = tok >> 6, tx = tok & 63;
  const int t = threadIdx.x;
  if (t < 4) {                    // position t = py*2+px of the 2x2 stage-1 grid
    const int py = t >> 1, px = t & 1;
    float v[4], mean = 0.f;
    for (int c = 0; c < 4; ++c) {
      float a = W.b1[c];
      for (int ky = 0; ky < 2; ++ky)
        for (int kx = 0; kx < 2; ++kx) a += W.w1[c * 4 + ky * 2 + kx] * mask[(size_t)(4 * ty + 2 * py + ky) * 256 + 4 * tx + 2 * px + kx];
      v[c] = a;
      mean += a;
    }
    mean *= 0.25f;
    float var = 0.f;
    for (int c = 0; c < 4; ++c) var += (v[c] - mean) * (v[c] - mean);
    const float rstd = 1.f / sqrtf(var * 0.25f + 1e-6f);
    for (int c = 0; c < 4; ++c) s1[t][c] = gelu_erf((v[c] - mean) * rstd * W.ln1w[c] + W.ln1b[c]);
  }
  __syncthreads();
  if (t < 16) {                   // conv 2x2 s2 over the 2x2 grid: weight [16][4][2][2]
    float a = W.b2[t];
    for (int ci = 0; ci < 4; ++ci)
      for (int p = 0; p < 4; ++p) a += W.w2[(t * 4 + ci) * 4 + p] * s1[p][ci];
    s2[t] = a;
  }
  __syncthreads();
  if (t == 0) {
    float mean = 0.f;
    for (int c = 0; c < 16; ++c) mean += s2[c];
    mean *= (1.f / 16.f);
    float var = 0.f;
    for (int c = 0; c < 16; ++c) var += (s2[c] - mean) * (s2[c] - mean);
    const float rstd = 1.f / sqrtf(var * (1.f / 16.f) + 1e-6f);
    for (int c = 0; c < 16; ++c) s2[c] = gelu_erf((s2[c] - mean) * rstd * W.ln2w[c] + W.ln2b[c]);
  }
  __syncthreads();
  float a = W.b3[t];
  for (int c = 0; c < 16; ++c) a += W.w3[t * 16 + c] * s2[c];
  out[(size_t)tok * 256 + t] = a;
}
}  // namespace

namespace {
__global__ void flag_to_score_kernel(const int* flag, float on, float off, float* out) { out[0] = flag[0] ? on : off; }
}  // namespace
hipError_t flag_to_score_launch(const int* flag, float on, float off, float* out, hipStream_t s) {
  flag_to_score_kernel<<<dim3(1), dim3(1), 0, s>>>(flag, on, off, out);
  return hipGetLastError();
}
hipError_t aa_down4_launch(const float* in, int S, float a, float b, float* out, hipStream_t s) {
  if (S % 4 || S < 8) return hipErrorInvalidValue;
  const int n = (S / 4) * (S / 4);
  aa_down4_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(in, S, a, b, out);
  return hipGetLastError();
}
hipError_t conv4x4s4_launch(const float* in, int S, const float* w, const float* bias, float* out, int* any_pos, hipStream_t s) {
  if (S % 4) return hipErrorInvalidValue;
  hipError_t e = hipMemsetAsync(any_pos, 0, sizeof(int), s);
  if (e != hipSuccess) return e;
  const int n = (S / 4) * (S / 4);
  conv4x4s4_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(in, S, w, bias, out, any_pos);
  return hipGetLastError();
}
hipError_t mask_embed_launch(const float* mask256, const MaskEmbedW& W, float* dense_tok, hipStream_t s) {
  mask_embed_kernel<<<dim3(4096), dim3(256), 0, s>>>(mask256, W, dense_tok);
  return hipGetLastError();
}
