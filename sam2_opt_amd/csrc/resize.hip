// Frame ingest on the device: the two resizers the reference applies to frames that are not 1024 x 1024.
//
//  * video frames  - load_video_frames_from_jpg_images / _load_img_as_tensor (/root/reference/sam2/sam2/utils/misc.py:92-101):
//    PIL `Image.resize((S, S))` on the decoded uint8 RGB frame.  Pillow's resize (src/libImaging/Resample.c, pinned by the
//    container's Pillow 12.2.0) is a separable convolution: bicubic kernel with a = -0.5 and support 2, stretched by the scale
//    factor when shrinking (antialiasing), coefficients normalised per output pixel in double precision and converted to
//    22-bit fixed point, horizontal pass first, a uint8 rounding (+2^21, >> 22, clamp to 0..255) after EACH pass.  All of it is
//    integer arithmetic, so this kernel is bit-exact against PIL (tests/test_ingest.py).
//  * images        - SAM2Transforms (utils/transforms.py:27-41): ToTensor (/255) and torchvision's Resize on a FLOAT tensor =
//    torch.nn.functional.interpolate(mode="bilinear", antialias=True, align_corners=False) = aten _upsample_bilinear2d_aa:
//    the same separable scheme with the triangle kernel, float weights, float accumulation, width first.
//
// The coefficient tables depend only on (input size, output size): they are built on the host once per size pair and cached in
// the context.  Kernels are memory-bound gathers: one thread per output element, coalesced over x and the 3 channels.
#include "engine.h"

namespace {

inline double bicubic_pil(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// Pillow precompute_coeffs + normalize_coeffs_8bpc (Resample.c): bounds[2 o] = first input index, bounds[2 o + 1] = tap count
void pil_coeffs(int in_size, int out_size, std::vector<int>& bounds, std::vector<int>& kk, int& ksize) {
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale;
  ksize = (int)std::ceil(support) * 2 + 1;
  bounds.assign((size_t)out_size * 2, 0);
  kk.assign((size_t)out_size * ksize, 0);
  std::vector<double> k(ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale, ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      k[x] = bicubic_pil((x + xmin - center + 0.5) * ss);
      ww += k[x];
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) k[x] /= ww;
      const double v = k[x] * (double)(1 << 22);
      kk[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v) : (int)(0.5 + v);
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
}

// aten _compute_indices_min_size_weights_aa with the bilinear (triangle) filter, float arithmetic, align_corners = False
void aa_coeffs(int in_size, int out_size, std::vector<int>& bounds, std::vector<float>& wt, int& ksize) {
  const float scale = (float)in_size / (float)out_size;
  const float support = (scale >= 1.0f) ? 1.0f * scale : 1.0f;            // interp_size * 0.5 = 1
  ksize = (int)std::ceil(support) * 2 + 1;
  bounds.assign((size_t)out_size * 2, 0);
  wt.assign((size_t)out_size * ksize, 0.f);
  for (int i = 0; i < out_size; ++i) {
    const float center = scale * (i + 0.5f);
    const float invscale = (scale >= 1.0f) ? 1.0f / scale : 1.0f;
    const int xmin = std::max((int)(center - support + 0.5f), 0);
    const int xsize = std::min((int)(center + support + 0.5f), in_size) - xmin;
    float total = 0.f;
    float* w = &wt[(size_t)i * ksize];
    for (int j = 0; j < xsize; ++j) {
      float x = (j + xmin - center + 0.5f) * invscale;
      if (x < 0.f) x = -x;
      w[j] = x < 1.0f ? 1.0f - x : 0.0f;
      total += w[j];
    }
    for (int j = 0; j < xsize; ++j)
      if (total != 0.f) w[j] /= total;
    bounds[2 * i] = xmin;
    bounds[2 * i + 1] = xsize;
  }
}

__device__ __forceinline__ uint8_t clip8(int ss) {
  const int v = ss >> 22;                                  // arithmetic shift = floor, as clip8_lookups[in >> PRECISION_BITS]
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: in [H, W, 3] -> tmp [H, S, 3]; vertical pass: tmp [H, S, 3] -> out [S, S, 3]
__global__ void pil_h_kernel(const uint8_t* __restrict__ in, int H, int W, uint8_t* __restrict__ tmp, int S, const int* __restrict__ bounds,
                             const int* __restrict__ kk, int ksize) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)H * S * 3) return;
  const int c = (int)(i % 3), xo = (int)((i / 3) % S), y = (int)(i / ((size_t)3 * S));
  const int xmin = bounds[2 * xo], n = bounds[2 * xo + 1];
  const int* k = kk + (size_t)xo * ksize;
  const uint8_t* row = in + ((size_t)y * W + xmin) * 3 + c;
  int ss = 1 << 21;
  for (int x = 0; x < n; ++x) ss += (int)row[(size_t)x * 3] * k[x];
  tmp[i] = clip8(ss);
}
__global__ void pil_v_kernel(const uint8_t* __restrict__ tmp, int H, uint8_t* __restrict__ out, int S, const int* __restrict__ bounds,
                             const int* __restrict__ kk, int ksize) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)S * S * 3) return;
  const size_t xc = i % ((size_t)3 * S);                   // (x, c) offset inside a row
  const int yo = (int)(i / ((size_t)3 * S));
  const int ymin = bounds[2 * yo], n = bounds[2 * yo + 1];
  const int* k = kk + (size_t)yo * ksize;
  int ss = 1 << 21;
  for (int y = 0; y < n; ++y) ss += (int)tmp[(size_t)(ymin + y) * 3 * S + xc] * k[y];
  out[i] = clip8(ss);
}

// width first: in u8 [H, W, 3] (/255) -> tmp f32 [H, S, 3]; then height: -> out f32 [3, S, S]
__global__ void aa_h_kernel(const uint8_t* __restrict__ in, int H, int W, float* __restrict__ tmp, int S, const int* __restrict__ bounds,
                            const float* __restrict__ wt, int ksize) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)H * S * 3) return;
  const int c = (int)(i % 3), xo = (int)((i / 3) % S), y = (int)(i / ((size_t)3 * S));
  const int xmin = bounds[2 * xo], n = bounds[2 * xo + 1];
  const float* w = wt + (size_t)xo * ksize;
  const uint8_t* row = in + ((size_t)y * W + xmin) * 3 + c;
  float t = ((float)row[0] / 255.0f) * w[0];               // ToTensor: uint8 -> float, div(255); then the aten accumulation order
  for (int x = 1; x < n; ++x) t += ((float)row[(size_t)x * 3] / 255.0f) * w[x];
  tmp[i] = t;
}
__global__ void aa_v_kernel(const float* __restrict__ tmp, int H, float* __restrict__ out, int S, const int* __restrict__ bounds,
                            const float* __restrict__ wt, int ksize) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)S * S * 3) return;
  const int xo = (int)(i % S), yo = (int)((i / S) % S), c = (int)(i / ((size_t)S * S));
  const int ymin = bounds[2 * yo], n = bounds[2 * yo + 1];
  const float* w = wt + (size_t)yo * ksize;
  const float* col = tmp + ((size_t)ymin * S + xo) * 3 + c;
  float t = col[0] * w[0];
  for (int y = 1; y < n; ++y) t += col[(size_t)y * S * 3] * w[y];
  out[i] = t;
}
}  // namespace

// cached device tables for one (kind, in, out) triple.  First use of a size triple allocates (hipMalloc: may synchronise the
// device once) and uploads the tables with hipMemcpyAsync on the CALLER's stream from host vectors the table keeps alive, so the
// launching thread is not parked behind work queued on other streams; later calls only launch kernels (sam2mi.h).
static int resize_table(sam2mi_ctx* ctx, hipStream_t s, int kind, int in_size, int out_size, const ResizeTable** out) {
  const uint64_t key = ((uint64_t)kind << 60) | ((uint64_t)in_size << 30) | (uint64_t)out_size;
  auto it = ctx->resize_tables.find(key);
  if (it == ctx->resize_tables.end()) {
    it = ctx->resize_tables.emplace(key, ResizeTable()).first;
    ResizeTable& t = it->second;
    if (kind == 0) {
      std::vector<int> kk;
      pil_coeffs(in_size, out_size, t.h_bounds, kk, t.ksize);
      t.h_coef.resize(kk.size() * sizeof(int));
      memcpy(t.h_coef.data(), kk.data(), t.h_coef.size());
    } else {
      std::vector<float> wt;
      aa_coeffs(in_size, out_size, t.h_bounds, wt, t.ksize);
      t.h_coef.resize(wt.size() * sizeof(float));
      memcpy(t.h_coef.data(), wt.data(), t.h_coef.size());
    }
    t.coef = dalloc_raw(ctx, t.h_coef.size());
    t.bounds = (int*)dalloc_raw(ctx, t.h_bounds.size() * sizeof(int));
    if (!t.coef || !t.bounds) {
      ctx->resize_tables.erase(it);
      return sam2mi_set_error(ctx, "resize", "hipMalloc failed");
    }
    CHK(hipMemcpyAsync(t.coef, t.h_coef.data(), t.h_coef.size(), hipMemcpyHostToDevice, s));
    CHK(hipMemcpyAsync(t.bounds, t.h_bounds.data(), t.h_bounds.size() * sizeof(int), hipMemcpyHostToDevice, s));
  }
  *out = &it->second;
  return 0;
}

// one scratch buffer per context, grown when a larger frame arrives; the old one is released (hipFree waits for the device, so
// work still reading it has finished) instead of staying allocated until sam2mi_destroy
static int resize_scratch(sam2mi_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->resize_tmp_bytes) {
    if (ctx->resize_tmp) dfree(ctx, ctx->resize_tmp);
    ctx->resize_tmp_bytes = 0;
    ctx->resize_tmp = dalloc_raw(ctx, bytes);
    if (!ctx->resize_tmp) return sam2mi_set_error(ctx, "resize", "hipMalloc failed");
    ctx->resize_tmp_bytes = bytes;
  }
  *out = ctx->resize_tmp;
  return 0;
}

static int check_sizes(sam2mi_ctx* ctx, const void* in, const void* out, int H, int W, int S) {
  if (!ctx) return 1;
  if (!in || !out) return sam2mi_set_error(ctx, "resize", "null buffer");
  if (H < 1 || W < 1 || S < 1 || H > 16384 || W > 16384 || S > 16384) return sam2mi_set_error(ctx, "resize", "size out of range (1..16384)");
  return 0;
}

extern "C" int sam2mi_resize_u8_pil_bicubic(sam2mi_ctx* ctx, void* stream, const uint8_t* in, int H, int W, uint8_t* out, int S) {
  CHKI(check_sizes(ctx, in, out, H, W, S));
  hipStream_t s = (hipStream_t)stream;
  DomainGuard guard_(ctx->dom_enc, s);
  if (H == S && W == S) {                                   // Pillow returns a copy when nothing changes
    CHK(hipMemcpyAsync(out, in, (size_t)S * S * 3, hipMemcpyDeviceToDevice, s));
    return 0;
  }
  const ResizeTable *tx, *ty;
  CHKI(resize_table(ctx, s, 0, W, S, &tx));
  CHKI(resize_table(ctx, s, 0, H, S, &ty));
  void* tmp;
  CHKI(resize_scratch(ctx, (size_t)H * S * 3, &tmp));
  // Pillow skips a pass whose size does not change; with equal sizes its coefficients are the identity, so running it is the same
  const size_t n1 = (size_t)H * S * 3, n2 = (size_t)S * S * 3;
  pil_h_kernel<<<dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s>>>(in, H, W, (uint8_t*)tmp, S, tx->bounds, (const int*)tx->coef, tx->ksize);
  CHK(hipGetLastError());
  pil_v_kernel<<<dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s>>>((const uint8_t*)tmp, H, out, S, ty->bounds, (const int*)ty->coef, ty->ksize);
  CHK(hipGetLastError());
  return 0;
}

extern "C" int sam2mi_resize_image_aa_bilinear(sam2mi_ctx* ctx, void* stream, const uint8_t* in, int H, int W, float* out, int S) {
  CHKI(check_sizes(ctx, in, out, H, W, S));
  hipStream_t s = (hipStream_t)stream;
  DomainGuard guard_(ctx->dom_enc, s);
  const ResizeTable *tx, *ty;
  CHKI(resize_table(ctx, s, 1, W, S, &tx));
  CHKI(resize_table(ctx, s, 1, H, S, &ty));
  void* tmp;
  CHKI(resize_scratch(ctx, (size_t)H * S * 3 * sizeof(float), &tmp));
  const size_t n1 = (size_t)H * S * 3, n2 = (size_t)S * S * 3;
  aa_h_kernel<<<dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s>>>(in, H, W, (float*)tmp, S, tx->bounds, (const float*)tx->coef, tx->ksize);
  CHK(hipGetLastError());
  aa_v_kernel<<<dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s>>>((const float*)tmp, H, out, S, ty->bounds, (const float*)ty->coef, ty->ksize);
  CHK(hipGetLastError());
  return 0;
}
