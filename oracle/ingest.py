"""TEST INFRASTRUCTURE (oracle): CPU restatements of the reference's two frame resizers.

 * pil_bicubic_u8    - what load_video_frames_from_jpg_images does to a decoded frame (/root/reference/sam2/sam2/utils/misc.py:
   92-101: `np.array(img_pil.convert("RGB").resize((S, S)))`): Pillow's Image.resize default = BICUBIC through
   src/libImaging/Resample.c (third-party, not under /root/reference; pinned version: the container's Pillow 12.2.0).  Published
   algorithm restated here in integer numpy: precompute_coeffs (double, a = -0.5, support 2 stretched by the scale when
   shrinking), normalize_coeffs_8bpc (22-bit fixed point), horizontal pass then vertical pass, each ending in
   clip8((acc + 2^21) >> 22).  Pinned bit-exactly against Pillow itself in tests/test_ingest.py.
 * aa_bilinear_f32   - SAM2Transforms (utils/transforms.py:27-41): ToTensor + torchvision Resize on a float tensor =
   torch.nn.functional.interpolate(mode="bilinear", antialias=True) (aten _upsample_bilinear2d_aa): float weights with the
   triangle filter, width first.  Pinned against torch itself.
Only tests/ may import this module.
"""
import math

import numpy as np


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_coeffs(in_size, out_size):
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [_bicubic((x + xmin - center + 0.5) / filterscale) for x in range(xmax)]
        ww = sum(k)
        kk = []
        for v in k:
            v = (v / ww if ww != 0.0 else v) * (1 << 22)
            kk.append(int(-0.5 + v) if v < 0 else int(0.5 + v))
        out.append((xmin, np.array(kk, dtype=np.int64)))
    return out


def _clip8(ss):
    return np.clip(ss >> 22, 0, 255).astype(np.uint8)


def pil_bicubic_u8(img: np.ndarray, size: int) -> np.ndarray:
    """img (H, W, 3) uint8 -> (size, size, 3) uint8, Image.resize((size, size)) semantics."""
    H, W, _ = img.shape
    if (H, W) == (size, size):
        return img.copy()
    src = img.astype(np.int64)
    tmp = np.empty((H, size, 3), np.uint8)
    for xo, (xmin, kk) in enumerate(pil_coeffs(W, size)):
        tmp[:, xo, :] = _clip8((1 << 21) + np.tensordot(src[:, xmin:xmin + len(kk), :], kk, axes=([1], [0])))
    t = tmp.astype(np.int64)
    out = np.empty((size, size, 3), np.uint8)
    for yo, (ymin, kk) in enumerate(pil_coeffs(H, size)):
        out[yo] = _clip8((1 << 21) + np.tensordot(t[ymin:ymin + len(kk)], kk, axes=([0], [0])))
    return out


def aa_coeffs(in_size, out_size):
    f = np.float32
    scale = f(in_size) / f(out_size)
    support = f(scale) if scale >= 1.0 else f(1.0)
    invscale = f(1.0) / scale if scale >= 1.0 else f(1.0)
    out = []
    for i in range(out_size):
        center = scale * f(i + 0.5)
        xmin = max(int(center - support + f(0.5)), 0)
        xsize = min(int(center + support + f(0.5)), in_size) - xmin
        w = np.zeros(xsize, np.float32)
        for j in range(xsize):
            x = abs(f(f(j + xmin) - center + f(0.5)) * invscale)
            w[j] = f(1.0) - x if x < 1.0 else f(0.0)
        total = f(0.0)
        for j in range(xsize):
            total = f(total + w[j])
        if total != 0:
            w = (w / total).astype(np.float32)
        out.append((xmin, w))
    return out


def aa_bilinear_f32(img: np.ndarray, size: int) -> np.ndarray:
    """img (H, W, 3) uint8 -> (3, size, size) float32 in [0, 1]: ToTensor, then the antialiased bilinear resize."""
    H, W, _ = img.shape
    x = img.astype(np.float32) / np.float32(255.0)
    tmp = np.empty((H, size, 3), np.float32)
    for xo, (xmin, w) in enumerate(aa_coeffs(W, size)):
        t = x[:, xmin, :] * w[0]
        for j in range(1, len(w)):
            t = t + x[:, xmin + j, :] * w[j]
        tmp[:, xo, :] = t
    out = np.empty((size, size, 3), np.float32)
    for yo, (ymin, w) in enumerate(aa_coeffs(H, size)):
        t = tmp[ymin] * w[0]
        for j in range(1, len(w)):
            t = t + tmp[ymin + j] * w[j]
        out[yo] = t
    return np.ascontiguousarray(out.transpose(2, 0, 1))
