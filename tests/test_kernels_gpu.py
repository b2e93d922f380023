"""GPU: single HIP kernels through the C ABI (sam2mi_debug_*) vs plain PyTorch fp32 on the same inputs.
Operands are rounded to f16 on both sides, so the tolerance only has to cover accumulation order."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from gpu_util import check

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from sam2_opt_amd.native import Engine
    e = Engine("large", state_dict=None, max_batch=1)
    yield e
    e.close()


def r16(x):
    return x.half().float()


@pytest.mark.parametrize("M,N,K,act,res", [
    (300, 200, 144, 0, False), (4096, 1728, 576, 0, False), (16384, 432, 144, 1, True), (8, 256, 256, 2, False),
    (1000, 64, 160, 0, True), (4096, 4, 32, 0, False), (129, 65, 2304, 1, True), (4096, 576, 2304, 0, True),
])
@pytest.mark.parametrize("hint", [0, 5, 9, 10, 13, 16])   # automatic, then the production tiles forced: 64x64, 64x64 4-stage, 128x128, 128x64, 128x192
def test_gemm(eng, M, N, K, act, res, hint):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    A = r16(torch.randn(M, K, generator=g)).cuda()
    W = r16(torch.randn(N, K, generator=g) / math.sqrt(K)).cuda()
    b = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).cuda() if res else None
    out = eng.debug_gemm(A, W, b, act, R, tile_hint=hint)
    ref = A.double() @ W.double().t() + b.double()
    if act == 1:
        ref = F.gelu(ref)
    elif act == 2:
        ref = F.relu(ref)
    if res:
        ref = ref + R.double()
    check(f"gemm {M}x{N}x{K} act{act} hint{hint}", out, ref.float(), 1e-4, 1e-5)


def _ref_attn(q, k, v, groups, heads, GQ, GK, wq, wk):
    C = heads * 72
    qh = q.view(groups, GQ, heads, 72).permute(0, 2, 1, 3).double()
    kh = k.view(groups, GK, heads, 72).permute(0, 2, 1, 3).double()
    vh = v.view(groups, GK, heads, 72).permute(0, 2, 1, 3).double()
    s = qh @ kh.transpose(-1, -2) / math.sqrt(72)
    qi = torch.arange(GQ, device=q.device) // wq
    ki = torch.arange(GK, device=q.device) // wk
    mask = qi[:, None] == ki[None, :]
    s = s.masked_fill(~mask, float("-inf"))
    o = torch.softmax(s, -1) @ vh
    return o.permute(0, 2, 1, 3).reshape(groups * GQ, C).float()


@pytest.mark.parametrize("groups,heads,GQ,GK,wq,wk", [
    (6, 2, 64, 64, 64, 64),        # stage-1 8x8 windows
    (5, 4, 32, 128, 16, 64),       # block 2: pooled queries, 2 windows packed
    (9, 4, 32, 32, 16, 16),        # stage-2 4x4 windows, 2 packed
    (3, 8, 32, 128, 4, 16),        # block 8: 8 windows packed
    (2, 8, 256, 256, 256, 256),    # stage-3 16x16 windows (shared staging)
    (1, 8, 1024, 1024, 1024, 1024),  # global attention (reduced length)
    (2, 16, 64, 256, 64, 256),     # block 44
])
def test_hiera_attention(eng, groups, heads, GQ, GK, wq, wk):
    g = torch.Generator(device="cpu").manual_seed(groups * 100 + GQ + GK)
    C = heads * 72
    q = r16(torch.randn(groups * GQ, C, generator=g) * 1.5).cuda()
    k = r16(torch.randn(groups * GK, C, generator=g) * 1.5).cuda()
    v = r16(torch.randn(groups * GK, C, generator=g)).cuda()
    out = eng.debug_hiera_attention(q, k, v, groups, heads, GQ, GK, wq, wk)
    ref = _ref_attn(q, k, v, groups, heads, GQ, GK, wq, wk)
    check(f"hiera_attn g{groups} h{heads} {GQ}/{GK} w{wq}/{wk}", out, ref, 4e-3, 2e-3)


@pytest.mark.parametrize("Nq,Nk", [(128, 32), (256, 4096), (128, 4100), (256, 8204), (128, 28736 + 64)])
def test_flash256(eng, Nq, Nk):
    g = torch.Generator(device="cpu").manual_seed(Nq + Nk)
    q = r16(torch.randn(Nq, 256, generator=g)).cuda()
    k = r16(torch.randn(Nk, 256, generator=g)).cuda()
    v = r16(torch.randn(Nk, 256, generator=g)).cuda()
    # spike one key so that the running max jumps late in the sweep (online-softmax rescale path)
    k[Nk - 3] = r16(q[5] * 2.0)
    out = eng.debug_flash256(q, k, v)
    s = (q.double() @ k.double().t()) / 16.0
    ref = (torch.softmax(s, -1) @ v.double()).float()
    check(f"flash256 {Nq}x{Nk}", out, ref, 4e-3, 2e-3)


@pytest.mark.parametrize("M,C", [(256, 144), (1000, 144), (128, 288), (777, 288), (8192, 144), (4096, 288)])
def test_mlp_fused(eng, M, C):
    """Fused fc1-GELU-fc2-residual kernel (stages 1-2 of Hiera, hieradet.py:163-165) vs fp64 PyTorch on f16-rounded operands
    (the hidden activation is rounded to f16 between the two products, exactly as the two-GEMM path stores it) and vs
    that two-GEMM path."""
    g = torch.Generator(device="cpu").manual_seed(M + C)
    xn = r16(torch.randn(M, C, generator=g)).cuda()
    W1 = r16(torch.randn(4 * C, C, generator=g) / math.sqrt(C)).cuda()
    b1 = (0.3 * torch.randn(4 * C, generator=g)).cuda()
    W2 = r16(torch.randn(C, 4 * C, generator=g) / math.sqrt(4 * C)).cuda()
    b2 = (0.3 * torch.randn(C, generator=g)).cuda()
    x = torch.randn(M, C, generator=g).cuda()
    out, _ = eng.debug_mlp(xn, W1, b1, W2, b2, x, fused=True)
    h = F.gelu(xn.double() @ W1.double().t() + b1.double()).half().double()
    ref = x.double() + h @ W2.double().t() + b2.double()
    check(f"mlp_fused {M}x{C}", out, ref.float(), 1e-4, 2e-4)
    two, _ = eng.debug_mlp(xn, W1, b1, W2, b2, x, fused=False)
    check(f"mlp_fused vs two-GEMM {M}x{C}", out, two, 1e-4, 2e-4)


@pytest.mark.parametrize("M,N,K,act,res", [
    (4096, 1728, 576, 0, False), (1000, 576, 576, 0, True), (4096, 2304, 576, 1, False), (2048, 432, 144, 0, False),
    (3000, 864, 288, 0, True), (512, 1152, 288, 1, False), (130, 144, 144, 0, True), (256, 40, 576, 0, False),
])
def test_gemm_xs(eng, M, N, K, act, res):
    """X-stationary short-K GEMM (gemm_xs.hip; QKV / projection / fc1 of Hiera stages 1-3) vs fp64 PyTorch on f16-rounded
    operands: row-major f32 output, optional erf-GELU and residual, ragged M and N (N % 8 == 0)."""
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K + 1)
    A = r16(torch.randn(M, K, generator=g)).cuda()
    W = r16(torch.randn(N, K, generator=g) / math.sqrt(K)).cuda()
    b = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).cuda() if res else None
    out = eng.debug_gemm(A, W, b, act, R, tile_hint=30)
    ref = A.double() @ W.double().t() + b.double()
    if act == 1:
        ref = F.gelu(ref)
    if res:
        ref = ref + R.double()
    check(f"gemm_xs {M}x{N}x{K} act{act}", out, ref.float(), 1e-4, 1e-5)


@pytest.mark.parametrize("M,K,res", [(4096, 2304, True), (1000, 576, True), (128, 128, False), (333, 1152, False)])
def test_gemm_ks(eng, M, K, res):
    if not os.environ.get("SAM2MI_EXPERIMENTAL"):
        pytest.skip("gemm_ks.hip is only built with SAM2MI_EXPERIMENTAL=1 (measured equal to the tiled kernel, not shipped)")
    """Accumulator-stationary N = 576 GEMM (gemm_ks.hip; projection / fc2 of Hiera stage 3) vs fp64 PyTorch on f16-rounded
    operands: f32 output with bias and optional residual, ragged M."""
    N = 576
    g = torch.Generator(device="cpu").manual_seed(M * 7 + K)
    A = r16(torch.randn(M, K, generator=g)).cuda()
    W = r16(torch.randn(N, K, generator=g) / math.sqrt(K)).cuda()
    b = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).cuda() if res else None
    out = eng.debug_gemm(A, W, b, 0, R, tile_hint=31)
    ref = A.double() @ W.double().t() + b.double()
    if res:
        ref = ref + R.double()
    check(f"gemm_ks {M}x{N}x{K}", out, ref.float(), 1e-4, 1e-5)


@pytest.mark.parametrize("w", [4, 8, 16])
@pytest.mark.parametrize("M,N,K,hint", [(8192, 288, 144, 0), (4096, 576, 288, 13), (2048, 1152, 576, 5), (1024, 64, 64, 10)])
def test_gemm_fused_maxpool(eng, M, N, K, hint, w):
    """2x2 max-pool of window-major token rows inside the GEMM epilogue (the shortcut of a Hiera transition block,
    hieradet.py:139-140: do_pool(self.proj(x), self.pool)) against torch: GEMM, then F.max_pool2d per window."""
    g = torch.Generator(device="cpu").manual_seed(M + N + w)
    A = r16(torch.randn(M, K, generator=g)).cuda()
    W = r16(torch.randn(N, K, generator=g) / math.sqrt(K)).cuda()
    b = torch.randn(N, generator=g).cuda()
    out = eng.debug_gemm(A, W, b, 0, None, tile_hint=hint, pool_w=w)
    full = (A.double() @ W.double().t() + b.double()).float()
    ref = F.max_pool2d(full.view(M // (w * w), w, w, N).permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).reshape(M // 4, N)
    check(f"gemm+pool {M}x{N}x{K} w{w} hint{hint}", out, ref, 1e-4, 1e-5)


@pytest.mark.parametrize("M,C", [(4096, 144), (2048, 288), (1024, 576), (32, 576)])
def test_projln_fused_projection_norm2(eng, M, C):
    """gemm_projln_kernel (Hiera out-projection + residual + norm2, hieradet.py:161-165) against torch in f64 on f16-rounded operands."""
    g = torch.Generator(device="cpu").manual_seed(M + C)
    a = r16(torch.randn(M, C, generator=g)).cuda()
    W = r16(torch.randn(C, C, generator=g) / math.sqrt(C)).cuda()
    b, x = torch.randn(C, generator=g).cuda(), (3 * torch.randn(M, C, generator=g)).cuda()
    lw, lb = (1 + 0.1 * torch.randn(C, generator=g)).cuda(), (0.1 * torch.randn(C, generator=g)).cuda()
    x_ref = x.double() + a.double() @ W.double().t() + b.double()
    h_ref = F.layer_norm(x_ref, (C,), lw.double(), lb.double(), 1e-6)
    x_out, h_out = eng.debug_projln(a, W, b, x, lw, lb)
    check(f"projln x {M}x{C}", x_out, x_ref.float(), 1e-4, 1e-5)
    check(f"projln h {M}x{C}", h_out, h_ref.float(), 2e-3, 2e-3)      # f16 output


@pytest.mark.parametrize("splits", [0, 1, 8])
def test_rowln_fused_attention_tail(eng, splits):
    """gemm_rowln_kernel (combine of the flash partials + out-projection + residual + LayerNorm) against torch in f64."""
    g = torch.Generator(device="cpu").manual_seed(11 + splits)
    M = 4096
    W = r16(torch.randn(256, 256, generator=g) / 16).cuda()
    b, x = torch.randn(256, generator=g).cuda(), torch.randn(M, 256, generator=g).cuda()
    lw, lb = (1 + 0.1 * torch.randn(256, generator=g)).cuda(), (0.1 * torch.randn(256, generator=g)).cuda()
    if splits == 0:
        a, ml = r16(torch.randn(M, 256, generator=g)).cuda(), None
        o = a.double()
    else:
        a = torch.randn(splits, M, 256, generator=g).cuda() * 3
        ml = torch.stack([torch.randn(splits, M, generator=g) * 4, torch.rand(splits, M, generator=g) * 5 + 0.5], dim=-1).cuda()
        w = torch.exp2(ml[..., 0].double() - ml[..., 0].double().max(dim=0, keepdim=True).values)            # (splits, M)
        o = (w[..., None] * a.double()).sum(0) / (w * ml[..., 1].double()).sum(0)[..., None]
        o = o.float().half().double()                                                                     # the kernel's f16 operand
    x_ref = x.double() + o @ W.double().t() + b.double()
    h_ref = F.layer_norm(x_ref, (256,), lw.double(), lb.double(), 1e-5)
    x_out, h_out = eng.debug_rowln(a, ml, W, b, x, lw, lb)
    check(f"rowln x splits{splits}", x_out, x_ref.float(), 2e-4, 2e-5)
    check(f"rowln h splits{splits}", h_out, h_ref.float(), 2e-3, 2e-3)      # f16 output


def test_profile_by_kernel_instantiation(eng):
    """sam2mi_profile_read_kernels: the per-instantiation accumulators (what bench.py's roofline object is built from) carry
    the rocprofv3 kernel names and add up to the family totals of sam2mi_profile_read."""
    g = torch.Generator(device="cpu").manual_seed(5)
    A = r16(torch.randn(4096, 256, generator=g)).cuda()
    W = r16(torch.randn(512, 256, generator=g) / 16).cuda()
    b = torch.zeros(512).cuda()
    eng.profile_enable(True)
    for _ in range(3):
        eng.debug_gemm(A, W, b, 0, None)
    tot = eng.profile_read()
    per = eng.profile_read_kernels()
    eng.profile_enable(False)
    assert tot["gemm_launches"] == 3 and len(per) == 1
    (name, v), = per.items()
    assert name.startswith("gemm_v2_kernel<") and v["launches"] == 3
    assert abs(v["ms"] - tot["gemm_ms"]) < 1e-6 and v["flops"] == 3 * 2.0 * 4096 * 512 * 256


def test_reserved_cu_stream(eng):
    """sam2mi_stream_create_reserved: a CU-masked HIP stream is usable as a torch stream and computes the same GEMM."""
    g = torch.Generator(device="cpu").manual_seed(6)
    A = r16(torch.randn(512, 256, generator=g)).cuda()
    W = r16(torch.randn(256, 256, generator=g) / 16).cuda()
    b = torch.randn(256, generator=g).cuda()
    ref = eng.debug_gemm(A, W, b, 0, None)
    st = eng.create_reserved_stream(8)
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        out = eng.debug_gemm(A, W, b, 0, None)
    st.synchronize()
    assert torch.equal(out, ref)
