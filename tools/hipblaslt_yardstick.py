"""Yardstick only (never on the product path): what the vendor GEMM library reaches, through torch.matmul in f16, on the
encoder's large GEMM shapes - to tell how far gemm_v2_kernel is from a tuned kernel on the same shape (tools/gemm_bench.py)."""
import torch
shapes = [("s3 qkv", 32768, 1728, 576), ("s3 proj", 32768, 576, 576), ("s3 fc1", 32768, 2304, 576), ("s3 fc2", 32768, 576, 2304),
          ("s4 qkv", 8192, 3456, 1152), ("s4 fc1", 8192, 4608, 1152), ("s4 fc2", 8192, 1152, 4608), ("s2 fc1", 131072, 1152, 288),
          ("s2 fc2", 131072, 288, 1152), ("big", 8192, 8192, 8192)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.float16)
    w = torch.randn(N, K, device="cuda", dtype=torch.float16)
    for _ in range(5):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        c = a @ w.t()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:8s} M={M:7d} N={N:5d} K={K:5d}: {ms*1e3:8.1f}us {2.0*M*N*K/ms/1e9:7.1f}TF", flush=True)
