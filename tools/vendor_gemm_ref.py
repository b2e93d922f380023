"""Calibration only: what PyTorch's vendor GEMM (hipBLASLt / rocBLAS behind torch.nn.functional.linear, f16 in/out) reaches on the
model's GEMM shapes - a reference point for the hand-written kernel's numbers in tools/gemm_bench.py.  Not used by the product."""
import torch, time
shapes = [("s1 qkv", 524288, 432, 144), ("s1 fc1", 524288, 576, 144), ("s1 fc2", 524288, 144, 576),
          ("s2 qkv", 131072, 864, 288), ("s2 fc1", 131072, 1152, 288), ("s2 fc2", 131072, 288, 1152),
          ("s3 qkv", 32768, 1728, 576), ("s3 proj", 32768, 576, 576), ("s3 fc1", 32768, 2304, 576), ("s3 fc2", 32768, 576, 2304),
          ("s4 qkv", 8192, 3456, 1152), ("s4 fc1", 8192, 4608, 1152), ("s4 fc2", 8192, 1152, 4608),
          ("ma ff1", 4096, 2048, 256), ("ma ff2", 4096, 256, 2048)]
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K in shapes:
    x = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).half()
    w = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1).half()
    for _ in range(3):
        y = torch.nn.functional.linear(x, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y = torch.nn.functional.linear(x, w)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:8s} M={M:7d} N={N:5d} K={K:5d}: {ms*1e3:8.1f}us {2.0*M*N*K/ms/1e9:7.1f}TF", flush=True)
