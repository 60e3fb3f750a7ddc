#!/usr/bin/env python3
"""What the stock BLAS stack (torch.mm -> hipBLASLt / rocBLAS, bf16 in, bf16 out) reaches on the plain GEMM shapes of the Swin-T
branch at 192 frames -- the yardstick for this library's own kernels on the same shapes (tools/bench_gemm.py; no epilogue, no
fp32 output here: the library's forward writes bias / GELU / residual in its epilogue and its weight gradient is float32)."""
import torch

dev = "cuda:0"
SH = [("s0", 602112, 128, 384), ("s0 fc2", 602112, 384, 128), ("s1 qkv", 150528, 192, 576), ("s1 fc1", 150528, 192, 768), ("s1 fc2", 150528, 768, 192),
      ("s2 qkv", 37632, 384, 1152), ("s2 proj", 37632, 384, 384), ("s2 fc1", 37632, 384, 1536), ("s2 fc2", 37632, 1536, 384),
      ("s3 qkv", 9408, 768, 2304), ("s3 fc1", 9408, 768, 3072), ("s3 fc2", 9408, 3072, 768)]


def timed(fn, n=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("shape (M x K -> N)              |  fwd ms   TF/s |  dgrad ms  TF/s |  wgrad ms  TF/s")
for name, M, K, N in SH:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()
    dy = torch.randn(M, N, device=dev).bfloat16()
    fl = 2.0 * M * K * N
    tf = timed(lambda: torch.mm(x, w.t()))
    td = timed(lambda: torch.mm(dy, w))
    tw = timed(lambda: torch.mm(dy.t(), x))
    print(f"{name:8s} {M:7d} x {K:4d} -> {N:4d} | {tf:7.3f} {fl / tf / 1e9:6.0f} | {td:8.3f} {fl / td / 1e9:6.0f} | {tw:8.3f} {fl / tw / 1e9:6.0f}")
