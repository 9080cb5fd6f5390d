"""Development yardstick: the library GEMM (torch.matmul -> hipBLASLt) on the encoder's four product shapes, bf16, random
operands.  Not part of the product; the figures say what the hand-written tile kernel should be read against."""
import sys
import torch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 192000
dev = torch.device("cuda", 0)
for name, K, N in (("fc1", 768, 3072), ("qk", 768, 1536), ("out", 768, 768), ("fc2", 3072, 768)):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = (0.03 * torch.randn(N, K, device=dev)).to(torch.bfloat16)
    for _ in range(3):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        c = a @ w.t()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 6
    print(f"{name:5s} M={M} K={K} N={N}  {us:8.1f} us  {2.0 * M * N * K / us * 1e-6:6.0f} TFLOP/s (bf16 out, no epilogue)", flush=True)
