#!/usr/bin/env python
"""Which hipBLASLt kernels serve the 1x1 layers' GEMM shapes (run under rocprofv3 --kernel-trace --stats: the kernel
names carry the library's macro tile, K depth, LDS / prefetch options)."""
import torch

SHAPES = [(278784, 256, 1024), (278784, 1024, 256), (540800, 768, 768), (278784, 2048, 512), (1081600, 512, 128)]
for m, n, k in SHAPES:
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    b = torch.randn(k, n, device="cuda").to(torch.bfloat16)
    for _ in range(5):
        c = torch.matmul(a, b)
    torch.cuda.synchronize()
    print(m, n, k, "done", flush=True)
