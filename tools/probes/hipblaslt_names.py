#!/usr/bin/env python3
"""Which hipBLASLt kernels serve the dense fp16 yardstick (torch.matmul) at the bench's shapes: run under
rocprofv3 --kernel-trace --stats and read the macro-tile (MTmxnxk) and split (GSU / SK) fields of the kernel names."""
import sys
import torch

ms = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2048,1024").split(",")]
dev = torch.device("cuda:0")
for M in ms:
    for N, K in [(4096, 4096), (11008, 4096), (4096, 11008)]:
        x = torch.randn(M, K, device=dev).half()
        w = torch.randn(N, K, device=dev).half()
        for _ in range(5):
            y = torch.matmul(x, w.t())
        torch.cuda.synchronize()
        print(M, N, K, flush=True)
