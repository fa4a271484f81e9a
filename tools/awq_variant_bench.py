#!/usr/bin/env python3
"""Time mxq_gemm_awq_f16 of several builds of the library in one process (same box, same inputs).
    python tools/awq_variant_bench.py tools/_variants/lib_a.so tools/_variants/lib_b.so ..."""
import ctypes
import sys
import torch

dev = torch.device("cuda:0")
M, IC, OC, G = 2048, 4096, 4096, 128
kern = torch.randint(-2**31, 2**31 - 1, (IC, OC // 8), dtype=torch.int32, device=dev)
zeros = torch.randint(-2**31, 2**31 - 1, (IC // G, OC // 8), dtype=torch.int32, device=dev)
scales = (torch.rand(IC // G, OC, device=dev) * 0.004 + 0.001).half()
x = torch.randn(M, IC, device=dev).half()
y = torch.empty(M, OC, device=dev, dtype=torch.float16)
ws = torch.zeros(80 << 20, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    fn = lib.mxq_gemm_awq_f16
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    call = lambda: fn(x.data_ptr(), kern.data_ptr(), scales.data_ptr(), zeros.data_ptr(), y.data_ptr(), M, IC, OC, G, ws.data_ptr(), ws.numel(), st)
    for _ in range(5):
        assert call() == 0
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    print(f"{path}: {best:.1f} us per launch (M={M} IC={IC} OC={OC})", flush=True)
