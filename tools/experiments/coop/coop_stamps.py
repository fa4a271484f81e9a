import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from mxq_amd import packing
dev = torch.device("cuda:0")
M, N, K = 2048, 4096, 11008
g = torch.Generator(device=dev).manual_seed(1)
p = packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half())
x = torch.randn(M, K, generator=g, device=dev).half()
out = torch.empty(M, N, device=dev, dtype=torch.float16)
for _ in range(5):
    packing.linear(x, p, out=out, path="gemm8")
torch.cuda.synchronize()
for dbg in sys.argv[1:]:
    os.environ["MXQ_COOP_DBG"] = dbg
    for _ in range(20):
        packing.linear(x, p, out=out, path="gemm8")
    torch.cuda.synchronize()
    ws = packing.gemm_workspace(dev, N, K)
    ctl = ws[65536:65536 + 4096].view(torch.int32).cpu()
    NT = K // 64
    dma = ctl[64:80].tolist()
    prod = ctl[80:96].tolist()
    print("dbg", dbg)
    print("  DMA wave  per step: pre-wait work", [round(v / (NT / 4)) for v in dma[0:4]], " wait", [round(v / (NT / 4)) for v in dma[4:8]], " barrier", [round(v / (NT / 4)) for v in dma[8:12]])
    print("  DMA wave: prologue ticks", dma[12], " loop ticks", dma[13], " kernel: realtime(100MHz) ticks", dma[14], " memtime ticks", dma[15],
          " => clock GHz", round(dma[15] / max(dma[14], 1) / 10, 3))
    print("  producer  per step: work", [round(v / (NT / 8)) for v in prod[0:8]], " barrier", [round(v / (NT / 8)) for v in prod[8:16]], flush=True)
