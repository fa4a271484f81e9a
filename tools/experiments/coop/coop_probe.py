import sys, subprocess, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import torch
    from mxq_amd import packing
    M, N, K = (int(v) for v in sys.argv[1:4])
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    p = packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half())
    x = torch.randn(M, K, generator=g, device=dev).half()
    a = packing.linear(x, p, path="nocoop"); torch.cuda.synchronize()
    b = packing.linear(x, p, path="gemm8"); torch.cuda.synchronize()
    print(M, N, K, "equal" if torch.equal(a, b) else "DIFF %g" % (a.float() - b.float()).abs().max().item(), flush=True)
else:
    for dbg in ["0", "4"]:
        shp = "2048 4224 1536"
        r = subprocess.run([sys.executable, __file__] + shp.split(), capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, MXQ_COOP_DBG=dbg))
        print("dbg", dbg, shp, "rc", r.returncode, r.stdout.strip()[-200:], r.stderr.strip()[-300:].replace("\n", " | "), flush=True)
