import ctypes, os, sys, torch
ROOT = "/root/repo" if os.path.exists("/root/repo/mxq_amd") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT)
lib = ctypes.CDLL(os.path.join(ROOT, "mxq_amd/libmxq_hip_prof.so"))
fn = lib.mxq_prof_dense256_clock
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
M, N, K = 32768, 4096, 4096
x = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.02).half()
y = torch.empty(M, N, device=dev, dtype=torch.float16); dbg = torch.zeros(2 + 8 * 5, dtype=torch.int64, device=dev)
for _ in range(30):
    assert fn(x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, dbg.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
vals = dbg.cpu().tolist()
mt, rt = vals[0], vals[1]
for wv in range(8):
    sw, sl, sv, sb, n = vals[2 + 5 * wv: 7 + 5 * wv]
    n = max(n, 1)
    print(f"  wave {wv}: per K-step work {sw / n:7.0f}  lgkm wait {sl / n:5.0f}  vm wait {sv / n:6.0f}  barrier {sb / n:6.0f}  total {(sw + sl + sv + sb) / n:7.0f} cycles ({n} steps)")
print(f"dense256e, 32768 x 4096^2: kernel {rt / 100:.1f} us on the 100 MHz clock, {mt} core cycles -> {mt / rt / 10:.2f} GHz held")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): torch.matmul(x, w.t(), out=y)
e1.record(); torch.cuda.synchronize()
print("hipBLASLt us", e0.elapsed_time(e1) / 20 * 1e3)
