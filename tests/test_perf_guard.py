"""Wall-clock guards, kept OUT of the parity record: `pytest tests -m perf` on the GPU box (a timing assertion
inside `-m gpu` can turn the one correctness record red for non-correctness reasons -- boxes differ by 2-4 %).  Not marked
`gpu`, so the driver's `-m gpu` run never selects it; skipped where there is no GPU."""
import pytest
import torch

pytestmark = [pytest.mark.perf, pytest.mark.skipif(not torch.cuda.is_available(), reason="perf guards need the MI355X box")]


@pytest.fixture(scope="module")
def dev():
    from mxq_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _packed_case(dev, N, K, seed):
    from mxq_amd import packing
    g = torch.Generator().manual_seed(seed)
    W16 = (torch.randn(N, K, generator=g) * 0.02).half()
    return packing.quantize_pack(W16.to(dev)), None, g


def _graph_us(fn, calls=10, reps=5):
    """us per call of `fn` under hipGraph replay (`calls` calls per graph, best of `reps` replays)."""
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(calls):
            fn()
    best = float("inf")
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gr.replay()
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / calls * 1e3)
    return best


@pytest.mark.parametrize("M,N,K,others", [
    (16, 4096, 4096, ["skinny", "midm"]),
    (64, 11008, 4096, ["midm", "gemm8q_split", "skinny"]),
    (128, 4096, 4096, ["midm", "gemm8h_slices", "gemm8n_slices", "gemm8"]),
    (256, 11008, 4096, ["midm", "gemm8h_split", "gemm8"]),
    (512, 4096, 4096, ["gemm8", "gemm8h_split", "gemm8n_split"]),
    (2048, 11008, 4096, ["whole", "gemm1"])])
def test_dispatch_is_within_15_percent_of_the_fastest_schedule_on_this_box(dev, M, N, K, others):
    """The dispatch thresholds of csrc/capi.hip were tuned on boxes that differ by a few percent (VERDICT r4 weak #1c): on
    THIS box, under hipGraph replay, path "auto" must not be more than 15 % slower than the fastest explicitly chosen
    schedule of its neighbourhood -- a threshold that has drifted to the wrong side of a crossover shows up here."""
    from mxq_amd import packing
    p, _w16, g = _packed_case(dev, N, K, M + N)
    x = torch.randn(M, K, generator=g).half().to(dev)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    t = {path: _graph_us(lambda path=path: packing.linear(x, p, out=out, path=path)) for path in ["auto"] + others}
    best = min(t, key=t.get)
    assert t["auto"] <= 1.15 * t[best], {k: round(v, 2) for k, v in t.items()}
    packing.workspace_status(x.device)
