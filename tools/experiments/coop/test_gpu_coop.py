"""Cooperative-dequant mode of the prefill GEMM (csrc/gemm8.hip "coop"): the workgroups of a weight panel convert each
128 x 64 weight tile once between them into an fp16 image in the workspace and stream it from there.  The products and
their summation order are those of the fused mode, so the two must agree BIT FOR BIT; the checks below also cover what
is new in the mode: the launch epoch (many launches on one workspace, no memset in between), different weights
sharing the workspace back to back, ragged M / N edges, K that is not a multiple of the 8-step production octet,
hipGraph replay, and a second stream with its own workspace.

Counterpart in the reference: the implicit nn.Linear on the fake-quant weight (mxq_quant/main.py:85); arithmetic
contract lib/quantizer.py:19-20 + mxqgpt.py:448."""
import pytest
import torch

pytestmark = pytest.mark.gpu

REL_TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _case(dev, M, N, K, seed):
    from mxq_amd import packing
    g = torch.Generator(device=dev).manual_seed(seed)
    p = packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half())
    x = torch.randn(M, K, generator=g, device=dev).half()
    return p, x


def _ref(x, p):
    from mxq_amd import packing
    return x.float() @ packing.dequant(p).float().t()


@pytest.mark.parametrize("M,N,K", [(2048, 4096, 4096), (2048, 11008, 4096), (2048, 4096, 11008),
                                   (2048, 1024, 1536),     # NT = 24: the smallest K the mode takes; one partial panel round
                                   (2300, 4224, 1600),     # ragged M (9 token tiles, the last one 252 rows), NT = 25
                                   (4096, 2000, 2048)])    # N not a multiple of 128 (last panel: 80 live rows of 128)
def test_coop_is_bit_identical_to_the_fused_mode(dev, M, N, K):
    from mxq_amd import packing
    p, x = _case(dev, M, N, K, seed=M + N + K)
    ref = _ref(x, p)
    fused = packing.linear(x, p, path="nocoop")
    coop = packing.linear(x, p, path="gemm8")
    torch.cuda.synchronize()
    assert ((coop.float() - ref).abs().max() / ref.abs().max()).item() <= REL_TOL
    assert torch.equal(coop, fused)


def test_coop_many_launches_and_weights_share_one_workspace(dev):
    """Epochs: 40 launches alternating between three weights of different shapes (the scratch image and the flags of
    one launch are stale garbage for the next) -- every result equals the fused mode's."""
    from mxq_amd import packing
    cases = [_case(dev, 2048, 4096, 4096, 1), _case(dev, 2048, 2048, 4096, 2), _case(dev, 2048, 4096, 2048, 3)]
    want = [packing.linear(x, p, path="nocoop") for p, x in cases]
    for it in range(40):
        p, x = cases[it % 3]
        got = packing.linear(x, p, path="gemm8")
        assert torch.equal(got, want[it % 3]), it
    # same weight, different activations, back to back without a host sync in between
    p, x = cases[0]
    xs = [torch.roll(x, i, 0) for i in range(6)]
    outs = [packing.linear(xi, p, path="gemm8") for xi in xs]
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert torch.equal(o, torch.roll(want[0], i, 0)), i


def test_coop_in_graphs_and_on_a_second_stream(dev):
    from mxq_amd import packing
    p, x = _case(dev, 2048, 4096, 4096, 7)
    want = packing.linear(x, p, path="nocoop")
    out = torch.empty_like(want)
    packing.linear(x, p, out=out, path="gemm8")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(3):
            packing.linear(x, p, out=out, path="gemm8")
    for _ in range(4):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want)
    s2 = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(s2):
        o2 = packing.linear(x, p, path="gemm8")
    o1 = packing.linear(x, p, path="gemm8")
    torch.cuda.synchronize()
    assert torch.equal(o1, want) and torch.equal(o2, want)


def test_coop_workspace_sizes(dev):
    from mxq_amd import _lib
    lib = _lib.load()
    base = lib.mxq_gemm_workspace_bytes()
    head = lib.mxq_gemm_workspace_head_bytes()
    assert 0 < head < base
    assert lib.mxq_gemm_workspace_bytes_for(4096, 4096) == base            # 32 MB image: fits the stream-K slots' room
    assert lib.mxq_gemm_workspace_bytes_for(11008, 4096) == head + 2 * 11008 * 4096
    assert lib.mxq_gemm_workspace_bytes_for(4096, 64) == base              # too short a K for the mode
