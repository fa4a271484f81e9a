"""Build-time checks on the generated gfx950 code (hipcc cross-compiles on the CPU box):
the pipelined GEMM must keep its LDS-DMA prefetch in flight across the K-loop barrier, i.e.
the compiler must not have inserted a draining `s_waitcnt vmcnt(0)` inside the main loop
(see the hipcc notes in DESIGN.md section 4), no kernel may spill, and the MFMA count must match."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mxq_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")


def _asm(src):
    out = os.path.join(ROOT, "tests", "_build", src.replace(".hip", ".s"))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                           "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S",
                           os.path.join(CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernel_body(s, pattern):
    m = re.search(r"^(\S*" + pattern + r"\S*):", s, flags=re.M)
    assert m, pattern + " not found"
    body = s[m.end():]
    return body[:body.index(".Lfunc_end")].splitlines()


def _innermost_loop(lines, at):
    """Smallest [label .. backward branch to that label] range containing line `at`."""
    labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    last = {}                       # label -> last backward branch to it (a loop may have several latches)
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            last[m.group(1)] = i
    best = None
    for lab, end in last.items():
        if labels[lab] <= at <= end and (best is None or end - labels[lab] < best[1] - best[0]):
            best = (labels[lab], end)
    assert best, "no enclosing loop"
    return lines[best[0]:best[1] + 1]


def test_gemm6_main_loops_keep_dma_in_flight():
    """Product kernel = the <ABL 0, mixed layout> instantiation.  Its DMA waves' steady-state K-step ends in the
    hand-placed counted wait (this step's 10 LDS-DMAs stay in flight across the barrier) and contains no
    compiler-inserted drain; its MFMA waves' K-step (32 MFMAs, 16 fragment reads) waits on no vmcnt at all."""
    lines = _kernel_body(_asm("gemm6.hip"), "mxq_gemm6_f16_kernelILi0ELi0E")
    waits = [i for i, l in enumerate(lines) if "s_waitcnt vmcnt(10) lgkmcnt(0)" in l]
    assert waits, "expected the counted steady-state wait of the DMA waves"
    for at in waits:
        loop = _innermost_loop(lines, at)
        assert not [l for l in loop if re.search(r"s_waitcnt.*vmcnt\(0\)", l)], "drain inside the DMA K-step"
        assert sum("global_load_lds_dwordx4" in l for l in loop) == 10
        assert sum(bool(re.search(r"\bs_barrier\b", l)) for l in loop) == 1
    first_mfma = [i for i, l in enumerate(lines) if "v_mfma_f32_16x16x32_f16" in l]
    loops = {id(lp): lp for lp in (_innermost_loop(lines, i) for i in first_mfma[16:48])}   # past the peeled first step
    steady = [lp for lp in loops.values() if sum("v_mfma_f32_16x16x32_f16" in l for l in lp) == 32]
    assert steady, "expected the 32-MFMA steady-state K-step of the MFMA waves"
    for lp in steady:
        assert not [l for l in lp if "vmcnt" in l], "the MFMA waves' K-step must not wait on VMEM"
        assert sum("ds_read_b128" in l for l in lp) == 16


@pytest.mark.parametrize("src", ["gemm.hip", "gemm4.hip", "gemm6.hip", "gemv.hip", "fakequant.hip", "actquant.hip", "pack.hip"])
def test_no_spills_no_scratch(src):
    s = _asm(src)
    for m in re.finditer(r"\.vgpr_spill_count:\s+(\d+)", s):
        assert int(m.group(1)) == 0
    for m in re.finditer(r"\.private_segment_fixed_size:\s+(\d+)", s):
        assert int(m.group(1)) == 0
