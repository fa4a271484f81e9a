"""Build-time checks on the generated gfx950 code (hipcc cross-compiles on the CPU box):
the pipelined GEMM must keep its LDS-DMA prefetch in flight across the K-loop barrier, i.e.
the compiler must not have inserted a draining `s_waitcnt vmcnt(0)` inside the main loop
(see the hipcc note in csrc/gemm2.hip), no kernel may spill, and the MFMA count must match."""
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


def test_gemm2_main_loop_keeps_dma_in_flight():
    s = _asm("gemm2.hip")
    # the product kernel is the ABL = 0 instantiation (the others are profiling-only ablations)
    m = re.search(r"^(\S*mxq_gemm2_f16_kernelILi0E\S*):", s, flags=re.M)
    assert m, "mxq_gemm2_f16_kernel<0> not found"
    body = s[m.end():]
    body = body[:body.index(".Lfunc_end")]
    lines = body.splitlines()
    # every counted wait closes a pipelined K-step (one per wave role): between the previous
    # s_barrier and the counted wait there must be no draining vmcnt(0), and all 16 fragment reads
    idx = [i for i, l in enumerate(lines) if "s_waitcnt vmcnt(5) lgkmcnt(0)" in l]
    assert len(idx) >= 2, "expected a counted steady-state wait per wave role"
    for end in idx:
        start = max(i for i, l in enumerate(lines[:end]) if "s_barrier" in l)
        step = lines[start:end + 1]
        drains = [l for l in step if re.search(r"s_waitcnt.*vmcnt\(0\)", l)]
        assert not drains, f"compiler-inserted drain inside the K loop: {drains}"
        assert sum("ds_read_b128" in l for l in step) == 16
        assert sum("v_mfma_f32_16x16x32_f16" in l for l in step) in (16, 32)   # 16: the peeled first step
        assert sum("global_load_lds_dwordx4" in l for l in step) == 5


@pytest.mark.parametrize("src", ["gemm.hip", "gemm2.hip", "gemm5.hip", "gemm6.hip", "gemv.hip", "fakequant.hip", "pack.hip"])
def test_no_spills_no_scratch(src):
    s = _asm(src)
    for m in re.finditer(r"\.vgpr_spill_count:\s+(\d+)", s):
        assert int(m.group(1)) == 0
    for m in re.finditer(r"\.private_segment_fixed_size:\s+(\d+)", s):
        assert int(m.group(1)) == 0
