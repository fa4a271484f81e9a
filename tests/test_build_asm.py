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


def _asm(src, *extra):
    out = os.path.join(ROOT, "tests", "_build", src.replace(".hip", ".s"))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                           *extra, "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S",
                           os.path.join(CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernel_body(s, pattern):
    m = re.search(r"^(\S*" + pattern + r"\S*):", s, flags=re.M)
    assert m, pattern + " not found"
    body = s[m.end():]
    return body[:body.index(".Lfunc_end")].splitlines()


def _innermost_loop(lines, at):
    """All instructions of the innermost loop around line `at`, from the compiler's own loop annotations on the
    basic-block labels ('This Inner Loop Header' / 'in Loop: Header=BBx_y Depth=d'); rotated loops and loops with
    several latches come out whole."""
    blocks = []                      # (first line, header id or None, depth)
    for i, l in enumerate(lines):
        m = re.match(r"^(?:\.L(BB\d+_\d+):|; %bb\.\d+:)(.*)$", l)
        if not m:
            continue
        own, rest = m.group(1), m.group(2)
        if i + 1 < len(lines) and re.match(r"^\s*;\s*(=>|Parent Loop|in Loop|Child Loop)", lines[i + 1]):
            rest += " " + lines[i + 1]            # the annotation may continue on the next line(s)
            if i + 2 < len(lines) and re.match(r"^\s*;\s*(=>|Parent Loop|in Loop|Child Loop)", lines[i + 2]):
                rest += " " + lines[i + 2]
        hdr = depth = None
        mh = re.search(r"This Inner Loop Header: Depth=(\d+)", rest)
        if mh and own:
            hdr, depth = own, int(mh.group(1))
        else:
            mi = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", rest)
            if mi:
                hdr, depth = mi.group(1), int(mi.group(2))
        blocks.append((i, hdr, depth))
    assert blocks, "no basic blocks found"
    bounds = [b[0] for b in blocks] + [len(lines)]
    mine = max(k for k, b in enumerate(blocks) if b[0] <= at)
    hdr = blocks[mine][1]
    assert hdr, "line is not inside a loop"
    body = []
    for k, (start, h, _d) in enumerate(blocks):
        if h == hdr:
            body += lines[start:bounds[k + 1]]
    return body


def test_gemm8_main_loops_keep_dma_in_flight():
    """Product kernel = the <ABL 0, mixed layout> instantiation.  The dequant waves' steady-state loop (one burst +
    3 K-steps per iteration) loads its packed words straight from global memory (no LDS-DMA), converts a whole group
    of 3 chunks behind ONE wait and keeps its fp32 arithmetic scalar; the MFMA waves' K-step (32 MFMAs, 16 fragment
    reads, 4 x DMAs) ends in its own counted wait, carries (almost) no VALU work and is never drained."""
    lines = _kernel_body(_asm("gemm8.hip", "-fno-slp-vectorize"), "mxq_gemm8_f16_kernelILi0ELi0E")
    starts = [i for i, l in enumerate(lines) if re.search(r"buffer_load_dword v\d+, v\d+, s\[\d+:\d+\], s\d+ offen", l)]
    loops = []
    for at in starts:
        try:
            lp = _innermost_loop(lines, at)
        except AssertionError:
            continue
        if sum(bool(re.search(r"\bs_barrier\b", l)) for l in lp) == 3 and lp not in loops:
            loops.append(lp)
    assert loops, "expected the dequant waves' burst + 3-step steady-state loop"
    for lp in loops:
        n_loads = sum(bool(re.search(r"buffer_load_(dword|ushort|dwordx2|dwordx4) ", l)) and " lds" not in l for l in lp)
        assert n_loads >= 18 and n_loads % 3 == 0, n_loads          # 6+ loads per chunk, 3 chunks per group
        assert sum("ds_write_b128" in l for l in lp) == 12          # 3 chunks x 4 x 16 bytes per thread
        assert not [l for l in lp if re.search(r"buffer_load_dword.* lds", l)], "LDS-DMA in the dequant waves' loop"
        assert not [l for l in lp if "v_pk_mul_f32" in l or "v_pk_add_f32" in l], "SLP-packed fp32 ops in the dequant"
    waits = [i for i, l in enumerate(lines) if "s_waitcnt vmcnt(4) lgkmcnt(0)" in l]
    steady = [lp for lp in (_innermost_loop(lines, at) for at in waits)
              if sum("v_mfma_f32_16x16x32_f16" in l for l in lp) == 32]
    assert steady, "expected the 32-MFMA steady-state K-step of the MFMA waves"
    for lp in steady:
        assert not [l for l in lp if re.search(r"s_waitcnt.*vmcnt\(0\)", l)], "drain inside the MFMA K-step"
        assert sum("ds_read_b128" in l for l in lp) == 16
        assert sum(bool(re.search(r"buffer_load_dwordx4.* lds", l)) for l in lp) == 4
        assert sum(bool(re.match(r"\s+v_(?!mfma)", l)) for l in lp) <= 12, "VALU work crept into the MFMA waves' loop"


def test_dense256_pair_loop_is_the_designed_schedule():
    """The 256 x 256 kernel's steady-state loop is one K-tile PAIR = 8 phases: per phase 16 MFMAs between two barriers, 2
    LDS-DMA pieces behind ONE counted wait (vmcnt(10): five units stay in flight) and 8 or 4 fragment reads; nothing in the
    loop drains the DMA queue, and the loop carries no VALU instruction at all (no accumulator copies)."""
    lines = _kernel_body(_asm("dense256.hip"), "mxq_dense256_f16_kernel")
    waits = [i for i, l in enumerate(lines) if "s_waitcnt vmcnt(10)" in l]
    loops = []
    for at in waits:
        try:
            lp = _innermost_loop(lines, at)
        except AssertionError:
            continue
        if lp not in loops:
            loops.append(lp)
    loops = [lp for lp in loops if sum("v_mfma_f32_16x16x32_f16" in l for l in lp) == 128]
    assert 1 <= len(loops) <= 2, "expected the K-tile pair loop (and the tile loop around it, which holds the last pair)"
    for lp in loops:
        assert sum(bool(re.search(r"\bs_barrier\b", l)) for l in lp) == 16
        assert sum("s_waitcnt vmcnt(10)" in l for l in lp) == 8
        assert sum(bool(re.search(r"buffer_load_dwordx4.* lds", l)) for l in lp) == 16
        assert sum("ds_read_b128" in l for l in lp) == 48                      # (8 + 4 + 8 + 4) x 2 K-tiles
        assert not [l for l in lp if re.search(r"s_waitcnt.*vmcnt\(0\)", l)], "drain inside the loop"
    inner = min(loops, key=len)          # the pair loop proper
    assert not [l for l in inner if re.match(r"\s+v_(?!mfma)", l)], "VALU work (accumulator copies?) inside the pair loop"


@pytest.mark.parametrize("src", ["gemm.hip", "gemm8.hip", "gemm8h.hip", "gemm8n.hip", "gemm8q.hip", "dense256.hip", "midm.hip", "gemv.hip",
                                 "skinny.hip", "fakequant.hip", "actquant.hip", "pack.hip"])
def test_no_spills_no_scratch(src):
    s = _asm(src, *(["-fno-slp-vectorize", "-I" + CSRC] if src.startswith("gemm8") else []))     # the Makefile's per-file flag
    for m in re.finditer(r"\.vgpr_spill_count:\s+(\d+)", s):
        assert int(m.group(1)) == 0
    for m in re.finditer(r"\.private_segment_fixed_size:\s+(\d+)", s):
        assert int(m.group(1)) == 0
