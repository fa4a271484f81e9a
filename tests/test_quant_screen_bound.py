"""The rounding-boundary screen of the quantise kernels (csrc/pack.hip quant_key), checked on the CPU in numpy fp32.

The kernels compute q = clamp(rne(x / s + z), 0, maxq) (reference mxq_quant/lib/quantizer.py:14-16) as
clamp(rne(med3(x * rcp(s) + z, -1, maxq + 1))) whenever every value of a wave keeps its clamped sum farther than
|x * rcp(s)| * 2^-21 + 2^-19 from the nearest k + 0.5, and with the IEEE division otherwise.  v_rcp_f32 is specified to
1 ulp; numpy's reciprocal is correctly rounded, so the test also runs it one ulp up and one ulp down: for EVERY value
whose key is positive under a reciprocal, the fast code must equal the exact one.  The inputs stress what the bound is
made of: ties and their one-ulp neighbours, groups riding on large offsets (|x / s| up to ~1e4), tiny / huge scales."""
import numpy as np
import pytest

F32 = np.float32


def _exact(x, s, z, maxq):
    return np.clip(np.rint((x / s).astype(F32) + z).astype(F32), 0, maxq)


def _fast(x, r, z, maxq):
    a = (x * r).astype(F32)
    t = np.clip((a + z).astype(F32), F32(-1), F32(maxq + 1))          # v_med3_f32
    n = np.rint(t).astype(F32)
    q = np.clip(n, 0, maxq)
    key = (F32(0.5) - np.abs((t - n).astype(F32))).astype(F32) - (np.abs(a) * F32(2.0 ** -21) + F32(2.0 ** -19)).astype(F32)
    return q, key


def _cases(rng, n):
    step = (2.0 ** rng.integers(-14, 6, n)).astype(F32)
    k = rng.integers(-2, 40, n).astype(F32) * F32(0.5)                # multiples of half a step: ties
    off = rng.choice([0.0, 0.0, 37.0, -517.0, 1000.0, 9000.0], n).astype(F32)
    x = ((off + k) * step).astype(F32)
    kind = rng.integers(0, 4, n)
    x = np.where(kind == 1, np.nextafter(x, F32(np.inf)), x)
    x = np.where(kind == 2, np.nextafter(x, F32(-np.inf)), x)
    x = np.where(kind == 3, (x + rng.standard_normal(n).astype(F32) * step).astype(F32), x)
    s = (step * (1 + rng.integers(0, 3, n).astype(F32) * F32(1e-3))).astype(F32)     # the scale is not exactly the data's step
    z = (-(off * step) / s).astype(F32)
    return x.astype(F32), np.maximum(s, F32(1e-9)), z


@pytest.mark.parametrize("maxq", [3, 15])
def test_screen_never_passes_a_wrong_code(maxq):
    rng = np.random.default_rng(11 + maxq)
    x, s, z = _cases(rng, 2_000_000)
    want = _exact(x, s, z, maxq)
    r0 = (F32(1.0) / s).astype(F32)
    passed = 0
    for r in (r0, np.nextafter(r0, F32(np.inf)), np.nextafter(r0, F32(0))):
        q, key = _fast(x, r, z, maxq)
        ok = key > 0                                                   # (NaN / Inf keys compare false: fallback)
        assert np.array_equal(q[ok], want[ok]), f"{np.count_nonzero(q[ok] != want[ok])} wrong codes behind a positive key"
        passed += np.count_nonzero(ok)
    assert passed > 1_000_000                                          # the screen is not vacuous: most values take the fast path


def test_screen_rejects_ties_and_nonfinite():
    s = np.full(8, 0.25, F32)
    z = np.full(8, 1.0, F32)
    x = np.array([0.125, 0.375, -0.125, np.nan, np.inf, -np.inf, 0.1249999, 1e30], F32)   # ties at k + 0.5, non-finite, huge
    _, key = _fast(x, (F32(1.0) / s).astype(F32), z, 3)
    assert not (key[:6] > 0).any()
    assert not key[7] > 0
