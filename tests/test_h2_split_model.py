"""The arithmetic contract of the 'f16x2' GEMM flavour (csrc/gemm_h2.h, gemm_h2a.h), restated in numpy so that it is checked without
a GPU: a tensor with bound b is scaled by 2^e, e = 15 - ceil(log2 b); every element is hi = f16(x 2^e), lo = f16(x 2^e - hi); a
product is lo.hi + hi.lo + hi.hi accumulated in fp32 and unscaled by 2^-(e_a + e_w).  Checked here: the exponent rule (no overflow,
top binade used), the split's residual (<= 2^-22 of the scaled value, <= 2^-25 absolute where lo is a subnormal), the image layout
([hi x 8 | lo x 8] per 8 elements in the fp32 matrix's byte geometry), that a K = 1000 product is at least as close to fp64 as a
k-ordered fp32 chain, that bounds loose by 2^12 lose nothing measurable, and that scaling an operand by a power of two scales the
result exactly (the property the GPU test of the backward pass's measured bounds relies on)."""
import numpy as np


def exp_of(bound):
    if not bound > 0:
        return 0
    m, e2 = np.frexp(np.float32(bound))
    c = e2 - 1 if m == 0.5 else e2                      # ceil(log2 b)
    return int(np.clip(15 - c, -100, 100))


def split(x, e):
    xs = (x.astype(np.float32) * np.float32(2.0 ** e)).astype(np.float32)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)
    return xs, hi, lo


def image(x, e):
    """fp16-pair image of a row-major fp32 matrix whose row length is a multiple of 8: bytes of [hi x 8 | lo x 8] per group"""
    _, hi, lo = split(x, e)
    g = x.size // 8
    out = np.empty((g, 16), np.float16)
    out[:, :8] = hi.reshape(g, 8)
    out[:, 8:] = lo.reshape(g, 8)
    return out.view(np.float32).reshape(x.shape)        # same shape, same bytes per element as the fp32 matrix


def gemm_h2(a, w, ea, ew):
    """C = A W^T the way the kernels form it: three fp16 products per element pair, fp32 accumulation in units of 2^(ea + ew)"""
    _, ah, al = split(a, ea)
    _, wh, wl = split(w, ew)
    f = lambda t: t.astype(np.float32)
    acc = np.zeros((a.shape[0], w.shape[0]), np.float32)
    for k0 in range(0, a.shape[1], 16):                 # one MFMA k-step: products exact in fp32, summed into the accumulator
        s = slice(k0, k0 + 16)
        for x, y in ((al, wh), (ah, wl), (ah, wh)):
            acc = (acc + (f(x[:, s]).astype(np.float64) @ f(y[:, s]).astype(np.float64).T).astype(np.float32)).astype(np.float32)
    return acc * np.float32(2.0 ** -(ea + ew))


def test_exponent_rule_uses_the_top_binade_and_cannot_overflow():
    for b in (1.0, 0.999, 1.001, 3e-5, 7.5, 65504.0, 2.0 ** -20, 1e30):
        e = exp_of(b)
        if abs(e) < 100:
            assert 2.0 ** 14 < b * 2.0 ** e <= 2.0 ** 15, (b, e)
    assert exp_of(1.0) == 15                              # the unit-bounded vectors (h1, h2, s_t, g_t): img_store's 32768
    assert exp_of(0.0) == 0


def test_split_residual_and_image_layout():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((64, 48)) * rng.choice([1e-4, 1e-2, 1.0], (64, 48))).astype(np.float32)
    e = exp_of(float(np.abs(x).max()))
    xs, hi, lo = split(x, e)
    assert np.isfinite(hi.astype(np.float32)).all()
    resid = np.abs(xs - (hi.astype(np.float32) + lo.astype(np.float32)))
    assert (resid <= np.maximum(np.abs(xs) * 2.0 ** -22, 2.0 ** -25)).all()
    img = image(x, e)
    assert img.shape == x.shape and img.dtype == np.float32
    halves = img.view(np.float16).reshape(-1, 16)
    np.testing.assert_array_equal(halves[:, :8].reshape(-1), hi.reshape(-1))
    np.testing.assert_array_equal(halves[:, 8:].reshape(-1), lo.reshape(-1))


def test_k1000_product_is_at_least_as_close_to_fp64_as_the_fp32_chain_and_loose_bounds_cost_nothing():
    rng = np.random.default_rng(1)
    a = rng.uniform(-0.5, 0.5, (24, 1008)).astype(np.float32)
    w = rng.uniform(-0.5, 0.5, (40, 1008)).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    chain = np.zeros((24, 40), np.float32)
    for k in range(a.shape[1]):                           # the k-ordered fma chain of the exact fp32 flavour
        chain = (chain.astype(np.float64) + np.outer(a[:, k].astype(np.float64), w[:, k].astype(np.float64))).astype(np.float32)
    rms = lambda c: float(np.sqrt(((c.astype(np.float64) - ref) ** 2).mean()))
    ea, ew = exp_of(0.5), exp_of(0.5)
    tight = gemm_h2(a, w, ea, ew)
    loose = gemm_h2(a, w, ea - 12, ew - 12)               # bounds 4096 x too large: almost every lo is an fp16 subnormal
    assert rms(tight) <= rms(chain)
    assert rms(loose) <= 1.05 * rms(tight)


def test_a_power_of_two_on_an_operand_comes_out_exactly():
    rng = np.random.default_rng(2)
    a = rng.standard_normal((8, 64)).astype(np.float32) * np.float32(1e-3)
    w = rng.standard_normal((16, 64)).astype(np.float32)
    ew = exp_of(float(np.abs(w).max()))
    base = gemm_h2(a, w, exp_of(float(np.abs(a).max())), ew)
    for k in (-40, 30):
        ak = a * np.float32(2.0 ** k)
        np.testing.assert_array_equal(gemm_h2(ak, w, exp_of(float(np.abs(ak).max())), ew), base * np.float32(2.0 ** k))
