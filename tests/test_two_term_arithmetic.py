"""CPU model of the arithmetic the 32-channel (and wide 64-channel) conv kernels run on the matrix pipe (ar-vae_amd/csrc/
conv32_common.h): every operand tensor is multiplied by the power of two that brings its largest magnitude into [2^14, 2^15) and
split into two fp16 terms h = fp16(s x), l = fp16(s x - h); a product is the three partial products l h', h l', h h' accumulated
in fp32, and the result is multiplied by the two inverse scales.  numpy restatement (float16 rounding is IEEE round-to-nearest-even,
as v_cvt_pk_f16_f32): pins the error bounds DESIGN.md section 4 item 23 states, without a GPU."""
import numpy as np


def pow2_for(amax):
    """the kernels' pow2_for(): scale and inverse from the biased exponent of max |x| (clamped at 16)"""
    bits = np.float32(amax).view(np.uint32)
    e = max(int((bits >> 23) & 0xff), 16)
    return np.float32(2.0) ** (141 - e), np.float32(2.0) ** (e - 141)


def split(x):
    s, inv = pow2_for(np.abs(x).max())
    y = (x * s).astype(np.float32)                       # exact: a power of two
    h = y.astype(np.float16)
    l = (y - h.astype(np.float32)).astype(np.float16)    # the subtraction is exact in fp32
    return h, l, s, inv


def matmul_two_term(a, b):
    """[M, K] x [K, N] as the kernels compute it; the three partial products are exact in fp32 (11 x 11 bits), the accumulation is
    modelled in float64 and rounded once (the kernels accumulate in fp32: their extra error is the fp32 summation's own)"""
    ah, al, sa, ia = split(a)
    bh, bl, sb, ib = split(b)
    f = np.float64
    acc = al.astype(f) @ bh.astype(f) + ah.astype(f) @ bl.astype(f) + ah.astype(f) @ bh.astype(f)
    return (acc * f(ia) * f(ib)).astype(np.float32)


def rel(x, ref):
    return np.linalg.norm(x.astype(np.float64) - ref) / np.linalg.norm(ref)


def test_operand_is_reproduced_to_22_bits_near_the_maximum_and_to_2_pow_minus_39_of_it_below():
    rs = np.random.RandomState(0)
    x = (rs.standard_normal(1 << 16) * np.exp(2.0 * rs.standard_normal(1 << 16))).astype(np.float32)
    h, l, s, inv = split(x)
    back = (h.astype(np.float64) + l.astype(np.float64)) * np.float64(inv)
    amax = np.abs(x).max()
    err = np.abs(back - x.astype(np.float64))
    assert 2.0 ** 14 <= amax * s < 2.0 ** 15
    assert np.all(err <= np.maximum(np.abs(x) * 2.0 ** -22, amax * 2.0 ** -39))
    near = np.abs(x) > amax * 2.0 ** -16
    assert near.sum() > 1000 and np.all(err[near] <= np.abs(x[near]) * 2.0 ** -22)


def test_dot_products_sit_at_fp32_rounding_noise():
    rs = np.random.RandomState(1)
    a = np.maximum(rs.standard_normal((256, 512)) * np.exp(rs.standard_normal((256, 512))), 0).astype(np.float32)   # ReLU-like, heavy tail
    b = (rs.standard_normal((512, 64)) * 0.05).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    e_two = rel(matmul_two_term(a, b), ref)
    acc = np.zeros((256, 64), np.float32)                # an fp32 multiply-add chain, as a CPU / vector-ALU kernel would run it
    for k in range(512):
        acc += a[:, k:k + 1] * b[k:k + 1, :]
    e_fp32 = rel(acc, ref)
    assert e_two < 1.5e-7, e_two
    assert e_two < e_fp32, (e_two, e_fp32)


def test_power_of_two_scaling_of_an_operand_changes_nothing_but_the_exponent():
    rs = np.random.RandomState(2)
    a = rs.standard_normal((64, 128)).astype(np.float32)
    b = rs.standard_normal((128, 32)).astype(np.float32)
    base = matmul_two_term(a, b)
    for ka, kb in ((-23, 0), (9, -30), (-12, 14)):
        got = matmul_two_term(a * np.float32(2.0 ** ka), b * np.float32(2.0 ** kb))
        np.testing.assert_array_equal(got, base * np.float32(2.0 ** (ka + kb)))


def test_all_zero_and_tiny_tensors_do_not_overflow_the_scale():
    z = np.zeros((8, 16), np.float32)
    b = np.ones((16, 4), np.float32)
    assert np.all(matmul_two_term(z, b) == 0)
    tiny = np.full((8, 16), 1e-38, np.float32)           # a denormal-range maximum: the exponent clamp keeps the scale finite
    s, inv = pow2_for(np.abs(tiny).max())
    assert np.isfinite(s) and np.isfinite(inv) and s * inv == 1.0
    out = matmul_two_term(tiny, b)
    assert np.all(np.isfinite(out))
