"""Oracle per-op tests (CPU): the C restatement of class cdsp against known answers and an
independent implementation (numpy / scipy pocketfft).  Reference semantics cited per test.
PARITY UNPINNED by the reference itself (no tests / golden vectors upstream)."""
import numpy as np
import pytest
import scipy.fft


def test_convtosigned_xor(oracle):
    # src/cdsp.cc:21-34: x ^ 0x80 turns offset-binary into two's complement
    u = np.arange(256, dtype=np.uint8)
    s = oracle.convtosigned(u)
    assert np.array_equal(s.astype(np.int16), u.astype(np.int16) - 128)


def test_convtofloat_reciprocal_multiply(oracle):
    # src/cdsp.cc:41-44 / volk_8i_s32f_convert_32f: multiply by the fp32 reciprocal, not divide
    i8 = np.arange(-128, 128, dtype=np.int8)
    out = oracle.convtofloat(i8).view(np.float32)
    expect = i8.astype(np.float32) * np.float32(1.0 / 127.0)
    assert np.array_equal(out, expect)
    assert out[0] == np.float32(-128) * np.float32(1.0 / 127.0) and out[-1] == np.float32(127) * np.float32(1.0 / 127.0)


def test_convto8bit_round_half_even_and_saturation(oracle):
    # src/cdsp.cc:51-54 / volk_32f_s32f_convert_8i: clamp to [-128,127] then rintf (half-even)
    vals = np.array([0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 126.5, 127.4, 127.6, 500.0, -128.5, -129.0, -1000.0, 0.0],
                    dtype=np.float32) / np.float32(127.0)
    x = (vals[0::2] + 1j * vals[1::2]).astype(np.complex64)
    out = oracle.convto8bit(x)
    r = x.view(np.float32) * np.float32(127.0)
    expect = np.clip(np.rint(r), -128, 127).astype(np.int8)
    assert np.array_equal(out, expect)
    assert out.min() == -128 and out.max() == 127


def test_scalarmul_conjmul_magsq_bitwise(oracle):
    rng = np.random.default_rng(1)
    a = (rng.standard_normal(1000) + 1j * rng.standard_normal(1000)).astype(np.complex64)
    b = (rng.standard_normal(1000) + 1j * rng.standard_normal(1000)).astype(np.complex64)
    s = np.complex64(0.3 - 0.8j)
    ar, ai, br, bi = a.real, a.imag, b.real, b.imag
    # one rounding per op, in the order of the VOLK generic kernels
    sm = oracle.scalarmul(a, s)
    assert np.array_equal(sm.real, ar * s.real - ai * s.imag) and np.array_equal(sm.imag, ar * s.imag + ai * s.real)
    cm = oracle.conjugatemul(a, b)          # in1 * conj(in2), src/cdsp.cc:105-108
    assert np.array_equal(cm.real, ar * br + ai * bi) and np.array_equal(cm.imag, ai * br - ar * bi)
    assert np.array_equal(oracle.magsquared(a), ar * ar + ai * ai)


def test_conj_dotproduct_matches_fp64(oracle):
    rng = np.random.default_rng(2)
    a = (rng.standard_normal(8192) + 1j * rng.standard_normal(8192)).astype(np.complex64)
    b = (rng.standard_normal(8192) + 1j * rng.standard_normal(8192)).astype(np.complex64)
    got = oracle.conj_dotproduct(a, b)
    ref = np.sum(a.astype(np.complex128) * np.conj(b.astype(np.complex128)))
    assert abs(got - ref) <= 2e-4 * np.sum(np.abs(a) * np.abs(b))


def test_indexofmax_first_strict_maximum(oracle):
    # volk_32f_index_max_32u generic keeps the FIRST maximum (strict >)
    m = np.zeros(100, dtype=np.float32)
    m[[17, 40, 99]] = 5.0
    assert oracle.indexofmax(m) == 17
    m[3] = 5.0
    assert oracle.indexofmax(m) == 3
    assert oracle.indexofmax(np.full(8, -1.0, dtype=np.float32)) == 0


@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64, 1024, 2048, 16384])
@pytest.mark.parametrize("sign", [-1, 1])
def test_fft_vs_pocketfft(oracle, n, sign):
    # FFTW definition (src/ccoherent.cc:87-93): unnormalised, sign -1 forward / +1 backward
    rng = np.random.default_rng(n + sign)
    x = (rng.standard_normal((3, n)) + 1j * rng.standard_normal((3, n))).astype(np.complex64)
    got = oracle.fft(x, sign)
    x64 = x.astype(np.complex128)
    ref = scipy.fft.fft(x64, axis=-1) if sign < 0 else scipy.fft.ifft(x64, axis=-1) * n
    rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    assert rel <= 1e-6, rel
    # independent fp32 implementation (pocketfft complex64)
    ref32 = scipy.fft.fft(x, axis=-1) if sign < 0 else scipy.fft.ifft(x, axis=-1) * np.float32(n)
    assert np.linalg.norm(got - ref32) / np.linalg.norm(ref) <= 2e-6


def test_fft_impulse_and_tone(oracle):
    n = 256
    x = np.zeros(n, dtype=np.complex64)
    x[1] = 1.0
    X = oracle.fft(x, -1)
    k = np.arange(n)
    assert np.allclose(X, np.exp(-2j * np.pi * k / n), atol=1e-6)
    tone = np.exp(2j * np.pi * 5 * k / n).astype(np.complex64)
    T = oracle.fft(tone, -1)
    assert np.argmax(np.abs(T)) == 5 and abs(T[5] - n) < 1e-3


def test_fft_rejects_bad_sizes(oracle):
    with pytest.raises(ValueError):
        oracle.fft(np.zeros(12, dtype=np.complex64), -1)
