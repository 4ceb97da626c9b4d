"""GPU parity, per-op level: every cdsp entry point of the C ABI (include/crsdr.h) against the
CPU oracle on the same inputs.  Bar: bit-exact for integer / byte / index results and for the
single-rounding fp32 ops; stated tolerance for reductions and the FFT."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def b():
    binding = importlib.import_module("coherent-rtlsdr_amd.binding")
    if binding.device_count() < 1:
        pytest.fail("no HIP device: the product path has no CPU fallback")
    return binding


def _crand(rng, *shape):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)


def test_convtosigned_bitexact(b, oracle):
    u = np.random.default_rng(0).integers(0, 256, 4096, dtype=np.uint8)
    assert np.array_equal(b.convtosigned(u), oracle.convtosigned(u))


def test_convtofloat_bitexact(b, oracle):
    i8 = np.tile(np.arange(-128, 128, dtype=np.int8), 8)
    assert np.array_equal(b.convtofloat(i8).view(np.float32), oracle.convtofloat(i8).view(np.float32))


def test_scalarmul_conjmul_magsq_bitexact(b, oracle):
    rng = np.random.default_rng(1)
    a, c = _crand(rng, 5000), _crand(rng, 5000)
    s = np.complex64(0.6 - 0.3j)
    assert np.array_equal(b.scalarmul(a, s).view(np.float32), oracle.scalarmul(a, s).view(np.float32))
    assert np.array_equal(b.conjugatemul(a, c).view(np.float32), oracle.conjugatemul(a, c).view(np.float32))
    assert np.array_equal(b.magsquared(a), oracle.magsquared(a))


def test_convto8bit_bitexact_including_ties_and_saturation(b, oracle):
    rng = np.random.default_rng(2)
    x = (_crand(rng, 4096) * np.float32(0.7)).astype(np.complex64)
    xf = x.view(np.float32)
    xf[:16] = np.array([0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 126.5, 127.5, 300, -128.5, -129, -300, 0, 63.5, -63.5, 1e9],
                       dtype=np.float32) / np.float32(127.0)
    assert np.array_equal(b.convto8bit(x), oracle.convto8bit(x))


def test_conj_dotproduct_tolerance(b, oracle):
    rng = np.random.default_rng(3)
    a, c = _crand(rng, 8192), _crand(rng, 8192)
    got, exp = b.conj_dotproduct(a, c), oracle.conj_dotproduct(a, c)
    ref = np.sum(a.astype(np.complex128) * np.conj(c.astype(np.complex128)))
    scale = np.sum(np.abs(a) * np.abs(c))
    assert abs(got - ref) <= 1e-6 * scale          # tree reduction in fp64 partials
    assert abs(got - exp) <= 2e-4 * scale          # VOLK-generic sequential fp32 accumulator


def test_indexofmax_first_maximum(b, oracle):
    rng = np.random.default_rng(4)
    m = rng.random(16384).astype(np.float32)
    assert b.indexofmax(m) == oracle.indexofmax(m)
    m[[100, 9000, 16383]] = 7.0                     # tie -> lowest index
    assert b.indexofmax(m) == 100 == oracle.indexofmax(m)
    m[:] = 1.0
    assert b.indexofmax(m) == 0


@pytest.mark.parametrize("n", [16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
@pytest.mark.parametrize("sign", [-1, 1])
def test_fft_vs_oracle_and_fp64(b, oracle, n, sign):
    # tolerance: relative L2 <= 1e-6 against fp64, and against the fp32 oracle
    rng = np.random.default_rng(n * 3 + sign)
    x = _crand(rng, 5, n)
    got = b.fft(x, sign)
    x64 = x.astype(np.complex128)
    ref = np.fft.fft(x64, axis=-1) if sign < 0 else np.fft.ifft(x64, axis=-1) * n
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= 1e-6
    exp = oracle.fft(x, sign)
    assert np.linalg.norm(got - exp) / np.linalg.norm(ref) <= 1e-6


def test_fft_linearity_and_roundtrip(b):
    # size-independent properties at the full block size: linearity and ifft(fft(x)) = N x
    n = 16384
    rng = np.random.default_rng(5)
    x, y = _crand(rng, n), _crand(rng, n)
    lhs = b.fft(x + np.complex64(2) * y)
    rhs = b.fft(x) + np.complex64(2) * b.fft(y)
    assert np.linalg.norm(lhs - rhs) / np.linalg.norm(rhs) <= 1e-6
    back = b.fft(b.fft(x, -1), +1) / np.float32(n)
    assert np.linalg.norm(back - x) / np.linalg.norm(x) <= 1e-6


def test_ragged_and_degenerate_sizes(b, oracle):
    # per-op entry points on sizes that are not multiples of anything convenient, and the error codes for
    # empty / invalid input (the reference's wrappers would pass such n straight to VOLK)
    rng = np.random.default_rng(7)
    for n in (1, 2, 3, 17, 63, 65, 1000, 4097):
        i8 = rng.integers(-128, 128, n, dtype=np.int8)
        if n % 2 == 0:
            assert np.array_equal(b.convtofloat(i8).view(np.float32), oracle.convtofloat(i8).view(np.float32))
        x = _crand(rng, n)
        assert np.array_equal(b.magsquared(x), oracle.magsquared(x))
        assert np.array_equal(b.convto8bit(x), oracle.convto8bit(x))
        assert np.array_equal(b.scalarmul(x, 1j).view(np.float32), oracle.scalarmul(x, 1j).view(np.float32))
        m = rng.random(n).astype(np.float32)
        assert b.indexofmax(m) == oracle.indexofmax(m)
    for bad in (lambda: b.convtofloat(np.zeros(0, dtype=np.int8)), lambda: b.magsquared(np.zeros(0, dtype=np.complex64)),
                lambda: b.convtosigned(np.zeros(12, dtype=np.uint8)),           # n % 8 != 0
                lambda: b.fft(np.zeros(32768, dtype=np.complex64)),             # beyond the LDS-resident per-op size
                lambda: b.fft(np.zeros(8, dtype=np.complex64))):                # below the smallest transform
        with pytest.raises(b.CrsdrError) as e:
            bad()
        assert e.value.code == -1


@pytest.mark.parametrize("nsig,L", [(3, 256), (21, 8192), (64, 1024), (100, 512), (257, 2048)])
def test_covariance_matches_the_beamformer_formula(b, synth, nsig, L):
    # Rxx = (1/L) X^H X with the per-channel mean removed, X = int8/127 over the signal channels
    # (beamformclient/heatmap2d2.cpp:189-199).  The GPU sums are exact integers (int8 MFMA), so the only
    # error is the final fp32 rounding: relative 1e-6 of the largest entry.
    rows, _ = synth.make_block(nsig, L, 1000 + nsig, 0, dmax=L // 8)
    rows = rows.copy()
    rows[1] = np.clip(rows[1].astype(np.int16) + 17, -128, 127).astype(np.int8)      # a channel with a DC offset
    got = b.covariance(rows)
    x = rows[1:].astype(np.float64) / 127.0
    X = (x[:, 0::2] + 1j * x[:, 1::2]).T                      # X(n, c)
    X = X - X.mean(axis=0, keepdims=True)
    ref = (X.conj().T @ X) / L
    assert got.shape == (nsig, nsig)
    assert np.abs(got - ref).max() <= 1e-6 * np.abs(ref).max()
    assert np.allclose(got, got.conj().T, atol=1e-6 * np.abs(ref).max())     # Hermitian


def test_covariance_of_the_aligned_matrix_is_rank_one_dominated(b, synth):
    # end to end: after digital alignment every channel is a scaled copy of the reference noise plus
    # independent noise, so Rxx has one dominant eigenvalue whose eigenvector has (nearly) equal phases --
    # the structure the MUSIC scan of the reference's beamformer relies on
    nsig, L = 21, 8192
    seed = synth.config_seed(2)
    params = synth.RowParams(nsig, L, seed)
    plan = b.Plan(nsig + 1, 2 * L, b.MODE_DIGITAL)
    for t in range(10):
        rows, _ = synth.make_block(nsig, L, seed, t, params=params)
        out = plan.block(rows, seq=t)
    R = b.covariance(out["matrix"]).astype(np.complex128)
    w, v = np.linalg.eigh(R)
    assert w[-1] > 5 * w[-2]
    ph = np.angle(v[:, -1] * np.conj(v[0, -1]))
    assert np.abs(ph).max() < 0.05
    plan.close()


def test_packed_complex_primitives_round_like_the_scalar_ones():
    # csrc/cpk.hpp: complex product / conjugate product / +-i additions as v_pk_* instructions with operand modifiers
    # must give the same bits as the scalar fma / mul / add formulations of fft_lds.hpp (tools/pk_check.hip, built
    # by __graft_entry__.build()); the K1 kernels rely on it for "packed == scalar" (tests/test_gpu_plan.py)
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "pk_check")
    if not os.path.exists(exe):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-o", exe, exe + ".hip"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "mismatches 0" in r.stdout, r.stdout + r.stderr
