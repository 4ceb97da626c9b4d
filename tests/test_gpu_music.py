"""GPU parity for the beamformer rows: crsdr_covariance -> crsdr_noisesubspace -> crsdr_pmusic2d
through the C ABI against oracle/beamformer_oracle.c and the fp64 numpy model.

Bars: covariance relative 1e-6 of the largest entry (exact integer sums, one fp32 rounding); singular values
relative 1e-5; noise-subspace PROJECTOR 1e-5 absolute (a basis of a degenerate subspace is not unique);
pseudo-spectrum relative 2e-3 against the fp32 oracle fed the same subspace, 1e-2 against the fp64 chain."""
import importlib

import numpy as np
import pytest

import ura

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def b():
    binding = importlib.import_module("coherent-rtlsdr_amd.binding")
    if binding.device_count() < 1:
        pytest.fail("no HIP device: the product path has no CPU fallback")
    return binding


def _projector(vec, k):
    un = vec[:, k:].astype(np.complex128)
    return un @ un.conj().T


@pytest.mark.parametrize("sources", [[(1.0, 1.3, 1.0)], [(0.6, 2.0, 1.0), (2.2, 1.0, 0.7)]])
def test_music_chain_vs_oracle_and_fp64(b, oracle, sources):
    rng = np.random.default_rng(11)
    L, k = 8192, len(sources)
    rows = ura.scene(L, sources, rng)
    rxx = b.covariance(rows)
    assert np.abs(rxx - oracle.covariance(rows)).max() <= 2e-6 * np.abs(rxx).max()
    vec, sv = b.noisesubspace(rxx)
    ovec, osv = oracle.noisesubspace(rxx)
    assert np.allclose(sv, osv, rtol=1e-5)
    assert np.abs(vec.conj().T @ vec - np.eye(21)).max() < 1e-6
    assert np.abs(_projector(vec, k) - _projector(ovec, k)).max() < 1e-5
    pm = b.pmusic2d(vec, k, ura.D, ura.MX, ura.MY, 100, 100)
    assert np.allclose(pm, oracle.pmusic2d(vec, k, ura.D, ura.MX, ura.MY, 100, 100), rtol=2e-3)
    _, _, s64, pm64 = ura.music_fp64(rows, k)
    assert np.allclose(sv, s64, rtol=1e-5)
    assert np.allclose(pm, pm64, rtol=1e-2)


@pytest.mark.parametrize("m", [2, 3, 8, 21, 33, 64])
def test_subspace_sizes_and_degenerate_spectra(b, oracle, m):
    rng = np.random.default_rng(m)
    g = rng.standard_normal((m, 3 * m)) + 1j * rng.standard_normal((m, 3 * m))
    R = (g @ g.conj().T / (3 * m)).astype(np.complex64)
    vec, sv = b.noisesubspace(R)
    w = np.linalg.eigvalsh(R.astype(np.complex128))[::-1]
    assert np.allclose(sv, w, rtol=1e-5, atol=1e-6 * w[0])
    assert np.abs(vec.conj().T @ vec - np.eye(m)).max() < 1e-6
    # eigen-equation: R v_r = sv_r v_r
    assert np.abs(R.astype(np.complex128) @ vec - vec * sv[None, :]).max() < 1e-5 * w[0]
    # rank one: an (m-1)-dimensional null space, still an orthonormal basis orthogonal to the source
    a = np.exp(2j * np.pi * rng.random(m))
    vec, sv = b.noisesubspace(np.outer(a, a.conj()).astype(np.complex64))
    assert abs(sv[0] - m) < 1e-4 * m and np.all(sv[1:] < 1e-5 * m)
    assert np.abs(vec.conj().T @ vec - np.eye(m)).max() < 1e-6
    assert np.abs(vec[:, 1:].conj().T @ a).max() < 1e-5 * np.sqrt(m)
    # identity: nothing to rotate, converges in the first sweep
    vec, sv = b.noisesubspace(np.eye(m, dtype=np.complex64))
    assert np.allclose(sv, 1) and np.abs(vec.conj().T @ vec - np.eye(m)).max() < 1e-6


def test_music_argument_checks(b):
    eye = np.eye(21, dtype=np.complex64)
    for bad in (lambda: b.noisesubspace(np.eye(65, dtype=np.complex64)),      # beyond the LDS-resident size
                lambda: b.noisesubspace(np.eye(1, dtype=np.complex64)),
                lambda: b.pmusic2d(eye, 1, ura.D, 7, 2),                          # mx*my != m
                lambda: b.pmusic2d(eye, 0, ura.D, 7, 3),                          # no signal subspace
                lambda: b.pmusic2d(eye, 21, ura.D, 7, 3),                         # no noise subspace
                lambda: b.pmusic2d(eye, 1, ura.D, 7, 3, 0, 10)):
        with pytest.raises(b.CrsdrError) as e:
            bad()
        assert e.value.code == -1


def test_calibrate_freeze_then_locate_a_source(b):
    # the system end to end on the array of the reference's beamformer (7 x 3 URA, heatmap2d2.cpp:41-42):
    # (1) reference noise on: the plan learns each receiver's phase (src/ccoherent.cc:271-275);
    # (2) noise off: the estimate is frozen and still applied (src/csdrdevice.cc:80-84), the array looks at a
    #     far-field source; (3) covariance -> noise subspace -> 2-D MUSIC scan of the published matrix.
    # Without the calibration the receiver phases scramble the steering vector and the peak is lost.
    rng = np.random.default_rng(3)
    L, M = 8192, 21
    phi = rng.uniform(-np.pi, np.pi, M)
    alpha, beta = 1.1, 1.9
    plan = b.Plan(M + 1, 2 * L, b.MODE_FAITHFUL)
    for t in range(16):
        plan.block(ura.calibration_block(L, phi, rng), seq=t)
    rows = ura.scene(L, [(alpha, beta, 1.0)], rng, phi=phi)
    out = plan.block(rows, seq=16, flags=0)
    vec, sv = b.noisesubspace(b.covariance(out["matrix"]))
    assert sv[0] > 10 * sv[1]
    pm = b.pmusic2d(vec, 1, ura.D, ura.MX, ura.MY)
    cx, cy = np.unravel_index(np.argmax(pm), pm.shape)
    # mirrored peak: Rxx = X^H X of heatmap2d2.cpp:197 has conj(a) as its principal vector (tests/test_oracle_music.py)
    assert abs(cx - (np.pi - alpha) * 100 / np.pi) <= 1.5 and abs(cy - (np.pi - beta) * 100 / np.pi) <= 1.5
    assert pm.max() > 100 * np.median(pm)
    # control: the uncalibrated matrix does not focus there
    vec0, _ = b.noisesubspace(b.covariance(rows))
    pm0 = b.pmusic2d(vec0, 1, ura.D, ura.MX, ura.MY)
    assert pm0[cx, cy] < 0.01 * pm[cx, cy]
    plan.close()


def test_music_chain_matches_the_committed_fixture(b, golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "music_ura21.npz"))
    k = int(g["k"])
    rxx = b.covariance(g["rows"])
    assert np.abs(rxx - g["rxx"]).max() <= 2e-6 * np.abs(g["rxx"]).max()
    vec, sv = b.noisesubspace(rxx)
    assert np.allclose(sv, g["sv"], rtol=1e-5, atol=1e-7)
    assert np.abs(_projector(vec, k) - g["projector"]).max() < 1e-5
    pm = b.pmusic2d(vec, k, float(g["d"]), int(g["mx"]), int(g["my"]), 40, 40)
    assert np.allclose(pm, g["pm"], rtol=1e-2)


def _rxx_reference(rows):
    """beamformclient/heatmap2d2.cpp:189-199 in exact integer sums (int8 products summed in float64: |sum| <= 2^28) and the fp64 epilogue
    crsdr_covariance uses: (1/L) sum conj(x_a) x_b - conj(mean_a) mean_b with x = (I + jQ) / 127, rounded once to float."""
    x = rows[1:].astype(np.float64)
    I, Q = x[:, 0::2], x[:, 1::2]
    L = I.shape[1]
    g1 = I @ I.T + Q @ Q.T
    g3, g2 = I @ Q.T, Q @ I.T
    si, sq = I.sum(axis=1), Q.sum(axis=1)
    scale = 1.0 / (127.0 * 127.0)
    re = (g1 / L - (np.outer(si, si) + np.outer(sq, sq)) / (L * L)) * scale
    im = ((g3 - g2) / L - (np.outer(si, sq) - np.outer(sq, si)) / (L * L)) * scale
    return (re + 1j * im).astype(np.complex64)


@pytest.mark.parametrize("nsig,B", [(200, 2048), (64, 128), (1024, 16384), (129, 1024)])
def test_covariance_lds_tiled_kernel_is_exact(b, nsig, B):
    # r03: from 64 channels on (and blocksize % 512 == 0; B = 128: the one-tile-per-wave kernel) crsdr_covariance runs 128 x 128 output tiles through LDS with the K range split
    # over the grid, exact int32 partial sums added by a second kernel.  Integer arithmetic: the result must equal the reference
    # expression evaluated on exact sums, to the one float rounding of the output -- ragged channel counts (200, 129: padded tile rows),
    # one chunk per block (64 x 128), the benchmark's shape.
    rng = np.random.default_rng(nsig * 7 + B)
    rows = rng.integers(-128, 128, size=(nsig + 1, B), dtype=np.int8)
    rows[3] = -128                                                   # full-scale DC row: the largest sums int32 has to hold
    rxx = b.covariance(rows)
    ref = _rxx_reference(rows)
    assert rxx.shape == ref.shape
    assert np.abs(rxx - ref).max() <= 2e-7 * np.abs(ref).max()
    assert np.array_equal(rxx, rxx.conj().T)                         # Hermitian to the bit: the lower triangle is written as the mirror of the upper
    # the matrix where a packet holds it -- 16 + 4 N bytes behind a 256-byte aligned base, i.e. 4-byte aligned only -- on the device:
    # the same kernels (16-byte loads from dword-aligned addresses), the same bits
    import torch
    dev = torch.device("cuda", 0)
    off = 16 + 4 * (nsig + 1)
    buf = torch.zeros(off + rows.size + 256, dtype=torch.int8, device=dev)
    buf[off: off + rows.size].copy_(torch.from_numpy(rows.reshape(-1)))
    out = torch.zeros((nsig, nsig, 2), dtype=torch.float32, device=dev)
    assert (buf.data_ptr() + off) % 16 != 0 and (buf.data_ptr() + off) % 4 == 0
    b.covariance_device(out.data_ptr(), buf.data_ptr() + off, nsig + 1, B)
    got = out.cpu().numpy().view(np.complex64).reshape(nsig, nsig)
    assert np.array_equal(got, rxx)


def test_covariance_of_long_full_scale_rows_does_not_overflow(b):
    # rows longer than 65536 bytes: a full-scale DC row sums to more than int32 holds (2^18 bytes: 2^32), so the K range is split until
    # every workgroup's partial is exact and the partials are added in 64 bits -- for any channel count (4 here: one padded tile)
    nsig, B = 4, 1 << 18
    rng = np.random.default_rng(5)
    rows = rng.integers(-128, 128, size=(nsig + 1, B), dtype=np.int8)
    rows[1] = -128
    rows[2, 0::2] = 127; rows[2, 1::2] = -128
    rxx = b.covariance(rows)
    ref = _rxx_reference(rows)
    assert np.abs(rxx - ref).max() <= 2e-7 * np.abs(ref).max()
    with pytest.raises(b.CrsdrError):
        b.covariance(np.zeros((3, 65536 + 32), dtype=np.int8))          # a long row that the K split cannot cut into 512-byte pairs
