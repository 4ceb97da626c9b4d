"""CPU: the beamformer restatement (oracle/beamformer_oracle.c) against an independent fp64 numpy model.
The reference holds no fixtures for this client (parity unpinned, see the oracle's header)."""
import numpy as np
import pytest

import ura


def _projector(vec, k):
    un = vec[:, k:].astype(np.complex128)
    return un @ un.conj().T


@pytest.mark.parametrize("sources", [[(1.0, 1.3, 1.0)], [(0.6, 2.0, 1.0), (2.2, 1.0, 0.7)]])
def test_oracle_music_chain_matches_fp64_model(oracle, sources):
    rng = np.random.default_rng(11)
    L, k = 2048, len(sources)
    rows = ura.scene(L, sources, rng)
    R, U, s, pm = ura.music_fp64(rows, k, ncx=40, ncy=40)
    rxx = oracle.covariance(rows)
    assert np.abs(rxx - R).max() <= 1e-6 * np.abs(R).max()
    vec, sv = oracle.noisesubspace(rxx)
    assert np.allclose(sv, s, rtol=1e-5)
    assert np.abs(vec.conj().T @ vec - np.eye(ura.MX * ura.MY)).max() < 1e-6          # orthonormal
    assert np.abs(_projector(vec, k) - _projector(U, k)).max() < 1e-5                  # same noise subspace
    got = oracle.pmusic2d(vec, k, ura.D, ura.MX, ura.MY, 40, 40)
    assert np.allclose(got, pm, rtol=2e-3)
    # the scan peaks at the grid point nearest the strongest source -- mirrored: the reference forms
    # Rxx = X^H X with X(sample, channel) (heatmap2d2.cpp:197), whose principal vector is conj(a(alpha, beta)) =
    # a(pi - alpha, pi - beta), so a source at (alpha, beta) shows up at (pi - alpha, pi - beta)
    cx, cy = np.unravel_index(np.argmax(got), got.shape)
    mirrored = [((np.pi - a) * 40 / np.pi, (np.pi - b) * 40 / np.pi) for a, b, _ in sources]
    assert any(abs(cx - mx_) <= 1 and abs(cy - my_) <= 1 for mx_, my_ in mirrored)
    for mx_, my_ in mirrored:                                  # every source stands far above the floor
        assert got[int(round(mx_)), int(round(my_))] > 30 * np.median(got)


def test_oracle_subspace_of_rank_deficient_and_small_matrices(oracle):
    a = ura.steering(0.9, 1.1)
    R = np.outer(a, a.conj()).astype(np.complex64)            # rank one: 20-dimensional null space
    vec, sv = oracle.noisesubspace(R)
    assert abs(sv[0] - 21.0) < 1e-4 and np.all(sv[1:] < 1e-5)
    assert np.abs(vec.conj().T @ vec - np.eye(21)).max() < 1e-6
    assert np.abs(vec[:, 1:].conj().T @ a).max() < 1e-5       # noise subspace orthogonal to the source
    R2 = np.array([[2, 1j], [-1j, 2]], dtype=np.complex64)
    vec, sv = oracle.noisesubspace(R2)
    assert np.allclose(sv, [3, 1], atol=1e-6)


def test_oracle_music_chain_matches_the_committed_fixture(oracle, golden_dir):
    # tests/golden/music_ura21.npz: two sources on the 7 x 3 URA, fp64 numpy chain (tests/golden/make_golden.py)
    import os
    g = np.load(os.path.join(golden_dir, "music_ura21.npz"))
    k = int(g["k"])
    rxx = oracle.covariance(g["rows"])
    assert np.abs(rxx - g["rxx"]).max() <= 1e-6 * np.abs(g["rxx"]).max()
    vec, sv = oracle.noisesubspace(rxx)
    assert np.allclose(sv, g["sv"], rtol=1e-5, atol=1e-7)
    assert np.abs(_projector(vec, k) - g["projector"]).max() < 1e-5
    pm = oracle.pmusic2d(vec, k, float(g["d"]), int(g["mx"]), int(g["my"]), 40, 40)
    assert np.allclose(pm, g["pm"], rtol=5e-3)
