"""Test helper: a synthetic uniform rectangular array scene for the beamformer rows.

Geometry and steering vector are those of beamformclient/heatmap2d2.cpp:103-115 (s_vecd2d): element
(ix, iy) at index iy*Mx + ix, phase 2 pi d (ix cos(alpha) sin(beta) + iy cos(beta)).  Plain numpy, no
reference code involved; the default array is the reference's 7 x 3 URA with d = 1.225*1.24/3 (:41-42,:198)."""
import numpy as np

MX, MY = 7, 3
D = np.float32((1.225 * 1.24) / 3.0)


def steering(alpha, beta, d=D, mx=MX, my=MY):
    ix = np.tile(np.arange(mx), my)
    iy = np.repeat(np.arange(my), mx)
    return np.exp(2j * np.pi * float(d) * (ix * np.cos(alpha) * np.sin(beta) + iy * np.cos(beta)))


def quantise(x):
    out = np.empty(x.shape[:-1] + (2 * x.shape[-1],), dtype=np.int8)
    out[..., 0::2] = np.clip(np.rint(x.real), -128, 127)
    out[..., 1::2] = np.clip(np.rint(x.imag), -128, 127)
    return out


def scene(L, sources, rng, phi=None, mx=MX, my=MY, d=D, sigma_s=25.0, sigma_n=8.0):
    """One block of the array looking at far-field sources [(alpha, beta, relative amplitude)...]:
    int8 rows [1 + M][2L]; row 0 (the reference-noise channel) carries only receiver noise.  phi = per-channel
    receiver phase offsets (what the calibration removes)."""
    M = mx * my
    x = sigma_n * (rng.standard_normal((M + 1, L)) + 1j * rng.standard_normal((M + 1, L))) / np.sqrt(2)
    for alpha, beta, amp in sources:
        s = sigma_s * amp * (rng.standard_normal(L) + 1j * rng.standard_normal(L)) / np.sqrt(2)
        x[1:] += steering(alpha, beta, d, mx, my)[:, None] * s[None, :]
    if phi is not None:
        x[1:] *= np.exp(1j * np.asarray(phi))[:, None]
    return quantise(x)


def calibration_block(L, phi, rng, sigma=30.0, sigma_n=5.0):
    """Reference noise switched on: every channel sees the same noise, rotated by its receiver phase."""
    M = len(phi)
    r = sigma * (rng.standard_normal(L) + 1j * rng.standard_normal(L)) / np.sqrt(2)
    x = np.empty((M + 1, L), dtype=np.complex128)
    x[0] = r
    x[1:] = r[None, :] * np.exp(1j * np.asarray(phi))[:, None]
    x[1:] += sigma_n * (rng.standard_normal((M, L)) + 1j * rng.standard_normal((M, L))) / np.sqrt(2)
    return quantise(x)


def music_fp64(matrix, k, d=D, mx=MX, my=MY, ncx=100, ncy=100):
    """fp64 numpy model of heatmap2d2.cpp:185-203: covariance -> SVD noise subspace -> (|a|^2/|Un^H a|^2)^2."""
    x = matrix[1:].astype(np.float64) / 127.0
    X = (x[:, 0::2] + 1j * x[:, 1::2]).T
    X = X - X.mean(axis=0, keepdims=True)
    R = X.conj().T @ X / X.shape[0]
    U, s, _ = np.linalg.svd(R)
    Un = U[:, k:]
    pm = np.empty((ncx, ncy))
    for cx in range(ncx):
        for cy in range(ncy):
            a = steering(cx * np.pi / ncx, cy * np.pi / ncy, d, mx, my)
            pm[cx, cy] = (np.vdot(a, a).real / np.linalg.norm(Un.conj().T @ a) ** 2) ** 2
    return R, U, s, pm
