"""Seeded synthetic block source (stands in for the RTL-SDR dongles + reference-noise source).

The reference reads int8 IQ blocks from librtlsdr ring buffers (src/crtlsdr.cc:54,61-68,
173-203); no hardware exists here, so this generator produces the per-block input matrix
`rows[nrows][B]` (row 0 = reference noise, rows 1.. = delayed / rotated / scaled copies plus
independent noise) described in SURVEY.md section 8(d).

Everything is integer or correctly-rounded IEEE double arithmetic (SplitMix64 counters, a
4x16-bit Irwin-Hall gaussian, rotations from integer pairs via sqrt/divide only), so a C
implementation of the same recipe (host/csynth.c) reproduces the bytes exactly.
"""
from __future__ import annotations

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
# 1 / sqrt(4 * (65536^2 - 1) / 12): unit-variance scale of the sum of four uniform u16
INV_SIGMA16 = 1.0 / 37837.22723004292
SIGMA_REF = 30.0     # LSB per component
SIGMA_NOISE = 10.0   # LSB per component


def splitmix64(seed: int, idx: np.ndarray) -> np.ndarray:
    """Counter-based SplitMix64: output for counter values idx (uint64 array)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + (idx.astype(np.uint64) + np.uint64(1)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def _mix(seed: int, a: int, b: int) -> int:
    """Derive a stream seed from (seed, a, b) with two SplitMix64 rounds."""
    s = int(splitmix64(seed, np.array([a], dtype=np.uint64))[0])
    return int(splitmix64(s, np.array([b], dtype=np.uint64))[0])


def _splitmix64_vec(seeds: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """SplitMix64 with one seed per row: seeds [K] uint64, idx [M] -> [K, M]."""
    with np.errstate(over="ignore"):
        z = seeds.astype(np.uint64)[:, None] + (idx.astype(np.uint64)[None, :] + np.uint64(1)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def _mix_vec(seed: int, a: np.ndarray, b: int) -> np.ndarray:
    """_mix for an array of `a` values -> [K] uint64 stream seeds."""
    s1 = splitmix64(seed, np.asarray(a, dtype=np.uint64))
    return _splitmix64_vec(s1, np.array([b], dtype=np.uint64))[:, 0]


def gauss16_rows(seeds: np.ndarray, count: int) -> np.ndarray:
    """[K, count] approximately N(0,1) doubles, row k from stream seeds[k], counters 0..count-1."""
    z = _splitmix64_vec(seeds, np.arange(count, dtype=np.uint64))
    m = np.uint64(0xFFFF)
    acc = ((z & m) + ((z >> np.uint64(16)) & m) + ((z >> np.uint64(32)) & m) + (z >> np.uint64(48))).astype(np.int64)
    return (acc - 131070).astype(np.float64) * INV_SIGMA16


def gauss16(seed: int, start: int, count: int) -> np.ndarray:
    """`count` approximately N(0,1) doubles from counters start..start+count-1."""
    z = splitmix64(seed, np.arange(start, start + count, dtype=np.uint64))
    m = np.uint64(0xFFFF)
    acc = ((z & m) + ((z >> np.uint64(16)) & m) + ((z >> np.uint64(32)) & m) + (z >> np.uint64(48))).astype(np.int64)
    return (acc - 131070).astype(np.float64) * INV_SIGMA16


def cgauss(seed: int, start: int, count: int) -> np.ndarray:
    """`count` complex samples (re = even counter, im = odd counter), unit variance per component."""
    g = gauss16(seed, 2 * start, 2 * count)
    return g[0::2] + 1j * g[1::2]


def _quant_i8(x: np.ndarray) -> np.ndarray:
    """round-half-even then saturate to int8; complex in -> interleaved I,Q out."""
    out = np.empty(x.shape[:-1] + (2 * x.shape[-1],), dtype=np.int8)
    out[..., 0::2] = np.clip(np.rint(x.real), -128, 127).astype(np.int8)
    out[..., 1::2] = np.clip(np.rint(x.imag), -128, 127).astype(np.int8)
    return out


class RowParams:
    """Per-signal-row channel: integer delay d, rotation (c + js), gain g (fixed over blocks)."""

    def __init__(self, nsig: int, L: int, seed: int, dmax: int | None = None, locked: bool = False):
        self.nsig, self.L, self.seed = nsig, L, seed
        self.dmax = (L // 4) if dmax is None else int(dmax)
        z = splitmix64(_mix(seed, 0xD1, 0), np.arange(3 * nsig, dtype=np.uint64))
        zd, zp, zg = z[0::3], z[1::3], z[2::3]
        span = np.uint64(2 * self.dmax + 1)
        d = (zd % span).astype(np.int64) - self.dmax
        if nsig >= 4:                       # forced cases 0, +1, -1 once there is room
            for i, v in enumerate((0, 1, -1)):
                if abs(v) <= self.dmax:
                    d[i] = v
        if locked:
            d[:] = 0
        self.d = d.astype(np.int64)
        a = ((zp & np.uint64(0xFFFF)).astype(np.int64) - 32768).astype(np.float64)
        b = (((zp >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64) - 32768).astype(np.float64)
        both0 = (a == 0) & (b == 0)
        a[both0] = 1.0
        hyp = np.sqrt(a * a + b * b)
        self.c, self.s = a / hyp, b / hyp
        self.phi = np.arctan2(self.s, self.c)
        self.g = 0.5 + 0.5 * ((zg & np.uint64(0xFFFF)).astype(np.float64) / 65536.0)


def make_block(nsig: int, L: int, seed: int, block: int = 0, params: RowParams | None = None,
               offset_binary: bool = False, dmax: int | None = None, locked: bool = False,
               noise_sigma: float = SIGMA_NOISE) -> tuple[np.ndarray, RowParams]:
    """One input block: int8 (or offset-binary uint8) array [1+nsig][2L], plus the row params.

    row 0: round(SIGMA_REF * r[n]);  row k: round(g_k * SIGMA_REF * r[n - d_k] * (c_k + j s_k)
    + noise_sigma * w_k[n]).
    """
    p = params if params is not None else RowParams(nsig, L, seed, dmax=dmax, locked=locked)
    dm = p.dmax
    r_ext = cgauss(_mix(seed, 0xA0, block), 0, L + 2 * dm)        # r[n] = r_ext[n + dm]
    rows = np.empty((1 + nsig, 2 * L), dtype=np.int8)
    rows[0] = _quant_i8(SIGMA_REF * r_ext[dm: dm + L])
    n = np.arange(L)
    # rows are independent given r_ext: vectorised over chunks of rows (same arithmetic, same order of
    # operations per element as the scalar recipe in host/csynth.c)
    chunk = max(1, min(nsig, (1 << 18) // max(L, 1)))
    for k0 in range(0, nsig, chunk):
        ks = np.arange(k0, min(nsig, k0 + chunk))
        idx = n[None, :] - p.d[ks][:, None] + dm
        rk_re, rk_im = r_ext.real[idx], r_ext.imag[idx]
        c, sn, g = p.c[ks][:, None], p.s[ks][:, None], p.g[ks][:, None]
        rot_re = rk_re * c - rk_im * sn
        rot_im = rk_re * sn + rk_im * c
        seeds = _mix_vec(seed, 0xB000 + ks, block)
        wg = gauss16_rows(seeds, 2 * L)
        gs = g * SIGMA_REF
        xr = gs * rot_re + noise_sigma * wg[:, 0::2]
        xi = gs * rot_im + noise_sigma * wg[:, 1::2]
        out = rows[1 + k0: 1 + k0 + ks.size]
        out[:, 0::2] = np.clip(np.rint(xr), -128, 127).astype(np.int8)
        out[:, 1::2] = np.clip(np.rint(xi), -128, 127).astype(np.int8)
    if offset_binary:
        return (rows.view(np.uint8) ^ np.uint8(0x80)), p
    return rows, p


def config_seed(cfg: int) -> int:
    """seed = 0xC0FFEE + cfg."""
    return 0xC0FFEE + int(cfg)
