"""fp64 numpy model of the ccoherent/cdsp hot path -- TEST INFRASTRUCTURE ONLY.

Independent of oracle/coherent_oracle.c (different language, different FFT: numpy's pocketfft
in complex128), used to bound the fp32 paths' error: lags exact, phase <= 1e-5 rad.
PARITY UNPINNED by the reference (no tests / fixtures; VOLK + FFTW absent) -- see
coherent_oracle.h.  Reference lines followed are cited per step.
"""
from __future__ import annotations

import numpy as np

FAITHFUL, DIGITAL = 0, 1


def to_complex(row_i8: np.ndarray) -> np.ndarray:
    """cdsp::convtofloat src/cdsp.cc:41-44 in fp64: (I + jQ) / 127."""
    x = row_i8.astype(np.float64) / 127.0
    return x[..., 0::2] + 1j * x[..., 1::2]


def xcorr_lag(sig: np.ndarray, ref: np.ndarray):
    """ccoherent::computelag src/ccoherent.cc:154-239 for one row in fp64.

    sig in [0,L), zeros [L,2L) (src/crtlsdr.cc:205-207); ref in [L,2L) (:215-218).
    Returns (lag, mag, frac, m) with m = |ifft|^2 (unnormalised backward transform).
    """
    L = sig.shape[-1]
    B = 2 * L
    a = np.zeros(B, dtype=np.complex128)
    b = np.zeros(B, dtype=np.complex128)
    a[:L] = sig
    b[L:] = ref
    c = np.fft.ifft(np.fft.fft(a) * np.conj(np.fft.fft(b))) * B   # FFTW backward is unnormalised
    m = c.real ** 2 + c.imag ** 2
    idx = int(np.argmax(m))                                        # first maximum
    mag = float(np.sqrt(m[idx] / L))                               # :204
    lag = idx - L                                                  # :232
    frac = 0.0
    if 0 < idx < B - 1:
        den = m[idx - 1] - 2.0 * m[idx] + m[idx + 1]
        if den != 0.0:
            frac = 0.5 * (m[idx - 1] - m[idx + 1]) / den
    return lag, mag, frac, m


class Model:
    """Stateful per-block model (EMA phasor + last lag per row), mirroring orc_engine."""

    def __init__(self, nrows: int, B: int, mode: int = FAITHFUL):
        self.nrows, self.B, self.L, self.mode = nrows, B, B // 2, mode
        self.p = np.ones(nrows, dtype=np.complex128)      # src/csdrdevice.cc:39-40
        self.lag = np.zeros(nrows, dtype=np.int64)
        self.mag = np.zeros(nrows)
        self.frac = np.zeros(nrows)
        self.raw = np.ones(nrows, dtype=np.complex128)    # un-averaged unit phasor of the last block
        self.frac_apply, self.frac_gain, self.frac_override = False, 1.0, None

    def set_frac_apply(self, enable=True, gain=1.0, frac_override=None):
        """Fractional-delay correction of the matrix rows (the build's crsdr_plan_set_frac_apply), DIGITAL mode."""
        self.frac_apply, self.frac_gain = bool(enable), float(gain)
        self.frac_override = None if frac_override is None else np.asarray(frac_override, dtype=np.float64)

    def block(self, rows: np.ndarray, lag_mask=None, refnoise_enabled: bool = True):
        L = self.L
        ref = to_complex(rows[0])
        matrix = np.empty_like(rows)
        matrix[0] = rows[0]                                # src/cpacketizer.cc:151
        for r in range(1, self.nrows):
            s = to_complex(rows[r])
            if lag_mask is None or lag_mask[r]:
                self.lag[r], self.mag[r], self.frac[r], _ = xcorr_lag(s, ref)
            y = s
            if self.mode == DIGITAL:
                d = int(self.lag[r])
                y = np.zeros(L, dtype=np.complex128)
                lo, hi = max(0, -d), min(L, L - d)
                if hi > lo:
                    y[lo:hi] = s[lo + d: hi + d]
            if refnoise_enabled:
                corr = np.sum(y * np.conj(ref))            # src/csdrdevice.cc:62
                a = abs(corr)
                if a != 0.0:
                    self.raw[r] = np.conj(corr) / a        # :63
                    self.p[r] = 0.5 * self.raw[r] + 0.5 * self.p[r]   # :66-67
            if self.frac_apply and self.mode == DIGITAL:
                # the zero-padded row advanced by lag + D samples: linear phase ramp over the SIGNED bin index
                D = self.frac_override[r] if self.frac_override is not None else self.frac_gain * self.frac[r]
                a = np.zeros(2 * L, dtype=np.complex128)
                a[:L] = s
                fs = np.fft.fftfreq(2 * L, d=1.0 / (2 * L))                   # 0 .. L-1, -L .. -1
                y = np.fft.ifft(np.fft.fft(a) * np.exp(2j * np.pi * fs * (float(self.lag[r]) + D) / (2 * L)))[:L]
            z = y * self.p[r] * 127.0                      # :80-84 then src/cdsp.cc:51-54
            q = np.empty(2 * L)
            q[0::2], q[1::2] = z.real, z.imag
            matrix[r] = np.clip(np.rint(q), -128, 127).astype(np.int8)
        phasor = self.p.copy()
        phasor[0] = 0.0
        return (self.lag.copy(), self.mag.copy(), self.frac.copy(), phasor, matrix)
