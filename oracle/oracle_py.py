"""ctypes view of oracle/_build/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
(see coherent_oracle.h).  Never imported by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

FAITHFUL, DIGITAL = 0, 1


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("coherent_oracle.c", "beamformer_oracle.c", "coherent_oracle.h", "Makefile")]
    outs = (_SO, os.path.join(_HERE, "_build", "liboracle_refflags.so"))
    stale = force or any(not os.path.exists(o) for o in outs) or any(os.path.getmtime(s) > min(os.path.getmtime(o) for o in outs) for s in src)
    if stale:
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def _declare(L):
    f32p, i8p, u8p, u32p, i32p = (C.POINTER(C.c_float), C.POINTER(C.c_int8), C.POINTER(C.c_uint8),
                                  C.POINTER(C.c_uint32), C.POINTER(C.c_int32))
    L.orc_convtosigned.argtypes = [u8p, u8p, C.c_int]
    L.orc_convtofloat.argtypes = [f32p, i8p, C.c_int]
    L.orc_scalarmul.argtypes = [f32p, f32p, C.c_float, C.c_float, C.c_int]
    L.orc_convto8bit.argtypes = [i8p, f32p, C.c_int]
    L.orc_conj_dotproduct.argtypes = [f32p, f32p, f32p, C.c_int]
    L.orc_magsquared.argtypes = [f32p, f32p, C.c_int]
    L.orc_conjugatemul.argtypes = [f32p, f32p, f32p, C.c_int]
    L.orc_indexofmax.argtypes = [f32p, C.c_int]
    L.orc_indexofmax.restype = C.c_uint32
    L.orc_fft.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_int]
    L.orc_engine_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_engine_create.restype = C.c_void_p
    L.orc_engine_destroy.argtypes = [C.c_void_p]
    L.orc_engine_reset.argtypes = [C.c_void_p]
    L.orc_engine_block_mt.argtypes = [C.c_void_p, i8p, u32p, u8p, C.c_int, C.c_uint32,
                                      i32p, f32p, f32p, f32p, i8p, C.c_int]
    L.orc_engine_set_frac_apply.argtypes = [C.c_void_p, C.c_int, C.c_float, f32p]
    L.orc_packet_bytes.argtypes = [C.c_int, C.c_int]
    L.orc_packet_bytes.restype = C.c_size_t
    L.orc_packet_matrix_offset.argtypes = [C.c_int]
    L.orc_packet_matrix_offset.restype = C.c_size_t
    L.orc_covariance.argtypes = [f32p, i8p, C.c_int, C.c_int]
    L.orc_noisesubspace.argtypes = [f32p, f32p, f32p, C.c_int]
    L.orc_pmusic2d.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
    return L


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = _declare(C.CDLL(_SO))
    return _lib


_SO_REFFLAGS = os.path.join(_HERE, "_build", "liboracle_refflags.so")


def lib_refflags():
    """The same sources built with the reference's own compiler flags (-O3, no -march: CMakeLists.txt:10) -- bench.py's
    cpu_baseline times this one; the checker is the -mavx2 build above (identical results: neither contracts to FMA)."""
    if not os.path.exists(_SO_REFFLAGS):
        build(force=True)
    return _declare(C.CDLL(_SO_REFFLAGS))


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty)) if a is not None else None


def _c64_as_f32(a):
    a = np.ascontiguousarray(a, dtype=np.complex64)
    return a, a.view(np.float32)


# ---- per-op wrappers (numpy in / numpy out) -----------------------------------------------

def convtosigned(u8):
    u8 = np.ascontiguousarray(u8, dtype=np.uint8)
    out = np.empty_like(u8)
    lib().orc_convtosigned(_p(u8, C.c_uint8), _p(out, C.c_uint8), u8.size)
    return out.view(np.int8)


def convtofloat(i8):
    i8 = np.ascontiguousarray(i8, dtype=np.int8)
    out = np.empty(i8.size, dtype=np.float32)
    lib().orc_convtofloat(_p(out, C.c_float), _p(i8, C.c_int8), i8.size)
    return out.view(np.complex64)


def scalarmul(x, s):
    x, xf = _c64_as_f32(x)
    out = np.empty_like(x)
    s = np.complex64(s)
    lib().orc_scalarmul(_p(out.view(np.float32), C.c_float), _p(xf, C.c_float), float(s.real), float(s.imag), x.size)
    return out


def convto8bit(x):
    x, xf = _c64_as_f32(x)
    out = np.empty(2 * x.size, dtype=np.int8)
    lib().orc_convto8bit(_p(out, C.c_int8), _p(xf, C.c_float), x.size)
    return out


def conj_dotproduct(a, b):
    a, af = _c64_as_f32(a)
    b, bf = _c64_as_f32(b)
    res = np.zeros(2, dtype=np.float32)
    lib().orc_conj_dotproduct(_p(res, C.c_float), _p(af, C.c_float), _p(bf, C.c_float), a.size)
    return np.complex64(res[0] + 1j * res[1])


def magsquared(x):
    x, xf = _c64_as_f32(x)
    out = np.empty(x.size, dtype=np.float32)
    lib().orc_magsquared(_p(out, C.c_float), _p(xf, C.c_float), x.size)
    return out


def conjugatemul(a, b):
    a, af = _c64_as_f32(a)
    b, bf = _c64_as_f32(b)
    out = np.empty_like(a)
    lib().orc_conjugatemul(_p(out.view(np.float32), C.c_float), _p(af, C.c_float), _p(bf, C.c_float), a.size)
    return out


def indexofmax(m):
    m = np.ascontiguousarray(m, dtype=np.float32)
    return int(lib().orc_indexofmax(_p(m, C.c_float), m.size))


def fft(x, sign=-1):
    """Batched along the last axis; x [..., n] complex64."""
    x, xf = _c64_as_f32(x)
    n = x.shape[-1]
    howmany = x.size // n
    out = np.empty_like(x)
    rc = lib().orc_fft(_p(out.view(np.float32), C.c_float), _p(xf, C.c_float), n, sign, howmany)
    if rc:
        raise ValueError("orc_fft: bad arguments")
    return out


def covariance(matrix):
    """beamformclient/heatmap2d2.cpp:185-199 on an int8 [nrows][B] matrix -> complex64 [nrows-1][nrows-1]."""
    m = np.ascontiguousarray(matrix, dtype=np.int8)
    nrows, B = m.shape
    out = np.empty((nrows - 1, nrows - 1), dtype=np.complex64)
    lib().orc_covariance(_p(out, C.c_float), _p(m, C.c_int8), nrows, B)
    return out


def noisesubspace(rxx):
    """heatmap2d2.cpp:69-79: (vec, sv); vec[:, r] is the singular vector of sv[r] (descending)."""
    r = np.ascontiguousarray(rxx, dtype=np.complex64)
    M = r.shape[0]
    vec = np.empty((M, M), dtype=np.complex64)
    sv = np.empty(M, dtype=np.float32)
    rc = lib().orc_noisesubspace(_p(vec, C.c_float), _p(sv, C.c_float), _p(r, C.c_float), M)
    if rc < 0:
        raise RuntimeError("oracle Jacobi iteration did not converge")
    return vec, sv


def pmusic2d(vec, k, d, mx, my, ncx, ncy):
    """heatmap2d2.cpp:103-147 pseudo-spectrum [ncx][ncy] (not normalised)."""
    v = np.ascontiguousarray(vec, dtype=np.complex64)
    M = v.shape[0]
    pm = np.empty((ncx, ncy), dtype=np.float32)
    lib().orc_pmusic2d(_p(pm, C.c_float), _p(v, C.c_float), M, k, d, mx, my, ncx, ncy)
    return pm


class Engine:
    def __init__(self, nrows, B, mode=FAITHFUL, nfft_cap=0, cdll=None):
        self.nrows, self.B, self.mode = nrows, B, mode
        self._L = cdll if cdll is not None else lib()
        self._h = self._L.orc_engine_create(nrows, B, mode, nfft_cap)
        if not self._h:
            raise ValueError("orc_engine_create failed")
        self.packet_bytes = int(self._L.orc_packet_bytes(nrows, B))
        self.matrix_offset = int(self._L.orc_packet_matrix_offset(nrows))

    def reset(self):
        self._L.orc_engine_reset(self._h)

    def set_frac_apply(self, enable=True, gain=1.0, frac_override=None):
        ov = None if frac_override is None else np.ascontiguousarray(frac_override, dtype=np.float32)
        if self._L.orc_engine_set_frac_apply(self._h, int(bool(enable)), C.c_float(gain), _p(ov, C.c_float)):
            raise RuntimeError("orc_engine_set_frac_apply failed")

    def block(self, rows, readcnt=None, lag_mask=None, refnoise_enabled=True, seq=0, nthreads=1,
              want_packet=True):
        rows = np.ascontiguousarray(rows, dtype=np.int8)
        assert rows.shape == (self.nrows, self.B)
        lag = np.zeros(self.nrows, dtype=np.int32)
        mag = np.zeros(self.nrows, dtype=np.float32)
        frac = np.zeros(self.nrows, dtype=np.float32)
        ph = np.zeros(2 * self.nrows, dtype=np.float32)
        pkt = np.zeros(self.packet_bytes, dtype=np.int8) if want_packet else None
        rc_arr = None if readcnt is None else np.ascontiguousarray(readcnt, dtype=np.uint32)
        mk = None if lag_mask is None else np.ascontiguousarray(lag_mask, dtype=np.uint8)
        rc = self._L.orc_engine_block_mt(self._h, _p(rows, C.c_int8), _p(rc_arr, C.c_uint32), _p(mk, C.c_uint8),
                                       int(bool(refnoise_enabled)), int(seq), _p(lag, C.c_int32),
                                       _p(mag, C.c_float), _p(frac, C.c_float), _p(ph, C.c_float),
                                       _p(pkt, C.c_int8), int(nthreads))
        if rc:
            raise RuntimeError("orc_engine_block failed")
        return dict(lag=lag, mag=mag, frac=frac, phasor=ph.view(np.complex64), packet=pkt,
                    matrix=None if pkt is None else pkt[self.matrix_offset:].reshape(self.nrows, self.B))

    def close(self):
        if self._h:
            self._L.orc_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
