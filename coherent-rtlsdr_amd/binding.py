"""ctypes view of the C ABI in include/crsdr.h (libcrsdr.so).

This is plumbing for tests and bench.py; the product boundary is the C ABI itself.  There is
no fallback: if the HIP extension is missing or no device is present, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("CRSDR_LIB") or os.path.join(_HERE, "libcrsdr.so")   # CRSDR_LIB: A/B a second build (diagnostics)
ROOT = os.path.dirname(_HERE)

MODE_FAITHFUL, MODE_DIGITAL = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
REFNOISE_ENABLED, OFFSET_BINARY, INPUT_READY, NO_LAG = 1, 2, 4, 8

# every symbol include/crsdr.h declares (tests check the .so exports all of them)
ABI_SYMBOLS = [
    "crsdr_abi_version", "crsdr_last_error", "crsdr_device_count",
    "crsdr_convtosigned", "crsdr_convtofloat", "crsdr_scalarmul", "crsdr_convto8bit",
    "crsdr_conj_dotproduct", "crsdr_magsquared", "crsdr_conjugatemul", "crsdr_indexofmax", "crsdr_fft",
    "crsdr_plan_create", "crsdr_plan_destroy", "crsdr_plan_reset", "crsdr_plan_set_stream",
    "crsdr_plan_submit", "crsdr_plan_fetch", "crsdr_plan_sync", "crsdr_plan_packet_bytes",
    "crsdr_plan_matrix_offset", "crsdr_plan_device_buffers", "crsdr_plan_bind_packet",
    "crsdr_plan_last_elapsed_ms", "crsdr_plan_enable_profiling", "crsdr_plan_kernel_times",
    "crsdr_plan_submit_batch", "crsdr_plan_fetch_block", "crsdr_plan_packet_stride", "crsdr_covariance",
    "crsdr_noisesubspace", "crsdr_pmusic2d", "crsdr_plan_bind_slab", "crsdr_assemble_slabs",
    "crsdr_device_info", "crsdr_host_alloc", "crsdr_host_free",
    "crsdr_plan_bind_slab_ex", "crsdr_exchange_geometry", "crsdr_exchange_rooted_blocks", "crsdr_assemble_slots",
    "crsdr_plan_set_frac_apply", "crsdr_plan_fetch_batch_async", "crsdr_plan_fetch_wait", "crsdr_exchange_unique_id", "crsdr_exchange_create", "crsdr_exchange_destroy", "crsdr_exchange_batch", "crsdr_exchange_schedule",
    "crsdr_exchange_bind_plan", "crsdr_exchange_submit_batch", "crsdr_exchange_fetch_rooted",
]
XCHG_STAGED, XCHG_INPLACE = 0, 1
EXCHANGE_ID_BYTES = 128
KERNEL_REF_SPECTRUM, KERNEL_XCORR_LAG, KERNEL_PHASE_DOT, KERNEL_ALIGN_QUANT = 0, 1, 2, 3


class CrsdrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"crsdr error {code}: {msg}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile libcrsdr.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(ROOT, "include", "crsdr.h")]
    stale = force or not os.path.exists(_SO) or any(os.path.getmtime(d) > os.path.getmtime(_SO) for d in deps)
    if stale:
        subprocess.run(["make", "-C", csrc], check=True)
    return _SO


class XOp(C.Structure):
    """crsdr_xop: one point-to-point operation of an exchange batch (crsdr_exchange_schedule)."""
    _fields_ = [("peer", C.c_int32), ("is_recv", C.c_int32), ("buffer", C.c_int32), ("block", C.c_int32),
                ("offset", C.c_uint64), ("bytes", C.c_uint64)]


class PlanDesc(C.Structure):
    _fields_ = [("nrows", C.c_int32), ("blocksize", C.c_int32), ("mode", C.c_int32), ("device", C.c_int32),
                ("row_begin", C.c_int32), ("row_count", C.c_int32), ("max_batch", C.c_int32), ("reserved", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise FileNotFoundError(f"{_SO} is missing: run __graft_entry__.build() (hipcc) first; there is no CPU fallback")
    # A Python process that also uses PyTorch must load torch's bundled HIP runtime BEFORE this library: both
    # carry the soname libamdhip64.so.7, the first one loaded serves both, and torch's device init fails on the
    # system runtime.  (C / C++ hosts are not affected: they link one runtime.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(_SO)
    f32p, i8p, u8p, u32p, i32p, vp = (C.POINTER(C.c_float), C.POINTER(C.c_int8), C.POINTER(C.c_uint8),
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.c_void_p)
    L.crsdr_abi_version.restype = C.c_int
    L.crsdr_last_error.restype = C.c_char_p
    L.crsdr_device_count.argtypes = [C.POINTER(C.c_int)]
    L.crsdr_host_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    L.crsdr_host_free.argtypes = [C.c_void_p]
    L.crsdr_device_info.argtypes = [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    L.crsdr_convtosigned.argtypes = [u8p, u8p, C.c_int]
    L.crsdr_convtofloat.argtypes = [f32p, i8p, C.c_int]
    L.crsdr_scalarmul.argtypes = [f32p, f32p, C.c_float, C.c_float, C.c_int]
    L.crsdr_convto8bit.argtypes = [i8p, f32p, C.c_int]
    L.crsdr_conj_dotproduct.argtypes = [f32p, f32p, f32p, C.c_int]
    L.crsdr_magsquared.argtypes = [f32p, f32p, C.c_int]
    L.crsdr_conjugatemul.argtypes = [f32p, f32p, f32p, C.c_int]
    L.crsdr_indexofmax.argtypes = [u32p, f32p, C.c_int]
    L.crsdr_fft.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_int]
    L.crsdr_covariance.argtypes = [f32p, i8p, C.c_int, C.c_int, C.c_int]
    L.crsdr_noisesubspace.argtypes = [f32p, f32p, f32p, C.c_int, C.c_int]
    L.crsdr_pmusic2d.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.crsdr_plan_create.argtypes = [C.POINTER(vp), C.POINTER(PlanDesc)]
    L.crsdr_plan_destroy.argtypes = [vp]
    L.crsdr_plan_reset.argtypes = [vp]
    L.crsdr_plan_set_stream.argtypes = [vp, vp]
    L.crsdr_plan_submit.argtypes = [vp, vp, C.c_int, u32p, u8p, C.c_uint32, C.c_uint32]
    L.crsdr_plan_fetch.argtypes = [vp, i32p, f32p, f32p, f32p, i8p]
    L.crsdr_plan_sync.argtypes = [vp]
    L.crsdr_plan_packet_bytes.argtypes = [vp]
    L.crsdr_plan_packet_bytes.restype = C.c_size_t
    L.crsdr_plan_matrix_offset.argtypes = [vp]
    L.crsdr_plan_matrix_offset.restype = C.c_size_t
    L.crsdr_plan_device_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.crsdr_plan_bind_packet.argtypes = [vp, vp, C.c_size_t]
    L.crsdr_plan_bind_slab.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int]
    L.crsdr_assemble_slabs.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp]
    L.crsdr_plan_submit_batch.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, u32p, u8p, C.c_uint32, C.c_uint32]
    szp = C.POINTER(C.c_size_t)
    L.crsdr_plan_set_frac_apply.argtypes = [vp, C.c_int, C.c_float, f32p]
    L.crsdr_plan_fetch_batch_async.argtypes = [vp, vp, vp, vp, vp, vp, C.c_size_t]
    L.crsdr_plan_fetch_wait.argtypes = [vp]
    L.crsdr_plan_bind_slab_ex.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, C.c_size_t]
    L.crsdr_exchange_geometry.argtypes = [C.c_int, C.c_int, C.c_int, szp, szp, szp]
    L.crsdr_exchange_rooted_blocks.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.crsdr_assemble_slots.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int, vp, vp]
    L.crsdr_exchange_unique_id.argtypes = [vp]
    L.crsdr_exchange_create.argtypes = [C.POINTER(vp), vp, C.c_int, C.c_int, C.c_int]
    L.crsdr_exchange_destroy.argtypes = [vp]
    L.crsdr_exchange_batch.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp, C.c_size_t, vp, C.c_size_t, C.c_int, C.c_int, vp]
    L.crsdr_exchange_schedule.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.POINTER(XOp), C.c_int, C.POINTER(C.c_int)]
    L.crsdr_exchange_bind_plan.argtypes = [vp, vp, C.c_int]
    L.crsdr_exchange_submit_batch.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, u32p, u8p, C.c_uint32, C.c_uint32]
    L.crsdr_exchange_fetch_rooted.argtypes = [vp, C.POINTER(C.c_int8), C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                              C.POINTER(C.c_int)]
    L.crsdr_plan_fetch_block.argtypes = [vp, C.c_int, i32p, f32p, f32p, f32p, i8p]
    L.crsdr_plan_packet_stride.argtypes = [vp]
    L.crsdr_plan_packet_stride.restype = C.c_size_t
    L.crsdr_plan_last_elapsed_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.crsdr_plan_enable_profiling.argtypes = [vp, C.c_int, C.c_uint32]
    L.crsdr_plan_kernel_times.argtypes = [vp, C.c_int, f32p, C.c_int, C.POINTER(C.c_int)]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise CrsdrError(rc, lib().crsdr_last_error().decode(errors="replace"))


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty)) if a is not None else None


def _c64(a):
    a = np.ascontiguousarray(a, dtype=np.complex64)
    return a, a.view(np.float32)


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().crsdr_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def device_info(device: int = 0) -> dict:
    name = C.create_string_buffer(256)
    cus, clk, mclk, mem = C.c_int(0), C.c_int(0), C.c_int(0), C.c_size_t(0)
    _check(lib().crsdr_device_info(int(device), name, 256, C.byref(cus), C.byref(clk), C.byref(mclk), C.byref(mem)))
    return {"name": name.value.decode(errors="replace"), "compute_units": cus.value, "clock_mhz": clk.value / 1e3,
            "memory_clock_mhz": mclk.value / 1e3, "memory_gib": round(mem.value / 2**30, 1)}


class PinnedArray:
    """numpy view of page-locked host memory from crsdr_host_alloc (freed on close / garbage collection)."""

    def __init__(self, shape, dtype=np.int8):
        self.shape, self.dtype = tuple(np.atleast_1d(shape)), np.dtype(dtype)
        nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self._ptr = C.c_void_p()
        _check(lib().crsdr_host_alloc(C.byref(self._ptr), nbytes))
        self.array = np.frombuffer((C.c_uint8 * nbytes).from_address(self._ptr.value), dtype=self.dtype).reshape(self.shape)

    def close(self):
        if self._ptr:
            self.array = None
            lib().crsdr_host_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- per-op wrappers (class cdsp), numpy host arrays in / out ----------------------------------

def convtosigned(u8):
    u8 = np.ascontiguousarray(u8, dtype=np.uint8)
    out = np.empty_like(u8)
    _check(lib().crsdr_convtosigned(_p(u8, C.c_uint8), _p(out, C.c_uint8), u8.size))
    return out.view(np.int8)


def convtofloat(i8):
    i8 = np.ascontiguousarray(i8, dtype=np.int8)
    out = np.empty(i8.size, dtype=np.float32)
    _check(lib().crsdr_convtofloat(_p(out, C.c_float), _p(i8, C.c_int8), i8.size))
    return out.view(np.complex64)


def scalarmul(x, s):
    x, xf = _c64(x)
    out = np.empty_like(x)
    s = np.complex64(s)
    _check(lib().crsdr_scalarmul(_p(out.view(np.float32), C.c_float), _p(xf, C.c_float), float(s.real), float(s.imag), x.size))
    return out


def convto8bit(x):
    x, xf = _c64(x)
    out = np.empty(2 * x.size, dtype=np.int8)
    _check(lib().crsdr_convto8bit(_p(out, C.c_int8), _p(xf, C.c_float), x.size))
    return out


def conj_dotproduct(a, b):
    a, af = _c64(a)
    b, bf = _c64(b)
    res = np.zeros(2, dtype=np.float32)
    _check(lib().crsdr_conj_dotproduct(_p(res, C.c_float), _p(af, C.c_float), _p(bf, C.c_float), a.size))
    return np.complex64(res[0] + 1j * res[1])


def magsquared(x):
    x, xf = _c64(x)
    out = np.empty(x.size, dtype=np.float32)
    _check(lib().crsdr_magsquared(_p(out, C.c_float), _p(xf, C.c_float), x.size))
    return out


def conjugatemul(a, b):
    a, af = _c64(a)
    b, bf = _c64(b)
    out = np.empty_like(a)
    _check(lib().crsdr_conjugatemul(_p(out.view(np.float32), C.c_float), _p(af, C.c_float), _p(bf, C.c_float), a.size))
    return out


def indexofmax(m):
    m = np.ascontiguousarray(m, dtype=np.float32)
    idx = C.c_uint32(0)
    _check(lib().crsdr_indexofmax(C.byref(idx), _p(m, C.c_float), m.size))
    return int(idx.value)


def fft(x, sign=-1):
    x, xf = _c64(x)
    n = x.shape[-1]
    out = np.empty_like(x)
    _check(lib().crsdr_fft(_p(out.view(np.float32), C.c_float), _p(xf, C.c_float), n, sign, x.size // n))
    return out


def covariance(matrix):
    """Rxx of the signal rows of an aligned int8 matrix [nrows][B] (beamformer's first step)."""
    m = np.ascontiguousarray(matrix, dtype=np.int8)
    nrows, B = m.shape
    out = np.empty((nrows - 1, nrows - 1), dtype=np.complex64)
    _check(lib().crsdr_covariance(_p(out.view(np.float32), C.c_float), _p(m, C.c_int8), nrows, B, MEM_HOST))
    return out


def covariance_device(rxx_ptr: int, matrix_ptr: int, nrows: int, B: int):
    """Same on device memory (both pointers on the current device); returns after the kernels finished."""
    _check(lib().crsdr_covariance(C.cast(C.c_void_p(int(rxx_ptr)), C.POINTER(C.c_float)),
                                  C.cast(C.c_void_p(int(matrix_ptr)), C.POINTER(C.c_int8)), nrows, B, MEM_DEVICE))


def noisesubspace(rxx):
    """noisesubspace(Rxx, K) of beamformclient/heatmap2d2.cpp:69-79 without the column cut: returns (vec, sv) with
    vec[:, r] the singular vector of sv[r], sv descending; the noise subspace for K sources is vec[:, K:]."""
    r = np.ascontiguousarray(rxx, dtype=np.complex64)
    if r.ndim != 2 or r.shape[0] != r.shape[1]:
        raise ValueError("rxx must be square")
    M = r.shape[0]
    vec = np.empty((M, M), dtype=np.complex64)
    sv = np.empty(M, dtype=np.float32)
    _check(lib().crsdr_noisesubspace(_p(vec.view(np.float32), C.c_float), _p(sv, C.c_float), _p(r.view(np.float32), C.c_float), M, MEM_HOST))
    return vec, sv


def pmusic2d(vec, k, d, mx, my, ncx=100, ncy=100):
    """pmusic2dvec(Un, d, Mx, My, Cx, Cy) of beamformclient/heatmap2d2.cpp:137-147 with Un = vec[:, k:]."""
    v = np.ascontiguousarray(vec, dtype=np.complex64)
    M = v.shape[0]
    pm = np.empty((max(ncx, 0), max(ncy, 0)), dtype=np.float32)
    _check(lib().crsdr_pmusic2d(_p(pm, C.c_float), _p(v.view(np.float32), C.c_float), M, int(k), C.c_float(d), int(mx), int(my),
                                int(ncx), int(ncy), MEM_HOST))
    return pm


def assemble_slabs(packets_ptr: int, packet_stride: int, nrows: int, B: int, recv_ptr: int, nsrc: int, nblocks: int, stream: int | None = None):
    """crsdr_assemble_slabs: the received slabs [nsrc][nblocks][per][B] into the matrix rows of nblocks packets (device pointers)."""
    _check(lib().crsdr_assemble_slabs(C.c_void_p(int(packets_ptr)), int(packet_stride), int(nrows), int(B), C.c_void_p(int(recv_ptr)),
                                      int(nsrc), int(nblocks), C.c_void_p(stream or 0)))


# ---- exchange slots + the RCCL exchange --------------------------------------------------

def exchange_geometry(nrows: int, B: int, nranks: int) -> dict:
    """crsdr_exchange_geometry: slot_stride / tail_offset of a rank's send slots and the stride of an assembled scalars block."""
    a, t, s = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    _check(lib().crsdr_exchange_geometry(int(nrows), int(B), int(nranks), C.byref(a), C.byref(t), C.byref(s)))
    return {"slot_stride": a.value, "tail_offset": t.value, "scalars_stride": s.value, "per": (nrows - 1) // nranks}


def rooted_blocks(nblocks: int, nranks: int, rank: int) -> range:
    """crsdr_exchange_rooted_blocks: the blocks of a batch of nblocks that `rank` assembles."""
    f, c = C.c_int(0), C.c_int(0)
    _check(lib().crsdr_exchange_rooted_blocks(int(nblocks), int(nranks), int(rank), C.byref(f), C.byref(c)))
    return range(f.value, f.value + c.value)


def assemble_slots(packets_ptr, packet_stride, scalars_ptr, scalars_stride, nrows, B, recv_ptr, nsrc, nblocks, slot_stride, tail_offset,
                   self_rank=-1, self_ptr=None, stream=None):
    """crsdr_assemble_slots: received slots [nsrc][nblocks][slot_stride] -> matrix rows of nblocks packets + their scalars blocks."""
    _check(lib().crsdr_assemble_slots(C.c_void_p(int(packets_ptr)), int(packet_stride), C.c_void_p(int(scalars_ptr or 0)), int(scalars_stride),
                                      int(nrows), int(B), C.c_void_p(int(recv_ptr or 0)), int(nsrc), int(nblocks), int(slot_stride), int(tail_offset),
                                      int(self_rank), C.c_void_p(int(self_ptr or 0)), C.c_void_p(stream or 0)))


def parse_scalars(block: np.ndarray, nrows: int) -> dict:
    """One assembled scalars block (bytes) -> {lag, mag, frac, phasor} like Plan.fetch."""
    raw = np.ascontiguousarray(block).view(np.uint8)
    n = nrows
    return dict(lag=raw[:4 * n].view(np.int32).copy(), mag=raw[4 * n:8 * n].view(np.float32).copy(),
                frac=raw[8 * n:12 * n].view(np.float32).copy(), phasor=raw[12 * n:20 * n].view(np.complex64).copy())


def exchange_schedule(nranks, rank, nblocks, mode, nrows, B, packet_stride=0):
    """crsdr_exchange_schedule: the point-to-point operations of one batch for `rank`, in issue order (list of dicts)."""
    n = C.c_int(0)
    _check(lib().crsdr_exchange_schedule(int(nranks), int(rank), int(nblocks), int(mode), int(nrows), int(B), int(packet_stride), None, 0, C.byref(n)))
    ops = (XOp * max(1, n.value))()
    _check(lib().crsdr_exchange_schedule(int(nranks), int(rank), int(nblocks), int(mode), int(nrows), int(B), int(packet_stride), ops, n.value, C.byref(n)))
    return [dict(peer=o.peer, is_recv=bool(o.is_recv), buffer=o.buffer, block=o.block, offset=int(o.offset), bytes=int(o.bytes)) for o in ops[: n.value]]


def exchange_unique_id() -> bytes:
    buf = C.create_string_buffer(EXCHANGE_ID_BYTES)
    _check(lib().crsdr_exchange_unique_id(buf))
    return buf.raw


class Exchange:
    """crsdr_exchange: the slot exchange over RCCL under the C ABI (one per rank; every rank passes rank 0's unique id)."""

    def __init__(self, unique_id: bytes, nranks: int, rank: int, device: int = 0):
        h = C.c_void_p()
        idbuf = C.create_string_buffer(bytes(unique_id), EXCHANGE_ID_BYTES)
        _check(lib().crsdr_exchange_create(C.byref(h), idbuf, int(nranks), int(rank), int(device)))
        self._h, self.nranks, self.rank = h, nranks, rank
        self._plan, self._keep = None, []

    def batch(self, mode, send_ptr, recv_ptr, nblocks, packets_ptr, packet_stride, scalars_ptr, scalars_stride, nrows, B, stream=None):
        _check(lib().crsdr_exchange_batch(self._h, int(mode), C.c_void_p(int(send_ptr)), C.c_void_p(int(recv_ptr or 0)), int(nblocks),
                                          C.c_void_p(int(packets_ptr)), int(packet_stride), C.c_void_p(int(scalars_ptr or 0)), int(scalars_stride),
                                          int(nrows), int(B), C.c_void_p(stream or 0)))

    # ---- a sharded plan and its exchange as one engine (crsdr_exchange_bind_plan / _submit_batch / _fetch_rooted) ----
    def bind_plan(self, plan, mode=None):
        _check(lib().crsdr_exchange_bind_plan(self._h, plan._h, int(XCHG_STAGED if mode is None else mode)))
        self._plan, self._keep = plan, []

    def submit_batch(self, rows, readcnt=None, lag_mask=None, seq=0, flags=REFNOISE_ENABLED):
        """rows: numpy int8/uint8 [nblocks][nrows][B] on the host (page-locked or not): plan submit + exchange, all enqueued on return."""
        keep = np.ascontiguousarray(rows)
        if keep.ndim == 2:
            keep = keep[None]
        assert keep.shape[1:] == (self._plan.nrows, self._plan.B) and keep.dtype in (np.int8, np.uint8)
        rc_arr = None if readcnt is None else np.ascontiguousarray(readcnt, dtype=np.uint32)
        mk = None if lag_mask is None else np.ascontiguousarray(lag_mask, dtype=np.uint8)
        self._keep.append(keep)                     # stays alive until the batch has been fetched
        _check(lib().crsdr_exchange_submit_batch(self._h, C.c_void_p(keep.ctypes.data), MEM_HOST, keep.shape[0], 0, _p(rc_arr, C.c_uint32),
                                                 _p(mk, C.c_uint8), int(seq), int(flags)))

    def fetch_rooted(self):
        """The oldest outstanding batch: (first, packets [count][packet_bytes] int8, scalars (list of parse_scalars dicts), own_tails dict of
        [nblocks][per] arrays: lag, mag, frac, phasor, readcnt)."""
        p = self._plan
        f, c, nb = C.c_int(), C.c_int(), C.c_int()
        if p is None:                                    # nothing bound: let the library say so
            _check(lib().crsdr_exchange_fetch_rooted(self._h, None, 0, None, 0, None, 0, C.byref(f), C.byref(c), C.byref(nb)))
        T, n, per = p.max_batch, p.nrows, (p.nrows - 1) // self.nranks
        bpr = -(-T // self.nranks)
        pk = np.zeros((bpr, p.packet_bytes), dtype=np.int8)
        sstride = (20 * n + 15) // 16 * 16
        tstride = (24 * per + 15) // 16 * 16
        sc = np.zeros((bpr, sstride), dtype=np.uint8)
        tl = np.zeros((T, tstride), dtype=np.uint8)
        _check(lib().crsdr_exchange_fetch_rooted(self._h, _p(pk, C.c_int8), pk.strides[0], C.c_void_p(sc.ctypes.data), sstride,
                                                 C.c_void_p(tl.ctypes.data), tstride, C.byref(f), C.byref(c), C.byref(nb)))
        if self._keep:
            self._keep.pop(0)
        t = tl[:nb.value]
        tails = dict(lag=t[:, :4 * per].copy().view(np.int32), mag=t[:, 4 * per:8 * per].copy().view(np.float32),
                     frac=t[:, 8 * per:12 * per].copy().view(np.float32), phasor=t[:, 12 * per:20 * per].copy().view(np.complex64),
                     readcnt=t[:, 20 * per:24 * per].copy().view(np.uint32))
        return f.value, pk[:c.value], [parse_scalars(sc[j], n) for j in range(c.value)], tails

    def close(self):
        if getattr(self, "_h", None):
            lib().crsdr_exchange_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- batched plan ---------------------------------------------------------------------------------

class Plan:
    """crsdr_plan: ccoherent + per-row csdrdevice DSP members + cpacketize::write on one GPU."""

    def __init__(self, nrows, blocksize, mode=MODE_FAITHFUL, device=0, row_begin=0, row_count=0, max_batch=1):
        self.nrows, self.B, self.mode, self.max_batch = nrows, blocksize, mode, max_batch
        d = PlanDesc(nrows, blocksize, mode, device, row_begin, row_count, max_batch, 0)
        h = C.c_void_p()
        _check(lib().crsdr_plan_create(C.byref(h), C.byref(d)))
        self._h = h
        self._keep = []
        self.packet_bytes = int(lib().crsdr_plan_packet_bytes(h))
        self.matrix_offset = int(lib().crsdr_plan_matrix_offset(h))

    def set_stream(self, hip_stream: int | None):
        _check(lib().crsdr_plan_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def submit(self, rows, readcnt=None, lag_mask=None, seq=0, flags=REFNOISE_ENABLED, nblocks=None, block_stride=0):
        """rows: numpy int8/uint8 [nrows][B] (or [nblocks][nrows][B]) on the host, or an int device pointer."""
        rc_arr = None if readcnt is None else np.ascontiguousarray(readcnt, dtype=np.uint32)
        mk = None if lag_mask is None else np.ascontiguousarray(lag_mask, dtype=np.uint8)
        if isinstance(rows, (int, np.integer)):
            ptr, kind = C.c_void_p(int(rows)), MEM_DEVICE
            nb = 1 if nblocks is None else int(nblocks)
        else:
            if rows.dtype not in (np.int8, np.uint8):
                raise TypeError("rows must be int8 (or offset-binary uint8)")
            keep = np.ascontiguousarray(rows)
            if keep.ndim == 2:
                keep = keep[None]
            assert keep.shape[1:] == (self.nrows, self.B)
            nb = keep.shape[0] if nblocks is None else int(nblocks)
            self._keep.append(keep)          # stays alive until the next fetch()/sync()
            ptr, kind = C.c_void_p(keep.ctypes.data), MEM_HOST
        _check(lib().crsdr_plan_submit_batch(self._h, ptr, kind, nb, int(block_stride), _p(rc_arr, C.c_uint32),
                                             _p(mk, C.c_uint8), int(seq), int(flags)))

    def fetch(self, want_packet=True, block=-1):
        n = self.nrows
        lag = np.zeros(n, dtype=np.int32)
        mag = np.zeros(n, dtype=np.float32)
        frac = np.zeros(n, dtype=np.float32)
        ph = np.zeros(2 * n, dtype=np.float32)
        pkt = np.zeros(self.packet_bytes, dtype=np.int8) if want_packet else None
        _check(lib().crsdr_plan_fetch_block(self._h, int(block), _p(lag, C.c_int32), _p(mag, C.c_float), _p(frac, C.c_float),
                                            _p(ph, C.c_float), _p(pkt, C.c_int8)))
        if block == -1:
            self._keep.clear()
        return dict(lag=lag, mag=mag, frac=frac, phasor=ph.view(np.complex64), packet=pkt,
                    matrix=None if pkt is None else pkt[self.matrix_offset:].reshape(self.nrows, self.B))

    def fetch_batch_async(self, lag=None, mag=None, frac=None, phasor=None, packets=None, host_packet_stride=0):
        """crsdr_plan_fetch_batch_async: numpy views of page-locked memory (PinnedArray.array) or None"""
        ptr = lambda a: C.c_void_p(a.ctypes.data) if a is not None else None
        _check(lib().crsdr_plan_fetch_batch_async(self._h, ptr(lag), ptr(mag), ptr(frac), ptr(phasor), ptr(packets), int(host_packet_stride)))

    def fetch_wait(self):
        _check(lib().crsdr_plan_fetch_wait(self._h))

    def block(self, rows, **kw):
        self.submit(rows, **kw)
        return self.fetch()

    def sync(self):
        _check(lib().crsdr_plan_sync(self._h))
        self._keep.clear()

    def reset(self):
        _check(lib().crsdr_plan_reset(self._h))

    def device_buffers(self):
        ptrs = [C.c_void_p() for _ in range(5)]
        _check(lib().crsdr_plan_device_buffers(self._h, *[C.byref(p) for p in ptrs]))
        return dict(zip(("packet", "lag", "mag", "frac", "phasor"), [p.value for p in ptrs]))

    def bind_packet(self, device_ptr: int | None, packet_stride: int = 0):
        _check(lib().crsdr_plan_bind_packet(self._h, C.c_void_p(device_ptr or 0), int(packet_stride)))

    def bind_slab(self, device_ptr: int | None, slab_stride: int = 0, hdr_first: int = 0, hdr_count: int = 0):
        """Slab output for sharded plans (crsdr_plan_bind_slab); None returns to packet output."""
        _check(lib().crsdr_plan_bind_slab(self._h, C.c_void_p(device_ptr or 0), int(slab_stride), int(hdr_first), int(hdr_count)))

    def set_frac_apply(self, enable=True, gain=1.0, frac_override=None):
        """crsdr_plan_set_frac_apply: fractional-delay correction of the matrix rows (long-block plans, digital mode)."""
        ov = None if frac_override is None else np.ascontiguousarray(frac_override, dtype=np.float32)
        # enable: False / 0 off, True / 1 on, 2 on without the second work area (the pass repeats its first stage)
        _check(lib().crsdr_plan_set_frac_apply(self._h, int(enable), C.c_float(gain), _p(ov, C.c_float)))

    def bind_slab_ex(self, device_ptr: int | None, slab_stride: int = 0, hdr_first: int = 0, hdr_count: int = 0, tail_offset: int = 0):
        """crsdr_plan_bind_slab_ex: slab output with the per-row {lag, mag, frac, phasor} tail behind the rows of every slot."""
        _check(lib().crsdr_plan_bind_slab_ex(self._h, C.c_void_p(device_ptr or 0), int(slab_stride), int(hdr_first), int(hdr_count), int(tail_offset)))

    @property
    def packet_stride(self) -> int:
        return int(lib().crsdr_plan_packet_stride(self._h))

    def last_elapsed_ms(self) -> float:
        ms = C.c_float(0)
        _check(lib().crsdr_plan_last_elapsed_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def enable_profiling(self, slots: int, kernel_mask: int = 0xF):
        """kernel_mask: bit KERNEL_*; 1 << 31 adds whole-submit start/stop events"""
        _check(lib().crsdr_plan_enable_profiling(self._h, int(slots), int(kernel_mask)))

    def kernel_times_ms(self, which: int, capacity: int = 4096) -> np.ndarray:
        out = np.zeros(capacity, dtype=np.float32)
        n = C.c_int(0)
        _check(lib().crsdr_plan_kernel_times(self._h, int(which), _p(out, C.c_float), capacity, C.byref(n)))
        return out[: n.value].copy()

    def close(self):
        if getattr(self, "_h", None):
            lib().crsdr_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
