"""Wire format + ZMQ publish: cpacketize::init/send publish hdr0 + readcnt + int8 matrix on a
ZMQ PUB socket and the N phase factors on the debug PUB (src/cpacketizer.cc:58-74,109-129).  A SUB client
written against libzmq's C API (ctypes) parses the packets exactly like the reference's MATLAB MEX client
(matlabclient/zmqsdr.c:118-144): u32 gseq, u32 c (channels), u32 r (samples), offset 16 + 4c, then
r*c*2 int8 values scaled by 1/128."""
import ctypes as C
import os
import socket
import subprocess
import threading
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "coherent-rtlsdr_amd", "host")
ZMQ_SUB, ZMQ_SUBSCRIBE, ZMQ_RCVTIMEO, ZMQ_LINGER = 2, 6, 27, 17


def _libzmq():
    for n in ("libzmq.so.5", "libzmq.so", "/opt/conda/lib/libzmq.so.5"):
        try:
            z = C.CDLL(n)
            break
        except OSError:
            z = None
    if z is None:
        pytest.skip("libzmq not present on this host")
    z.zmq_ctx_new.restype = C.c_void_p
    z.zmq_socket.restype = C.c_void_p
    z.zmq_socket.argtypes = [C.c_void_p, C.c_int]
    z.zmq_connect.argtypes = [C.c_void_p, C.c_char_p]
    z.zmq_setsockopt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    z.zmq_recv.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    z.zmq_close.argtypes = [C.c_void_p]
    z.zmq_ctx_term.argtypes = [C.c_void_p]
    return z


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class Subscriber(threading.Thread):
    def __init__(self, z, addr, bufsize, want):
        super().__init__(daemon=True)
        self.z, self.addr, self.bufsize, self.want, self.msgs = z, addr, bufsize, want, []

    def run(self):
        z = self.z
        ctx = z.zmq_ctx_new()
        s = z.zmq_socket(ctx, ZMQ_SUB)
        tmo, lin = C.c_int(4000), C.c_int(0)
        z.zmq_setsockopt(s, ZMQ_RCVTIMEO, C.byref(tmo), 4)
        z.zmq_setsockopt(s, ZMQ_LINGER, C.byref(lin), 4)
        z.zmq_setsockopt(s, ZMQ_SUBSCRIBE, b"", 0)
        z.zmq_connect(s, self.addr.encode())
        buf = (C.c_char * self.bufsize)()
        while len(self.msgs) < self.want:
            n = z.zmq_recv(s, buf, self.bufsize, 0)
            if n <= 0:
                break
            self.msgs.append(bytes(buf[:n]))
        z.zmq_close(s)
        z.zmq_ctx_term(ctx)


def _parse_like_zmqsdr_c(msg):
    """matlabclient/zmqsdr.c:118-144"""
    gseq, c, r = np.frombuffer(msg[:12], dtype=np.uint32)
    offset = 16 + 4 * int(c)
    seqs = np.frombuffer(msg[16:offset], dtype=np.uint32)
    data = np.frombuffer(msg[offset:offset + 2 * int(r) * int(c)], dtype=np.int8)
    X = (data.astype(np.float32) * np.float32(1.0 / 128.0)).view(np.complex64).reshape(int(c), int(r))
    return int(gseq), int(c), int(r), seqs, data.reshape(int(c), 2 * int(r)), X


@pytest.fixture(scope="module")
def host_build():
    import importlib
    importlib.import_module("coherent-rtlsdr_amd.binding").build()
    subprocess.run(["make", "-C", HOST, "all"], check=True, stdout=subprocess.DEVNULL)
    return HOST


def test_packetizer_publishes_reference_wire_format(host_build, synth):
    z = _libzmq()
    nsig, L, blocks = 3, 256, 40
    p1, p2 = _free_port(), _free_port()
    addr, dbg = f"tcp://127.0.0.1:{p1}", f"tcp://127.0.0.1:{p2}"
    sub = Subscriber(z, addr, 1 << 20, 5)
    subd = Subscriber(z, dbg, 4096, 5)
    sub.start(); subd.start()
    time.sleep(0.2)
    r = subprocess.run([os.path.join(host_build, "packetizer_selftest"), "--nsig", str(nsig), "--L", str(L), "--blocks",
                        str(blocks), "--zmq", addr, "--zmq-debug", dbg], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    sub.join(10); subd.join(10)
    assert len(sub.msgs) >= 3 and len(subd.msgs) >= 3
    params = synth.RowParams(nsig, L, 4242, dmax=L // 8)
    last = None
    for m in sub.msgs:
        assert len(m) == 16 + 4 * (nsig + 1) + 2 * (nsig + 1) * 2 * L      # the reference's packetlength, src/cpacketizer.cc:91-96
        gseq, c, rr, seqs, data, X = _parse_like_zmqsdr_c(m)
        assert (c, rr) == (nsig + 1, L)
        assert np.all(seqs == 1000 + gseq)                    # per-channel readcnt words
        rows, _ = synth.make_block(nsig, L, 4242, gseq, params=params)
        assert np.array_equal(data, rows)                     # the int8 matrix, row-major by channel
        assert np.allclose(X, (rows.astype(np.float32) / 128).view(np.complex64))
        if last is not None:
            assert gseq == last + 1                           # globalseqn continuity (README.md:42)
        last = gseq
    for m in subd.msgs:                                       # debug PUB: N complex<float> phase factors
        ph = np.frombuffer(m, dtype=np.complex64)
        assert ph.size == nsig + 1 and np.array_equal(ph.real, np.arange(nsig + 1))


@pytest.mark.parametrize("refpadding", [None, 1, 0])
def test_on_wire_message_length_is_the_reference_formula(host_build, synth, refpadding):
    # the reference sends packetlength = (16 + 4N) + 2*N*blocksize bytes per message (src/cpacketizer.cc:91-96,125: it
    # doubles the data size), of which clients parse the first N*blocksize data bytes (matlabclient/zmqsdr.c:121-143).
    # That length is the DEFAULT on the wire (zero tail); cpacketize::refpadding = false is the opt-in short form.
    z = _libzmq()
    nsig, L, blocks = 2, 128, 25
    addr, dbg = f"tcp://127.0.0.1:{_free_port()}", f"tcp://127.0.0.1:{_free_port()}"
    sub = Subscriber(z, addr, 1 << 20, 3)
    sub.start()
    time.sleep(0.2)
    extra = [] if refpadding is None else ["--refpadding", str(refpadding)]
    r = subprocess.run([os.path.join(host_build, "packetizer_selftest"), "--nsig", str(nsig), "--L", str(L), "--blocks", str(blocks),
                        "--zmq", addr, "--zmq-debug", dbg] + extra, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    sub.join(10)
    assert len(sub.msgs) >= 3
    N, B = nsig + 1, 2 * L
    ref_len = (16 + 4 * N) + 2 * N * B                         # packetlength() of the reference with its own argument names
    params = synth.RowParams(nsig, L, 4242, dmax=L // 8)
    for m in sub.msgs:
        assert len(m) == (ref_len if refpadding != 0 else ref_len - N * B)
        gseq, c, rr, seqs, data, _ = _parse_like_zmqsdr_c(m)
        rows, _ = synth.make_block(nsig, L, 4242, gseq, params=params)
        assert (c, rr) == (N, L) and np.array_equal(data, rows)
        assert not any(m[16 + 4 * N + N * B:])                 # the tail (if any) is zero


@pytest.mark.gpu
def test_engine_publishes_aligned_matrix_over_zmq(host_build, synth):
    # full chain on the GPU box: csyntheticsdr -> ccoherent::step (libcrsdr plan) -> cpacketize::send -> ZMQ SUB
    z = _libzmq()
    nsig, L, blocks = 3, 8192, 30
    addr, dbg = f"tcp://127.0.0.1:{_free_port()}", f"tcp://127.0.0.1:{_free_port()}"
    sub = Subscriber(z, addr, 1 << 20, 6)
    sub.start()
    time.sleep(0.2)
    r = subprocess.run([os.path.join(host_build, "coherent_demo"), "--blocks", str(blocks), "--zmq", addr, "--zmq-debug", dbg,
                        "--pace-ms", "20"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "DEMO OK" in r.stdout, r.stdout + r.stderr
    sub.join(10)
    assert len(sub.msgs) >= 3
    params = synth.RowParams(nsig, L, synth.config_seed(1))
    for m in sub.msgs:
        gseq, c, rr, seqs, data, X = _parse_like_zmqsdr_c(m)
        assert (c, rr) == (nsig + 1, L)
        rows, _ = synth.make_block(nsig, L, synth.config_seed(1), gseq, params=params)
        assert np.array_equal(data[0], rows[0])               # row 0 = raw reference block
        if gseq >= 10:                                        # aligned rows: xcorr with the ref row peaks at zero lag
            ref = X[0]
            for k in range(1, nsig + 1):
                cc = np.fft.ifft(np.fft.fft(X[k], 2 * L) * np.conj(np.fft.fft(ref, 2 * L)))
                assert int(np.argmax(np.abs(cc))) == 0
                assert abs(np.angle(cc[0])) < 0.05
