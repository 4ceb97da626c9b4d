"""C++ host side (coherent-rtlsdr_amd/host): the reference's class surface over the C ABI.
CPU: the C synthetic block source reproduces synth.py byte for byte, the host code compiles.
GPU: coherent_demo (cdsp / csdrdevice / ccoherent / cpacketize on config 1) recovers every
injected delay and phase and ships well-formed packets."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "coherent-rtlsdr_amd", "host")


@pytest.fixture(scope="module")
def host_build():
    import importlib
    importlib.import_module("coherent-rtlsdr_amd.binding").build()
    subprocess.run(["make", "-C", HOST, "all", "libcsynth.so"], check=True, stdout=subprocess.DEVNULL)
    return HOST


def _c_block(lib, nsig, L, seed, block, dmax, locked=0):
    lib.csynth_params_create.restype = C.c_void_p
    lib.csynth_params_create.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int]
    lib.csynth_make_block.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p]
    lib.csynth_params_destroy.argtypes = [C.c_void_p]
    p = lib.csynth_params_create(nsig, L, seed, dmax, locked)
    rows = np.zeros((1 + nsig, 2 * L), dtype=np.int8)
    lib.csynth_make_block(p, block, -1.0, rows.ctypes.data)
    lib.csynth_params_destroy(p)
    return rows


@pytest.mark.parametrize("nsig,L,seed,dmax", [(3, 8192, 0xC0FFEE + 1, -1), (7, 512, 12345, 100), (21, 1024, 0xC0FFEE + 2, -1)])
def test_c_synth_matches_python_byte_for_byte(host_build, synth, nsig, L, seed, dmax):
    lib = C.CDLL(os.path.join(host_build, "libcsynth.so"))
    for block in (0, 3):
        got = _c_block(lib, nsig, L, seed, block, dmax)
        exp, _ = synth.make_block(nsig, L, seed, block, dmax=None if dmax < 0 else dmax)
        assert np.array_equal(got, exp)


def test_host_demo_builds(host_build):
    assert os.access(os.path.join(host_build, "coherent_demo"), os.X_OK)


def test_block_ring_under_overrun_never_tears_a_block(host_build):
    # the cbuffer ring with a producer faster than its reader (no GPU work): read() hands the oldest slot to the engine,
    # which copies it OUTSIDE the lock; on overrun the producer must not reuse that slot (a torn block under a valid
    # readcnt) nor make the reader's consume() drop a second, unread block.  ring_selftest checks every block it gets
    # against the generator, that readcnts rise strictly and that consumed + dropped <= produced, with real drops.
    r = subprocess.run([os.path.join(host_build, "ring_selftest"), "400"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "RING OK" in r.stdout, r.stdout + r.stderr


def _servo_restatement(lag, samplerate, blocksize):
    """src/ccontrol.cc:25-29,36-45,73-76,99-119 with the reference's own types: lag, scale, frac_t, p, fs and the
    quotient lag / (p * fs) are float (unqualified tanh / fabs of a float pick the float overloads in C++), maxppm is
    double 2^13 / 2^24, t is a double holding a float product; include/common.h:32 sync_threshold = 0.005f."""
    f32 = np.float32
    lag = f32(lag)
    xtal22 = 28800000.0 * 4194304.0
    fsratio = int(xtal22 / samplerate) & 0x0ffffffc                    # realfs(): :36-45
    real_fsratio = fsratio | ((fsratio & 0x08000000) << 1)
    realfs = xtal22 / real_fsratio
    correct = bool(np.abs(lag) > f32(0.005))                           # :99
    p = f32((2.0 ** 13 / 2.0 ** 24) * float(np.tanh(lag / f32(100.0))))   # descent(): :73-76
    if not correct:
        return correct, float(p), 0.0, 0, realfs
    t = float(f32(0.90) * np.abs(lag / (p * f32(realfs))))             # :102
    block_s = (blocksize // 2) / samplerate                            # the model's clock: one block = L / fs seconds
    return correct, float(p), t, max(1, int(np.ceil(t / block_s))), realfs


@pytest.mark.parametrize("fs", [2048000, 1000000, 250000])
def test_servo_numbers_match_the_reference_formulas(host_build, fs):
    # SURVEY 8 f3, numeric: ccontrol::descent, the hold time at the altered sample rate and its block-quantised count,
    # over a lag grid that includes both sides of sync_threshold (0.005: at the threshold itself the row is marked
    # synchronized, src/ccontrol.cc:99,116-119), small lags (hold ~ 0.9 * 100 * 2^11 / fs whatever the lag), the knee of
    # the tanh and full-scale lags (p saturates at 2^-11, hold grows linearly)
    lags = [0.004, 0.005, 0.0050001, 0.0051, -0.0051, 0.01, 0.5, 1, -1, 2, -7, 13, 50, -99, 100, 150, 250, -400, 1000, 2048, -4095, 8191, -8192]
    L = 8192
    r = subprocess.run([os.path.join(host_build, "coherent_demo"), "--L", str(L), "--fs", str(fs), "--servo-table", ",".join(repr(x) for x in lags)],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("servo lag")]
    assert len(rows) == len(lags)
    for lag, w in zip(lags, rows):
        got = dict(zip(w[1::2], w[2::2]))
        correct, p, t, nblk, realfs = _servo_restatement(lag, fs, 2 * L)
        assert np.float32(float(got["lag"])) == np.float32(lag)
        assert int(got["correct"]) == int(correct), lag
        assert abs(float(got["p"]) - p) <= 2e-7 * abs(p), (lag, got["p"], p)          # float tanh: libm vs numpy, <= 1 ulp
        assert abs(float(got["realfs"]) - realfs) <= 1e-9 * realfs
        if correct:
            assert abs(float(got["t"]) - t) <= 4e-7 * t, (lag, got["t"], t)
            assert abs(int(got["blocks"]) - nblk) <= (1 if abs(t / ((L) / fs) - round(t / (L / fs))) < 1e-5 else 0), (lag, got["blocks"], nblk)
            assert np.sign(float(got["p"])) == np.sign(lag)                           # the resampler is slewed TOWARDS zero lag
    # anchors: below the tanh knee the hold time does not depend on the lag: 0.9 * 100 * 2^11 / fs
    _, _, t1, n1, _ = _servo_restatement(1.0, fs, 2 * L)
    assert abs(t1 - 0.9 * 100 * 2048 / _servo_restatement(1.0, fs, 2 * L)[4]) < 1e-4 * t1
    assert n1 == int(np.ceil(t1 / (L / fs)))


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["--cdsp", "--blocks", "12"], ["--faithful", "--blocks", "3"],
                                  ["--nsig", "21", "--L", "8192", "--blocks", "12"]])
def test_host_demo_on_gpu(host_build, args, tmp_path, synth):
    # config 1 (four.cfg: 1 + 3) and config 2 (URA21.cfg: 1 + 21) through ccoherent::step()
    dump = tmp_path / "block0.bin"
    r = subprocess.run([os.path.join(host_build, "coherent_demo"), "--dump", str(dump)] + args,
                       capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "DEMO OK" in r.stdout
    nsig = int(args[args.index("--nsig") + 1]) if "--nsig" in args else 3
    rows = np.fromfile(dump, dtype=np.int8).reshape(1 + nsig, 16384)
    exp, _ = synth.make_block(nsig, 8192, synth.config_seed(1), 0)
    assert np.array_equal(rows, exp)


@pytest.mark.gpu
def test_host_demo_beamformer_chain_sees_the_aligned_noise_at_broadside(host_build):
    # SURVEY 8 f4 through the host mirror of the beamformer client (cbeamformer.h): 21 aligned channels of the
    # reference noise are one source with an all-ones steering vector -> MUSIC peak at alpha = beta = pi/2
    r = subprocess.run([os.path.join(host_build, "coherent_demo"), "--nsig", "21", "--blocks", "12", "--music"],
                       capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "peak (50, 50) of 100 x 100" in r.stdout and "DEMO OK" in r.stdout


@pytest.mark.gpu
def test_pipelined_cpp_engine_bench(host_build):
    # the batched engine from C++ (ccoherent::submit_batch / collect_batch over crsdr_plan_submit_batch +
    # crsdr_plan_fetch_batch_async): host rows in, host packets out for every block, upload of batch b + 1 under the download
    # of batch b; lags of the last batch checked against the injected delays by the demo itself
    r = subprocess.run([os.path.join(host_build, "coherent_demo"), "--bench", "--nsig", "128", "--batch", "8", "--blocks", "96"],
                       capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "DEMO OK" in r.stdout, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("bench:")][0]
    assert float(line.split("s: ")[1].split(" blocks/s")[0]) > 1000.0       # 129 rows: far above the 1025-row link bound


@pytest.mark.gpu
def test_servo_model_drives_track_to_locked(host_build):
    # ccontrol's loop (descent = 2^-11 tanh(lag/100), hold 0.9 |lag/(p fs)|, sync_threshold) over
    # the modelled resampler: every row ends synchronized with zero residual delay, the engine then runs its
    # locked (phase-only) cadence, and the EMA phasor settles on exp(-j phi)
    r = subprocess.run([os.path.join(host_build, "coherent_demo"), "--servo", "--dmax", "40", "--blocks", "400"],
                       capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "DEMO OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_streaming_devices_ring_and_readcnt_continuity(host_build):
    # producer thread per device -> cbuffer ring (raw offset-binary uint8, XOR fused on the GPU)
    # -> ccoherent thread -> publish loop; per-row readcnt continuous, no ring overruns, lags recovered
    r = subprocess.run([os.path.join(host_build, "coherent_demo"), "--threads", "--blocks", "60", "--pace-ms", "4"],
                       capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "DEMO OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_replay_of_recorded_streams(host_build, synth, tmp_path):
    # a hardware-free block source fed from recordings -- one raw offset-binary uint8 IQ file per channel,
    # the format librtlsdr delivers.  Each device's producer thread replays its file into the ring, the engine runs in
    # its thread, the main thread publishes; the lags the engine reports are the delays the recording was made with.
    nsig, L, nblocks = 5, 8192, 10
    params = synth.RowParams(nsig, L, 4321, dmax=1500)
    blocks = np.stack([synth.make_block(nsig, L, 4321, t, params=params)[0] for t in range(nblocks)])
    for r in range(nsig + 1):
        (blocks[:, r, :].view(np.uint8) ^ np.uint8(0x80)).tofile(tmp_path / f"rec{r}.u8")
    r = subprocess.run([os.path.join(host_build, "coherent_demo"), "--nsig", str(nsig), "--blocks", str(nblocks),
                        "--replay", str(tmp_path / "rec")], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "DEMO OK" in r.stdout, r.stdout + r.stderr
    lags = [int(line.split("lag")[1]) for line in r.stdout.splitlines() if line.startswith("row ") and "lag" in line]
    assert lags == [int(d) for d in params.d]
    assert f"streaming: {nblocks} packets, 0 readcnt gaps" in r.stdout


@pytest.mark.gpu
def test_batched_engine_packets_equal_the_per_block_loop(host_build):
    # src/ccoherent.cc:245-294 a batch at a time: fill_batch keeps every device's OWN read counter for the header
    # (src/cpacketizer.cc:142,163) and the per-device lag gate (src/ccoherent.cc:266); collect_batch calls set_lag for every
    # block and publishes every block's packet.  Against ccoherent::step() on the same synthetic stream, three batches, with
    # one device's counter offset by 3 and one device that stops asking for a lag after the first batch: every message
    # (header, read counters, matrix, zero tail) and every phase-factor payload must be equal bit for bit.
    for extra in (["--nsig", "5", "--batch", "4"], ["--nsig", "3", "--batch", "6", "--faithful"]):
        r = subprocess.run([os.path.join(host_build, "coherent_demo"), "--batch-parity"] + extra, capture_output=True, text=True, timeout=300)
        print(r.stdout, r.stderr)
        assert r.returncode == 0 and "DEMO OK" in r.stdout, r.stdout + r.stderr
        n = 3 * int(extra[3])
        assert f"{n} + {n} packets" in r.stdout and f"{n} with row 2's own read counter" in r.stdout and "lags equal" in r.stdout


@pytest.mark.gpu
def test_batched_streaming_engine_shows_ring_overruns_in_the_header_readcnts(host_build):
    # producer threads -> rings -> the engine thread a batch at a time -> cpacketize::publish.  A reader that is slower than
    # the producers loses blocks in the rings (README.md:42), and the only trace of a lost block is a jump in that row's read
    # counter -- r02's batched engine numbered the rows seq + t and hid it.  Clean run first: no overruns, no jumps.
    exe = os.path.join(host_build, "coherent_demo")
    clean = subprocess.run([exe, "--threads", "--batched", "--nsig", "3", "--blocks", "48", "--batch", "4", "--pace-us", "4000"],
                           capture_output=True, text=True, timeout=300)
    print(clean.stdout, clean.stderr)
    assert clean.returncode == 0 and "DEMO OK" in clean.stdout and " 0 read-counter jumps" in clean.stdout and " 0 ring overruns" in clean.stdout, clean.stdout + clean.stderr
    slow = subprocess.run([exe, "--threads", "--batched", "--nsig", "3", "--blocks", "240", "--batch", "4", "--pace-us", "300", "--engine-delay-ms", "40"],
                          capture_output=True, text=True, timeout=300)
    print(slow.stdout, slow.stderr)
    assert slow.returncode == 0 and "DEMO OK" in slow.stdout, slow.stdout + slow.stderr
    import re
    m = re.search(r"(\d+) packets, (\d+) read-counter jumps \((\d+) blocks skipped\), (\d+) backwards, (\d+) ring overruns", slow.stdout)
    assert m and int(m.group(2)) > 0 and 0 < int(m.group(3)) <= int(m.group(5)) and int(m.group(4)) == 0


@pytest.mark.gpu
def test_cpp_engine_one_process_per_gpu_self_exchange_equals_the_unsharded_engine(host_build, tmp_path):
    # coherent_demo --engine-batches: the C++ host of the multi-GPU run -- ccoherent over a sharded plan (row_begin / row_count),
    # crsdr_exchange_bind_plan / _submit_batch / _fetch_rooted (RCCL under the C ABI), the rank's rooted blocks through
    # cpacketize::publish.  A one-GPU box can run ONE rank (RCCL refuses two ranks on a device): with CRSDR_XCHG_SELF=1 that
    # rank's chunk goes through ncclSend / ncclRecv to itself, so communicator, grouped send / recv, assembly and the engine's
    # bookkeeping all run; its packets and phase-factor payloads must equal the unsharded batched engine's, digest for digest.
    exe = os.path.join(host_build, "coherent_demo")
    args = ["--engine-batches", "--nsig", "6", "--blocks", "24", "--batch", "8"]
    plain = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
    print(plain.stdout[-2000:], plain.stderr[-2000:])
    assert plain.returncode == 0 and "DEMO OK" in plain.stdout, plain.stdout + plain.stderr
    idf = tmp_path / "xid"
    for mode_env in ({"CRSDR_XCHG_SELF": "1"}, {}):          # through the transport, and with the own chunk read straight from the send slots
        if idf.exists():
            idf.unlink()
        shard = subprocess.run([exe] + args + ["--ranks", "1", "--rank", "0", "--device", "0", "--id-file", str(idf)], capture_output=True, text=True, timeout=300,
                               env=dict(os.environ, **mode_env))
        print(shard.stdout[-2000:], shard.stderr[-2000:])
        assert shard.returncode == 0 and "DEMO OK" in shard.stdout, shard.stdout + shard.stderr
        dig = lambda out: sorted(ln for ln in out.splitlines() if ln.startswith("packet seq"))
        assert len(dig(plain.stdout)) == 24 and dig(plain.stdout) == dig(shard.stdout)
