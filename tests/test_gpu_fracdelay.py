"""GPU: fractional-delay correction of the long-block path (BASELINE config 5: "long-block HBM-streaming regime +
fractional-delay phase correction"; SURVEY 8 note "Fractional delay": reported as `frac` and, if applied, applied as a
linear phase ramp in the frequency domain).  crsdr_plan_set_frac_apply is opt-in -- the reference computes a fractional
estimate and discards it (src/ccoherent.cc:206-219) -- and is checked against the oracle and the fp64 model, which apply the
same ramp, plus a known answer: a band-limited row delayed by a non-integer number of samples comes out aligned
(correlation peak centred, no residual phase ramp across the band), which the integer shift alone cannot do.
Bars: int8 matrix equal to the oracle / model except +-1 LSB at rounding boundaries (the two sides run different fp32 FFT
factorisations; v_sin / v_cos against libm), lags and phasors as everywhere else."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def b():
    binding = importlib.import_module("coherent-rtlsdr_amd.binding")
    if binding.device_count() < 1:
        pytest.fail("no HIP device: the product path has no CPU fallback")
    return binding


def _matrix_close(got, exp, frac_allowed):
    d = got.astype(np.int16) - exp.astype(np.int16)
    assert np.abs(d).max() <= 1
    assert np.count_nonzero(d) <= frac_allowed * d.size, (np.count_nonzero(d), d.size)


@pytest.mark.parametrize("log2B", [15, 16, 17, 18, 20])
def test_frac_apply_matches_oracle_and_fp64_model(b, oracle, model, synth, log2B):
    nsig, B = 4, 1 << log2B
    L = B // 2
    seed = 900 + log2B
    params = synth.RowParams(nsig, L, seed)
    D = np.array([0.0, 0.5, -0.25, 0.3, 0.0], dtype=np.float32)               # entry 0 (the reference row) is ignored
    plan, orc, mod = b.Plan(nsig + 1, B, b.MODE_DIGITAL), oracle.Engine(nsig + 1, B, oracle.DIGITAL), model.Model(nsig + 1, B, model.DIGITAL)
    for x in (plan, orc, mod):
        x.set_frac_apply(True, 1.0, D)
    for t in range(2):
        rows, _ = synth.make_block(nsig, L, seed, t, params=params)
        got, exp = plan.block(rows, seq=t), orc.block(rows, seq=t)
        assert np.array_equal(got["lag"], exp["lag"]) and np.array_equal(got["lag"][1:], params.d)
        # (the GPU's dot product is exact in integers; the fp32 oracle's sequential accumulator carries ~1e-5 rad of its own
        # from 2^19 samples per row on: 1.1e-5 seen at B = 2^20)
        assert np.abs(np.angle(got["phasor"][1:] * np.conj(exp["phasor"][1:]))).max() <= (1e-5 if log2B <= 18 else 3e-5)
        _matrix_close(got["matrix"], exp["matrix"], 2e-3)
        _matrix_close(got["matrix"], mod.block(rows)[4], 2e-3)
        assert np.array_equal(got["matrix"][0], rows[0])                       # raw reference row, header as ever
        assert np.array_equal(got["packet"][:plan.matrix_offset], exp["packet"][:orc.matrix_offset])
    # row 4 has D = 0: the frequency-domain path reproduces the integer shift of the digital mode (zero-filled) to rounding
    plain = b.Plan(nsig + 1, B, b.MODE_DIGITAL)
    for t in range(2):
        rows, _ = synth.make_block(nsig, L, seed, t, params=params)
        ref_out = plain.block(rows, seq=t)
    _matrix_close(got["matrix"][4:5], ref_out["matrix"][4:5], 2e-3)
    # estimate-driven (no override): D = gain * frac of the same block on both sides; the two frac estimates differ by ~1e-3
    # of a sample, so a few more entries sit on the other side of a rounding boundary
    plan.set_frac_apply(True, 2.0, None)
    orc.set_frac_apply(True, 2.0, None)
    rows, _ = synth.make_block(nsig, L, seed, 2, params=params)
    got, exp = plan.block(rows, seq=2), orc.block(rows, seq=2)
    assert np.allclose(got["frac"], exp["frac"], atol=5e-3)
    _matrix_close(got["matrix"], exp["matrix"], 3e-2)
    # and off again: back to the plain digital mode
    plan.set_frac_apply(False)
    rows, _ = synth.make_block(nsig, L, seed, 3, params=params)
    got = plan.block(rows, seq=3)
    exp = plain.block(rows, seq=3)
    assert np.array_equal(got["lag"], exp["lag"])
    plan.close(); plain.close()


def _bandlimited_rows(L, total_delays, phis, seed=3, band=0.25, sigma=30.0):
    """ref = low-pass complex Gaussian noise (|f| < band * fs / 2); row k = ref delayed by total_delays[k] samples
    (any real number: the delay is a phase ramp on a 4L-point spectrum of a longer realisation, so nothing wraps into
    the block) and rotated by phis[k]; int8, round half even."""
    rng = np.random.default_rng(seed)
    n = 4 * L
    X = np.fft.fft(rng.standard_normal(n) + 1j * rng.standard_normal(n))
    f = np.fft.fftfreq(n)
    X[np.abs(f) > band / 2] = 0
    base = np.fft.ifft(X)
    base *= sigma / np.sqrt(np.mean(np.abs(base) ** 2) / 2)

    def q(x):
        out = np.empty(2 * L, dtype=np.int8)
        out[0::2] = np.clip(np.rint(x.real), -128, 127)
        out[1::2] = np.clip(np.rint(x.imag), -128, 127)
        return out

    rows = np.zeros((1 + len(total_delays), 2 * L), dtype=np.int8)
    rows[0] = q(base[L:2 * L])
    for k, (tau, phi) in enumerate(zip(total_delays, phis)):
        xs = np.fft.ifft(X * np.exp(-2j * np.pi * f * tau)) * (sigma / np.sqrt(np.mean(np.abs(np.fft.ifft(X)) ** 2) / 2))
        rows[1 + k] = q(xs[L:2 * L] * np.exp(1j * phi))
    return rows


def _residual_delay(y, r, band=0.25):
    """slope of the cross-spectrum phase over the occupied band, in samples (0 = aligned)"""
    n = y.size
    win = np.hanning(n)
    Y, R = np.fft.fft(y * win), np.fft.fft(r * win)
    f = np.fft.fftfreq(n)
    m = np.abs(f) < 0.8 * band / 2
    c = Y[m] * np.conj(R[m])
    ph = np.angle(c * np.exp(-1j * np.angle(np.sum(c))))
    w = np.abs(c)
    slope = np.sum(w * f[m] * ph) / np.sum(w * f[m] ** 2)      # phase = -2 pi f tau
    return -slope / (2 * np.pi)


@pytest.mark.parametrize("L", [1 << 15, 1 << 13, 1 << 12])
def test_non_integer_delays_come_out_aligned_with_the_ramp_applied(b, model, L):
    # known answer: rows delayed by d + delta samples (delta = +-0.5: the half-sample case, 0.25, -0.4).  The integer lag is
    # whatever the correlation peak says (d or d + 1 for a half-sample delay); the caller supplies the remainder
    # (matlabclient/notes.m:9-40: the parabolic estimate needs a signal-dependent calibration before it can be consumed).
    # L = 2^15: the long blocks' second four-step pass; 2^13 (B = 16384, the reference's block size): the pass on K1's network;
    # 2^12: the generic LDS kernel -- an answer that does not come from the oracle.
    total = [37.5, -120.5, 300.25, -7.4]
    phis = [0.4, -1.1, 2.5, 0.0]
    rows = _bandlimited_rows(L, total, phis)
    plan = b.Plan(5, 2 * L, b.MODE_DIGITAL)
    out = plan.block(rows)
    lag = out["lag"].astype(np.float64)
    assert np.all(np.abs(lag[1:] - np.array(total)) <= 0.5 + 1e-9)              # the integer part, as reported
    ref = model.to_complex(rows[0])
    before = [_residual_delay(model.to_complex(out["matrix"][1 + k]), ref) for k in range(4)]
    rem = np.zeros(5, dtype=np.float32)
    rem[1:] = np.array(total) - lag[1:]
    plan.set_frac_apply(True, 1.0, rem)
    for t in range(8):                                                          # EMA phasor settles
        out = plan.block(rows, seq=1 + t)
    for k in range(4):
        y = model.to_complex(out["matrix"][1 + k])
        after = _residual_delay(y, ref)
        assert abs(before[k] - rem[1 + k]) < 0.05, (k, before[k], rem[1 + k])   # the integer shift leaves the fraction in
        assert abs(after) < 0.02, (k, after)                                     # the ramp takes it out
        lg, _, fr, _ = model.xcorr_lag(y, ref)
        assert lg == 0 and abs(fr) < 0.03, (k, lg, fr)                           # centred peak, symmetric neighbours
        assert abs(np.angle(np.sum(y * np.conj(ref)))) < 0.03                    # and the phase is out as well
    plan.close()


def test_cfg5_full_size_with_fractional_delays(b, oracle, synth):
    # BASELINE config 5 at full size: 1 + 21 rows x 2^20 samples, every row with its own fractional delay
    nsig, L = 21, 1 << 20
    seed = synth.config_seed(5)
    params = synth.RowParams(nsig, L, seed)
    rng = np.random.default_rng(5)
    D = np.concatenate([[0.0], rng.uniform(-0.5, 0.5, nsig)]).astype(np.float32)
    rows, _ = synth.make_block(nsig, L, seed, 0, params=params)
    plan, orc = b.Plan(nsig + 1, 2 * L, b.MODE_DIGITAL), oracle.Engine(nsig + 1, 2 * L, oracle.DIGITAL)
    plan.set_frac_apply(True, 1.0, D)
    orc.set_frac_apply(True, 1.0, D)
    got, exp = plan.block(rows), orc.block(rows, nthreads=8)
    assert np.array_equal(got["lag"][1:], params.d)
    _matrix_close(got["matrix"], exp["matrix"], 2e-3)
    plan.close()


@pytest.mark.parametrize("log2B", [6, 10, 13, 14])
def test_frac_apply_on_lds_resident_blocks(b, oracle, model, synth, log2B):
    # The correction at the reference's own block size (B = 16384) and below: one kernel behind the phase kernels (int8 row ->
    # forward transform -> x exp(2 pi i f_s (lag + D) / B) . p / B -> inverse -> int8) instead of the long blocks' second four-step
    # pass.  Same definition, so the same checks as test_frac_apply_matches_oracle_and_fp64_model: matrix equal to the oracle and
    # to the fp64 model except +-1 LSB on a few entries per thousand, the raw reference row and the header untouched, batched
    # submits equal to block-by-block submits bit for bit, a row with D = 0 equal to the plain digital mode to rounding.
    nsig, B = 5, 1 << log2B
    L = B // 2
    seed = 700 + log2B
    params = synth.RowParams(nsig, L, seed, dmax=max(1, L // 8))
    D = np.array([0.0, 0.5, -0.25, 0.3, 0.0, -0.45], dtype=np.float32)
    T = 3
    plan, orc, mod = b.Plan(nsig + 1, B, b.MODE_DIGITAL, max_batch=T), oracle.Engine(nsig + 1, B, oracle.DIGITAL), model.Model(nsig + 1, B, model.DIGITAL)
    for x in (plan, orc, mod):
        x.set_frac_apply(True, 1.0, D)
    blocks = [synth.make_block(nsig, L, seed, t, params=params)[0] for t in range(2 * T)]
    single = []
    for t in range(T):
        got, exp = plan.block(blocks[t], seq=t), orc.block(blocks[t], seq=t)
        single.append(got)
        assert np.array_equal(got["lag"], exp["lag"])
        assert np.abs(np.angle(got["phasor"][1:] * np.conj(exp["phasor"][1:]))).max() <= 1e-5
        # short rows have few samples per quantisation boundary: allow the same fraction, at least a handful of entries
        frac_allowed = max(2e-3, 4.0 / got["matrix"].size)
        _matrix_close(got["matrix"], exp["matrix"], frac_allowed)
        _matrix_close(got["matrix"], mod.block(blocks[t])[4], frac_allowed)
        assert np.array_equal(got["matrix"][0], blocks[t][0])
        assert np.array_equal(got["packet"][:plan.matrix_offset], exp["packet"][:orc.matrix_offset])
    # one batched submit of the same three blocks on a fresh plan: bit for bit what the three single submits gave
    plan2 = b.Plan(nsig + 1, B, b.MODE_DIGITAL, max_batch=T)
    plan2.set_frac_apply(True, 1.0, D)
    plan2.submit(np.stack(blocks[:T]), seq=0)
    for t in range(T):
        out = plan2.fetch(block=t)
        for key in ("lag", "mag", "frac", "phasor", "packet"):
            assert np.array_equal(out[key].view(np.uint8), single[t][key].view(np.uint8)), (t, key)
    # rows 4 has D = 0: the frequency-domain path reproduces the zero-filled integer shift of the plain digital mode to rounding
    plain = b.Plan(nsig + 1, B, b.MODE_DIGITAL, max_batch=T)
    for t in range(T):
        ref_out = plain.block(blocks[t], seq=t)
    _matrix_close(single[T - 1]["matrix"][4:5], ref_out["matrix"][4:5], max(2e-3, 4.0 / B))
    # estimate-driven, then off again
    plan.set_frac_apply(True, 1.0, None)
    orc.set_frac_apply(True, 1.0, None)
    got, exp = plan.block(blocks[T], seq=T), orc.block(blocks[T], seq=T)
    assert np.allclose(got["frac"], exp["frac"], atol=5e-3)
    _matrix_close(got["matrix"], exp["matrix"], 3e-2)
    plan.set_frac_apply(False)
    plain.block(blocks[T], seq=T)
    got, exp = plan.block(blocks[T + 1], seq=T + 1), plain.block(blocks[T + 1], seq=T + 1)
    assert np.array_equal(got["lag"], exp["lag"])
    for p_ in (plan, plan2, plain):
        p_.close()


def test_frac_apply_argument_checks(b):
    short = b.Plan(3, 16384, b.MODE_DIGITAL)
    short.set_frac_apply(True)                        # LDS-resident blocks: offered since r03 (one kernel behind the phase kernels)
    short.set_frac_apply(False)                       # switching it off is always fine
    faithful = b.Plan(3, 1 << 15, b.MODE_FAITHFUL)
    with pytest.raises(b.CrsdrError):
        faithful.set_frac_apply(True)                 # the faithful mode never shifts samples
    longp = b.Plan(3, 1 << 15, b.MODE_DIGITAL)
    with pytest.raises(b.CrsdrError):
        longp.set_frac_apply(True, 1000.0)            # |gain| <= 128: D = gain * frac stays where the fp32 phase keeps 1e-5 rad
    with pytest.raises(b.CrsdrError):
        longp.set_frac_apply(True, 1.0, np.array([0.0, 0.5, 200.0], dtype=np.float32))   # |D| <= 64 samples: the rest belongs in the lag
    with pytest.raises(b.CrsdrError):
        longp.set_frac_apply(True, 1.0, np.array([0.0, np.nan, 0.0], dtype=np.float32))
    with pytest.raises(b.CrsdrError):
        longp.set_frac_apply(3)                       # enable: 0, 1 or 2
    longp.set_frac_apply(True, 1.0, np.array([7.0, 64.0, -64.0], dtype=np.float32))      # entry 0 (the reference row) is not looked at
    short.close(); faithful.close(); longp.close()


def test_frac_apply_without_second_work_area_is_bit_identical(b, synth):
    # crsdr_plan_set_frac_apply(enable = 1) keeps the correlation pass's first stage in a second cf32 work area (cfg5: 352 MB) so
    # that the correction pass does not repeat it; when that buffer cannot be had -- or is not wanted: enable = 2 -- the pass runs
    # its first stage again.  Same arithmetic on the same inputs: every output must agree bit for bit.  cfg5's shape, because
    # only launches of >= 1024 lines take the two-line stage-B kernel that writes out of place.  Switching the correction off
    # gives the buffers back.
    import torch
    nsig, L = 21, 1 << 20
    params = synth.RowParams(nsig, L, 4242, dmax=L // 8)
    D = np.linspace(-0.5, 0.5, nsig + 1).astype(np.float32)
    rows = [synth.make_block(nsig, L, 4242, t, params=params)[0] for t in range(2)]
    torch.cuda.synchronize()
    outs, held = {}, {}
    for enable in (1, 2):
        plan = b.Plan(nsig + 1, 2 * L, b.MODE_DIGITAL)
        free0, _ = torch.cuda.mem_get_info()
        plan.set_frac_apply(enable, 1.0, D)
        free1, _ = torch.cuda.mem_get_info()
        held[enable] = free0 - free1
        outs[enable] = [plan.block(r, seq=t) for t, r in enumerate(rows)]
        plan.set_frac_apply(False)
        free2, _ = torch.cuda.mem_get_info()
        assert free2 >= free0 - (8 << 20), (enable, free0, free2)            # the buffers set_frac_apply allocated are freed again
        plan.close()
    assert held[1] - held[2] >= 8 * 2 * L * nsig - (16 << 20), held          # enable = 1 really held the second work area, enable = 2 did not
    for t in range(2):
        for key in ("lag", "mag", "frac", "phasor", "packet"):
            assert np.array_equal(outs[1][t][key].view(np.uint8), outs[2][t][key].view(np.uint8)), (t, key)
        assert np.array_equal(outs[1][t]["lag"][1:], params.d)


def test_frac_apply_at_16384_both_kernels_and_bound_packets(tmp_path):
    # B = 16384 runs the correction on K1's 32 x 32 x 16 network (k_frac_apply14: junction with the response, natural-order bytes through
    # the LDS image); CRSDR_FRAC_GENERIC=1 keeps the generic radix-16 kernel every other size uses.  Same definition, different
    # factorisation: the two agree except +-1 LSB on a few entries per thousand -- on the plan's own packets, on a caller's packet whose
    # matrix is only 4-byte aligned (dword stores) and through the three-kernel phase path.  Children: the switch is read per plan
    # from the environment of the process.
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent('''
        import importlib, sys, numpy as np, torch
        sys.path.insert(0, %r)
        pkg = importlib.import_module("coherent-rtlsdr_amd")
        b, synth = pkg.binding, pkg.synth
        nsig, L, T = 9, 8192, 3
        nrows, B = nsig + 1, 2 * L
        params = synth.RowParams(nsig, L, 733, dmax=900)
        D = np.linspace(-0.5, 0.5, nrows).astype(np.float32)
        blocks = np.stack([synth.make_block(nsig, L, 733, t, params=params)[0] for t in range(T)])
        res = {}
        plan = b.Plan(nrows, B, b.MODE_DIGITAL, max_batch=T)
        plan.set_frac_apply(True, 1.0, D)
        plan.submit(blocks, seq=0)
        res["own"] = np.stack([plan.fetch(block=t)["packet"] for t in range(T)])
        dev = torch.device("cuda", 0)
        pstride = (plan.packet_bytes + 255) // 256 * 256
        buf = torch.zeros(T * pstride + 64, dtype=torch.uint8, device=dev)
        off = (-buf.data_ptr()) %% 256                                  # packets 256-byte aligned: the matrix at +16 + 4 N is 4-byte aligned only
        assert (buf.data_ptr() + off + plan.matrix_offset) %% 16 != 0
        plan2 = b.Plan(nrows, B, b.MODE_DIGITAL, max_batch=T)
        plan2.set_frac_apply(True, 1.0, D)
        plan2.bind_packet(buf.data_ptr() + off, pstride)
        plan2.submit(blocks, seq=0)
        plan2.sync()
        res["bound"] = np.stack([buf[off + t * pstride: off + t * pstride + plan.packet_bytes].cpu().numpy().view(np.int8) for t in range(T)])
        np.savez(sys.argv[1], **res)
    ''') % root
    outs = {}
    for name, env in (("fast", {}), ("generic", {"CRSDR_FRAC_GENERIC": "1"}), ("fast_three", {"CRSDR_K2_FUSED": "0"})):
        out = tmp_path / f"{name}.npz"
        r = subprocess.run([sys.executable, "-c", code, str(out)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs[name] = np.load(out)
    moff = 16 + 4 * 10
    for name in ("fast", "generic", "fast_three"):
        # the 4-byte aligned bound packet holds what the plan's own packet holds, bit for bit
        assert np.array_equal(outs[name]["own"], outs[name]["bound"]), name
    assert np.array_equal(outs["fast"]["own"], outs["fast_three"]["own"])
    a, g = outs["fast"]["own"][:, moff:].astype(np.int32), outs["generic"]["own"][:, moff:].astype(np.int32)
    diff = np.abs(a - g)
    assert diff.max() <= 1 and np.count_nonzero(diff) <= 2e-3 * diff.size, (diff.max(), np.count_nonzero(diff), diff.size)
    assert np.array_equal(outs["fast"]["own"][:, :moff], outs["generic"]["own"][:, :moff])          # header + read counters
