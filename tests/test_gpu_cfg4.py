"""GPU parity at BASELINE config 4 (1 + 1024 rows x 8192) and the boundary of the bit-exact-lag claim.

  (a) the whole configuration on one plan, T = 8 blocks per submit, against the CPU oracle at FULL size (the oracle
      takes ~0.6 s per 1025-row block) plus the size-independent acceptance properties of
      matlabclient/seqnum_and_correlation.m:27-43 (centred peak after alignment, continuous sequence words);
  (b) the shape ONE rank of the 8-GPU run computes: row_begin / row_count = 128 of the 1025 rows, T = 64 blocks per
      submit, dense slab output (crsdr_plan_bind_slab) -- against the oracle on the same slab;
  (c) an SNR sweep at L = 8192: the signal level under the row's own noise is lowered until the correlation peak
      drowns.  Per PAPR bin (PAPR as the reference prints it, |max c|^2 / rms(c)^2 over the 2L-1 lags,
      matlabclient/seqnum_and_correlation.m:40) it records GPU argmax == oracle argmax and == injected delay,
      and asserts the floor DESIGN.md section 2 states.
"""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# above this PAPR (linear; 16 dB) every lag is the injected delay on both sides, bit for bit
PAPR_FLOOR = 40.0


@pytest.fixture(scope="module")
def b():
    binding = importlib.import_module("coherent-rtlsdr_amd.binding")
    if binding.device_count() < 1:
        pytest.fail("no HIP device: the product path has no CPU fallback")
    return binding


def _check_block(got, exp, rows_slice=slice(None), phase_tol=1e-5):
    """HIP path against the fp32 oracle for the rows of rows_slice (same bars as tests/test_gpu_plan.py::_compare)."""
    assert np.array_equal(got["lag"][rows_slice], exp["lag"][rows_slice])
    assert np.allclose(got["mag"][rows_slice], exp["mag"][rows_slice], rtol=1e-4)
    assert np.allclose(got["frac"][rows_slice], exp["frac"][rows_slice], atol=5e-3)
    gp, ep = got["phasor"][rows_slice], exp["phasor"][rows_slice]
    nz = np.abs(ep) > 0
    assert np.abs(np.angle(gp[nz] * np.conj(ep[nz]))).max() <= phase_tol
    assert np.allclose(np.abs(gp), np.abs(ep), rtol=1e-5, atol=1e-7)


def _check_matrix(got_m, exp_m):
    d = got_m.astype(np.int16) - exp_m.astype(np.int16)
    assert np.abs(d).max() <= 1
    assert np.count_nonzero(d) <= 1e-3 * d.size


def test_cfg4_full_size_vs_oracle_and_properties(b, oracle, model, synth):
    nsig, L, T = 1024, 8192, 8
    nrows, B = nsig + 1, 2 * L
    seed = synth.config_seed(4)
    params = synth.RowParams(nsig, L, seed)
    blocks = np.stack([synth.make_block(nsig, L, seed, t, params=params)[0] for t in range(T)])
    readcnt = (np.arange(T * nrows, dtype=np.uint32).reshape(T, nrows) * 3 + 1)
    plan = b.Plan(nrows, B, b.MODE_DIGITAL, max_batch=T)
    plan.submit(blocks, seq=1000, readcnt=readcnt)
    orc = oracle.Engine(nrows, B, oracle.DIGITAL)
    for t in range(T):
        got = plan.fetch(block=t)
        exp = orc.block(blocks[t], seq=1000 + t, readcnt=readcnt[t], nthreads=8)
        assert np.array_equal(got["lag"][1:], params.d), t                   # every lag == injected delay
        _check_block(got, exp)
        _check_matrix(got["matrix"], exp["matrix"])
        # header + readcnt words (matlabclient/zmqsdr.c:118-144), byte for byte
        assert np.array_equal(got["packet"][:plan.matrix_offset], exp["packet"][:orc.matrix_offset])
        hdr = got["packet"][:16].view(np.uint32)
        assert list(hdr) == [1000 + t, nrows, L, 0]
        assert np.array_equal(got["packet"][16:16 + 4 * nrows].view(np.uint32), readcnt[t])
        assert np.array_equal(got["matrix"][0], blocks[t][0])                # row 0 = raw reference row
    # centred peak + residual phase after alignment, 16 sampled rows of the last block
    ref = model.to_complex(got["matrix"][0])
    for r in (1, 2, 3, 64, 128, 129, 255, 256, 257, 511, 512, 640, 777, 900, 1023, 1024):
        y = model.to_complex(got["matrix"][r])
        lag, _, _, _ = model.xcorr_lag(y, ref)
        assert lag == 0, r
        assert abs(np.angle(np.sum(y * np.conj(ref)))) < 0.03, r
    assert np.abs(np.angle(got["phasor"][1:] * np.exp(1j * params.phi))).max() < 0.05   # EMA after 8 blocks: 2^-8 left
    plan.close()


def test_cfg4_per_rank_slab_T64_vs_oracle(b, oracle, synth):
    # rank 3 of 8: signal rows [385, 513) of the 1025, 64 blocks per submit, dense slab output; header / readcnt / row 0
    # only for the 8 blocks this rank assembles (blocks 24..31 of the batch)
    import torch
    nsig, L, T, G, rank = 1024, 8192, 64, 8, 3
    nrows, B = nsig + 1, 2 * L
    per, Tg = nsig // G, T // G
    r0 = 1 + rank * per
    seed = synth.config_seed(4)
    params = synth.RowParams(nsig, L, seed)
    sub = synth.RowParams.__new__(synth.RowParams)
    sub.nsig, sub.L, sub.seed, sub.dmax = per, L, seed, params.dmax
    lo, hi = r0 - 1, r0 - 1 + per
    sub.d, sub.c, sub.s, sub.phi, sub.g = params.d[lo:hi], params.c[lo:hi], params.s[lo:hi], params.phi[lo:hi], params.g[lo:hi]
    dev = torch.device("cuda", 0)
    d_in = torch.zeros((T, nrows, B), dtype=torch.uint8, device=dev)          # rows outside the slab stay zero: never read
    small = []
    for t in range(T):
        part, _ = synth.make_block(per, L, seed, t, params=sub)               # [1 + per][B]: ref + the slab's rows
        small.append(part)
        pt = torch.from_numpy(part.view(np.uint8)).to(dev)
        d_in[t, 0] = pt[0]
        d_in[t, r0:r0 + per] = pt[1:]
    plan = b.Plan(nrows, B, b.MODE_DIGITAL, row_begin=r0, row_count=per, max_batch=T)
    pstride = (plan.packet_bytes + 255) // 256 * 256
    pk = torch.zeros(Tg * pstride + 256, dtype=torch.uint8, device=dev)
    off = (-(pk.data_ptr() + plan.matrix_offset)) % 16
    slab = torch.zeros(T * per * B, dtype=torch.uint8, device=dev)
    plan.bind_packet(pk.data_ptr() + off, pstride)
    plan.bind_slab(slab.data_ptr(), per * B, rank * Tg, Tg)
    plan.submit(d_in.data_ptr(), seq=500, nblocks=T, block_stride=nrows * B)
    plan.sync()
    slab_h = slab.cpu().numpy().view(np.int8).reshape(T, per, B)
    pk_h = pk.cpu().numpy()
    orc = oracle.Engine(1 + per, B, oracle.DIGITAL)
    own = slice(r0, r0 + per)
    for t in range(T):
        exp = orc.block(small[t], seq=500 + t, nthreads=8)
        got = plan.fetch(want_packet=False, block=t)
        assert np.array_equal(got["lag"][own], params.d[lo:hi]), t
        sel = {k: got[k][own] for k in ("lag", "mag", "frac", "phasor")}
        ex1 = {k: exp[k][1:] for k in ("lag", "mag", "frac", "phasor")}
        _check_block(sel, ex1)
        _check_matrix(slab_h[t], exp["matrix"][1:])
        if rank * Tg <= t < (rank + 1) * Tg:                                  # a block this rank assembles
            p = pk_h[off + (t - rank * Tg) * pstride:][:plan.packet_bytes]
            assert list(p[:16].view(np.uint32)) == [500 + t, nrows, L, 0]
            assert np.all(p[16:16 + 4 * nrows].view(np.uint32) == 500 + t)
            assert np.array_equal(p[plan.matrix_offset:plan.matrix_offset + B].view(np.int8), small[t][0])
    plan.close()


def _papr(row_i8, ref_i8):
    """PAPR of matlabclient/seqnum_and_correlation.m:36-40: c = xcorr(ref, row) over the 2L-1 lags, |max c|^2 / rms(c)^2."""
    x = row_i8[0::2].astype(np.float64) + 1j * row_i8[1::2]
    r = ref_i8[0::2].astype(np.float64) + 1j * ref_i8[1::2]
    n = 2 * x.size
    c = np.fft.ifft(np.fft.fft(x, n) * np.conj(np.fft.fft(r, n)))
    c = np.concatenate([c[-(x.size - 1):], c[:x.size]])          # lags -(L-1) .. L-1
    p = np.abs(c) ** 2
    return p.max() / p.mean()


def test_snr_sweep_states_the_papr_floor_for_bit_exact_lags(b, oracle, synth, capsys):
    # rows = g * delayed rotated ref + the row's own sigma-10 noise, g swept from the generator's 0.5..1 down to where the
    # peak is lost (signal ~0.1 LSB under 10 LSB of noise: it survives the int8 quantiser as dither).  72 gain steps x 28
    # rows per step; the rows of a step share g and differ in delay, phase and noise.
    L, per = 8192, 28
    gains = np.concatenate([np.geomspace(1.0, 0.05, 12), np.geomspace(0.04, 0.004, 60)])
    nsig = per * gains.size
    seed = 0xFACE
    params = synth.RowParams(nsig, L, seed)
    params.g = np.repeat(gains, per)
    nrows, B = nsig + 1, 2 * L
    plan, orc = b.Plan(nrows, B, b.MODE_DIGITAL), oracle.Engine(nrows, B, oracle.DIGITAL)
    papr, same, hit = [], [], []
    for t in range(2):
        rows, _ = synth.make_block(nsig, L, seed, t, params=params)
        got, exp = plan.block(rows, seq=t), orc.block(rows, seq=t, nthreads=8)
        papr += [_papr(rows[k], rows[0]) for k in range(1, nrows)]
        same += list(got["lag"][1:] == exp["lag"][1:])
        hit += list(got["lag"][1:] == params.d)
    papr, same, hit = np.array(papr), np.array(same), np.array(hit)
    edges = [0, 12, 16, 20, 25, 32, PAPR_FLOOR, 64, 128, 1e9]
    lines = []
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (papr >= lo) & (papr < hi)
        lines.append(f"PAPR [{lo:g}, {hi:g}): rows {m.sum():4d}  gpu==oracle {same[m].sum():4d}  gpu==injected {hit[m].sum():4d}")
    with capsys.disabled():
        print("\n" + "\n".join(lines))
    above = papr >= PAPR_FLOOR
    assert above.sum() >= 300 and (~above).sum() >= 300          # the sweep really straddles the floor
    assert same[above].all() and hit[above].all()                # the claim: bit-exact and correct above the floor
    assert (~hit[papr < 16]).sum() >= 100                        # and the sweep does reach rows whose peak is lost
    # below the floor the argmax is a noise peak; both sides still pick the same one unless two peaks tie within fp32
    # rounding of two different FFT factorisations (expected ~1e-4 of such rows)
    assert same.mean() >= 0.995
    plan.close()
