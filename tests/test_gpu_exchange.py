"""GPU: the multi-GPU exchange of SURVEY 8e on ONE device.

  * G sharded plans in slot mode (crsdr_plan_bind_slab_ex: owned rows + the 24 B/row tail {lag, mag, frac, phasor, readcnt}); the
    transport is emulated by executing every rank's crsdr_exchange_schedule with device copies (sends paired with the
    peer's receives first in first out, as RCCL pairs them); crsdr_assemble_slots then builds the packets and the scalars
    blocks.  Every assembled packet must equal the packet of an unsharded plan bit for bit, every scalars block its
    lag / mag / frac / phasor -- full and ragged batches, two batches in a row (carried state).
  * the RCCL transport itself (crsdr_exchange_create / crsdr_exchange_batch, librccl through dlopen) with one rank: RCCL
    refuses two ranks on one GPU, so the N > 1 wiring is covered by the schedule simulation (tests/test_exchange_schedule.py)
    and this test covers the plumbing -- communicator set-up, grouped ncclSend / ncclRecv on a caller stream (the rank's own
    chunk is routed through RCCL to itself with CRSDR_XCHG_SELF=1), staged and in-place assembly.
"""
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


@pytest.mark.parametrize("G,T", [(2, 6), (4, 8), (4, 6), (8, 4), (8, 20)])
def test_slots_with_tails_through_the_schedule_equal_the_single_plan(b, synth, G, T):
    import torch
    nsig, L = 16, 1024
    nrows, B = nsig + 1, 2 * L
    geo = b.exchange_geometry(nrows, B, G)
    per, slot, toff, sstride = geo["per"], geo["slot_stride"], geo["tail_offset"], geo["scalars_stride"]
    dev = torch.device("cuda", 0)
    params = synth.RowParams(nsig, L, 177, dmax=200)
    blocks = np.stack([synth.make_block(nsig, L, 177, t, params=params)[0] for t in range(2 * T)])
    full = b.Plan(nrows, B, b.MODE_DIGITAL, max_batch=T)
    plans = [b.Plan(nrows, B, b.MODE_DIGITAL, row_begin=1 + r * per, row_count=per, max_batch=T) for r in range(G)]
    pbytes = full.packet_bytes
    pstride = (pbytes + 255) // 256 * 256
    roots = [b.rooted_blocks(T, G, q) for q in range(G)]
    bpr = max(len(r) for r in roots)
    send = [torch.zeros(T * slot, dtype=torch.uint8, device=dev) for _ in range(G)]
    recv = [torch.zeros(G * bpr * slot, dtype=torch.uint8, device=dev) for _ in range(G)]
    pk = [torch.zeros(bpr * pstride + 64, dtype=torch.uint8, device=dev) for _ in range(G)]
    sc = [torch.full((bpr * sstride,), 0xAB, dtype=torch.uint8, device=dev) for _ in range(G)]
    off = [(-(p.data_ptr() + full.matrix_offset)) % 16 for p in pk]
    ops = [b.exchange_schedule(G, q, T, b.XCHG_STAGED, nrows, B, pstride) for q in range(G)]
    for half in range(2):
        d_in = torch.from_numpy(blocks[half * T:(half + 1) * T].view(np.uint8)).to(dev)
        # per-device read counters (src/cpacketizer.cc:142,163): the unsharded plan is given all of them; a sharded plan only those of
        # row 0 and of the rows it owns -- garbage for the rest, as in a process that reads only its own dongles -- in the first half,
        # none at all (seq + t everywhere) in the second.  The assembled headers must equal the unsharded plan's either way.
        rc_true = (1000 * half + 7 * np.arange(T)[:, None] + 13 * np.arange(nrows)[None, :]).astype(np.uint32) if half == 0 else None
        full.submit(d_in.data_ptr(), readcnt=rc_true, seq=10 + half * T, nblocks=T, block_stride=nrows * B)
        exp = [full.fetch(block=t) for t in range(T)]
        for r, pl in enumerate(plans):
            rc_r = None
            if rc_true is not None:
                rc_r = np.full_like(rc_true, 0xDEAD0000 + r)
                rc_r[:, 0] = rc_true[:, 0]
                rc_r[:, 1 + r * per: 1 + (r + 1) * per] = rc_true[:, 1 + r * per: 1 + (r + 1) * per]
            pl.bind_packet(pk[r].data_ptr() + off[r], pstride)
            pl.bind_slab_ex(send[r].data_ptr(), slot, roots[r].start, len(roots[r]), toff)
            pl.submit(d_in.data_ptr(), readcnt=rc_r, seq=10 + half * T, nblocks=T, block_stride=nrows * B)
            pl.sync()
        for a in range(G):                                    # the transport: a's sends to c meet c's receives from a in order
            for c in range(G):
                sends = [o for o in ops[a] if o["peer"] == c and not o["is_recv"]]
                recvs = [o for o in ops[c] if o["peer"] == a and o["is_recv"]]
                assert len(sends) == len(recvs)
                for s, r in zip(sends, recvs):
                    assert s["bytes"] == r["bytes"] and r["buffer"] == 1
                    recv[c][r["offset"]: r["offset"] + r["bytes"]] = send[a][s["offset"]: s["offset"] + s["bytes"]]
        for q in range(G):
            cnt = len(roots[q])
            if cnt == 0:
                continue
            self_ptr = send[q].data_ptr() + roots[q].start * slot
            want_scalars = not (G == 4 and T == 6)             # one shape without a scalars block: the read counters still reach the headers
            b.assemble_slots(pk[q].data_ptr() + off[q], pstride, sc[q].data_ptr() if want_scalars else None, sstride, nrows, B, recv[q].data_ptr(), G, cnt,
                             slot, toff, self_rank=q, self_ptr=self_ptr)
            torch.cuda.synchronize()
            for j, t in enumerate(roots[q]):
                got = pk[q][off[q] + j * pstride: off[q] + j * pstride + pbytes].cpu().numpy().view(np.int8)
                assert np.array_equal(got, exp[t]["packet"]), (half, q, j)
                if not want_scalars:
                    continue
                s = b.parse_scalars(sc[q][j * sstride: (j + 1) * sstride].cpu().numpy(), nrows)
                for k in ("lag", "mag", "frac", "phasor"):
                    assert np.array_equal(s[k].view(np.uint8), exp[t][k].view(np.uint8)), (half, q, j, k)
    # argument checks of the slot binding
    with pytest.raises(b.CrsdrError):
        plans[0].bind_slab_ex(send[0].data_ptr(), slot, 0, 1, toff - 4)         # tail inside the rows
    with pytest.raises(b.CrsdrError):
        plans[0].bind_slab_ex(send[0].data_ptr(), per * B + 8, 0, 1, per * B)   # tail does not fit the slot
    for pl in plans + [full]:
        pl.close()


def test_rccl_exchange_one_rank_staged_and_in_place(tmp_path):
    # a child per setting: CRSDR_XCHG_SELF is read when the exchange is created, and RCCL is loaded once per process
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent('''
        import importlib, sys, numpy as np, torch
        sys.path.insert(0, %r)
        pkg = importlib.import_module("coherent-rtlsdr_amd")
        b, synth = pkg.binding, pkg.synth
        nsig, L, T = 24, 8192, 5
        nrows, B = nsig + 1, 2 * L
        geo = b.exchange_geometry(nrows, B, 1)
        slot, toff, sstride = geo["slot_stride"], geo["tail_offset"], geo["scalars_stride"]
        dev = torch.device("cuda", 0)
        params = synth.RowParams(nsig, L, 41, dmax=500)
        blocks = np.stack([synth.make_block(nsig, L, 41, t, params=params)[0] for t in range(T)])
        d_in = torch.from_numpy(blocks.view(np.uint8)).to(dev)
        ref = b.Plan(nrows, B, b.MODE_DIGITAL, max_batch=T)
        ref.submit(d_in.data_ptr(), seq=7, nblocks=T, block_stride=nrows * B)
        exp = [ref.fetch(block=t) for t in range(T)]
        x = b.Exchange(b.exchange_unique_id(), 1, 0, 0)
        stream = torch.cuda.Stream(device=dev)
        pstride = (ref.packet_bytes + 255) // 256 * 256
        for mode in (b.XCHG_STAGED, b.XCHG_INPLACE):
            plan = b.Plan(nrows, B, b.MODE_DIGITAL, max_batch=T)
            plan.set_stream(stream.cuda_stream)
            send = torch.zeros(T * slot, dtype=torch.uint8, device=dev)
            recv = torch.zeros(T * slot, dtype=torch.uint8, device=dev)
            pk = torch.zeros(T * pstride + 64, dtype=torch.uint8, device=dev)
            sc = torch.full((T * sstride,), 0xCD, dtype=torch.uint8, device=dev)
            off = (-(pk.data_ptr() + ref.matrix_offset)) %% 16
            plan.bind_packet(pk.data_ptr() + off, pstride)
            plan.bind_slab_ex(send.data_ptr(), slot, 0, T, toff)
            plan.submit(d_in.data_ptr(), seq=7, nblocks=T, block_stride=nrows * B)
            x.batch(mode, send.data_ptr(), recv.data_ptr() if mode == b.XCHG_STAGED else None, T, pk.data_ptr() + off, pstride,
                    sc.data_ptr(), sstride, nrows, B, stream.cuda_stream)
            stream.synchronize()
            for t in range(T):
                got = pk[off + t * pstride: off + t * pstride + ref.packet_bytes].cpu().numpy().view(np.int8)
                assert np.array_equal(got, exp[t]["packet"]), (mode, t)
                s = b.parse_scalars(sc[t * sstride: (t + 1) * sstride].cpu().numpy(), nrows)
                for k in ("lag", "mag", "frac", "phasor"):
                    assert np.array_equal(s[k].view(np.uint8), exp[t][k].view(np.uint8)), (mode, t, k)
            plan.close()
        x.close()
        print("EXCHANGE OK")
    ''') % root
    for env in ({"CRSDR_XCHG_SELF": "1"}, {}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "EXCHANGE OK" in r.stdout, (env, r.stdout[-2000:], r.stderr[-4000:])


def test_engine_api_one_rank_equals_the_unsharded_plan(tmp_path):
    # crsdr_exchange_bind_plan / _submit_batch / _fetch_rooted from Python (the C++ host drives the same calls, tests/test_host_cpp.py):
    # host rows in, this rank's assembled packets + scalars + its own rows' tails out, two batches outstanding, a ragged last batch,
    # read counters and a masked row -- against an unsharded plan fed the same batches.  Then the state errors: a third outstanding
    # batch, a fetch with nothing submitted, a second bind, a plan of the wrong shape.  A child per transport setting (self-loop through
    # ncclSend / ncclRecv, and the local copy).
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent('''
        import importlib, sys, numpy as np
        sys.path.insert(0, %r)
        pkg = importlib.import_module("coherent-rtlsdr_amd")
        b, synth = pkg.binding, pkg.synth
        nsig, L, T = 24, 8192, 5
        nrows, B = nsig + 1, 2 * L
        params = synth.RowParams(nsig, L, 43, dmax=500)
        blocks = np.stack([synth.make_block(nsig, L, 43, t, params=params)[0] for t in range(3 * T)])
        rc = (100 + 5 * np.arange(3 * T)[:, None] + np.arange(nrows)[None, :]).astype(np.uint32)
        mask = np.ones(nrows, dtype=np.uint8); mask[4] = 0                     # row 4 never asks for a lag: it keeps its carried one
        sizes = [T, T, 3]                                                       # the last batch is short
        ref = b.Plan(nrows, B, b.MODE_DIGITAL, max_batch=T)
        exp, at = [], 0
        for nb in sizes:
            ref.submit(blocks[at:at + nb], readcnt=rc[at:at + nb], lag_mask=mask, seq=at)
            exp.append([ref.fetch(block=t) for t in range(nb)])
            at += nb
        x = b.Exchange(b.exchange_unique_id(), 1, 0, 0)
        try:
            x.fetch_rooted()
            raise SystemExit("fetch without a bound plan did not fail")
        except b.CrsdrError:
            pass
        plan = b.Plan(nrows, B, b.MODE_DIGITAL, max_batch=T)
        wrong = b.Plan(nrows, B, b.MODE_DIGITAL, row_begin=1, row_count=nsig // 2, max_batch=T)
        try:
            x.bind_plan(wrong)
            raise SystemExit("a plan that does not own the rank's slab was accepted")
        except b.CrsdrError:
            pass
        x.bind_plan(plan)
        try:
            x.bind_plan(plan)
            raise SystemExit("second bind accepted")
        except b.CrsdrError:
            pass
        try:
            x.fetch_rooted()
            raise SystemExit("fetch with nothing submitted did not fail")
        except b.CrsdrError:
            pass
        got = []
        x.submit_batch(blocks[0:T], readcnt=rc[0:T], lag_mask=mask, seq=0)
        x.submit_batch(blocks[T:2 * T], readcnt=rc[T:2 * T], lag_mask=mask, seq=T)      # two outstanding
        try:
            x.submit_batch(blocks[2 * T:2 * T + 3], seq=2 * T)
            raise SystemExit("a third outstanding batch was accepted")
        except b.CrsdrError:
            pass
        got.append(x.fetch_rooted())
        x.submit_batch(blocks[2 * T:2 * T + 3], readcnt=rc[2 * T:2 * T + 3], lag_mask=mask, seq=2 * T)
        got.append(x.fetch_rooted())
        got.append(x.fetch_rooted())
        at = 0
        for k, nb in enumerate(sizes):
            first, pk, sc, tails = got[k]
            assert first == 0 and len(pk) == nb and len(sc) == nb, (k, first, len(pk))
            for t in range(nb):
                e = exp[k][t]
                assert np.array_equal(pk[t], e["packet"]), (k, t)
                for key in ("lag", "mag", "frac", "phasor"):
                    assert np.array_equal(sc[t][key].view(np.uint8), e[key].view(np.uint8)), (k, t, key)
                    assert np.array_equal(tails[key][t].view(np.uint8), e[key][1:].view(np.uint8)), (k, t, key)      # one rank owns every signal row
                assert np.array_equal(tails["readcnt"][t], rc[at + t][1:]), (k, t)
            at += nb
        x.close(); plan.close(); wrong.close(); ref.close()
        print("ENGINE OK")
    ''') % root
    for env in ({"CRSDR_XCHG_SELF": "1"}, {}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ENGINE OK" in r.stdout, (env, r.stdout[-2000:], r.stderr[-4000:])
