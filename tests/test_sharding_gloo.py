"""Multi-GPU path on CPU: the row sharding + rotating-root gather of coherent-rtlsdr_amd/sharding.py
with torch.distributed's gloo backend, world_size 2 (and 4).  The per-rank compute stand-in is
the CPU oracle (tests only -- on GPUs each rank runs its own crsdr_plan over the same slab, see
test_gpu_plan.py::test_slab_plans_reassemble_the_full_matrix); what is checked here is the
partition, the in-place gather into the packet layout and the root rotation."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nsig, L, nblocks, outdir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sharding = importlib.import_module("coherent-rtlsdr_amd.sharding")
    synth = importlib.import_module("coherent-rtlsdr_amd.synth")
    import oracle_py as O

    nrows, B = nsig + 1, 2 * L
    slab = sharding.slab_for_rank(nrows, world, rank)
    params = synth.RowParams(nsig, L, 55, dmax=L // 8)
    eng = O.Engine(nrows, B, O.DIGITAL)
    for t in range(nblocks):
        rows, _ = synth.make_block(nsig, L, 55, t, params=params)
        # this rank only "owns" its slab: everything else in its packet is poisoned before the gather
        mask = np.zeros(nrows, dtype=np.uint8)
        mask[slab.row_begin: slab.row_begin + slab.row_count] = 1
        out = eng.block(rows, seq=t, lag_mask=mask)
        pkt = torch.from_numpy(out["packet"].copy())
        m = sharding.matrix_view(pkt, nrows, B)
        own = np.zeros(nrows, dtype=bool)
        own[0] = True
        own[slab.row_begin: slab.row_begin + slab.row_count] = True
        m[torch.from_numpy(~own)] = 99
        scal = torch.zeros((nrows, 2), dtype=torch.float32)
        scal[slab.row_begin: slab.row_begin + slab.row_count, 0] = torch.from_numpy(
            out["lag"][slab.row_begin: slab.row_begin + slab.row_count].astype(np.float32))
        scal[slab.row_begin: slab.row_begin + slab.row_count, 1] = torch.from_numpy(
            out["mag"][slab.row_begin: slab.row_begin + slab.row_count])
        root = sharding.gather_root(t, world)
        sharding.gather_matrix(pkt, nrows, B, slab, root)
        sharding.gather_scalars(scal, slab, root)
        if rank == root:
            np.save(os.path.join(outdir, f"pkt_{t}.npy"), pkt.numpy())
            np.save(os.path.join(outdir, f"scal_{t}.npy"), scal.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _worker_batch(rank, world, port, nsig, L, T, outdir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sharding = importlib.import_module("coherent-rtlsdr_amd.sharding")
    synth = importlib.import_module("coherent-rtlsdr_amd.synth")
    import oracle_py as O

    nrows, B = nsig + 1, 2 * L
    slab = sharding.slab_for_rank(nrows, world, rank)
    params = synth.RowParams(nsig, L, 56, dmax=L // 8)
    eng = O.Engine(nrows, B, O.DIGITAL)
    mask = np.zeros(nrows, dtype=np.uint8)
    mask[slab.row_begin: slab.row_begin + slab.row_count] = 1
    own = np.zeros(nrows, dtype=bool)
    own[0] = True
    own[slab.row_begin: slab.row_begin + slab.row_count] = True
    first = 3                                     # batch starts at block 3: roots 3 % G, 4 % G, ...
    packets = []
    for t in range(T):
        rows, _ = synth.make_block(nsig, L, 56, first + t, params=params)
        pkt = torch.from_numpy(eng.block(rows, seq=first + t, lag_mask=mask)["packet"].copy())
        sharding.matrix_view(pkt, nrows, B)[torch.from_numpy(~own)] = 77   # poison what this rank does not own
        packets.append(pkt)
    for w in sharding.gather_batch(packets, nrows, B, slab, first):
        w.wait()
    for t in range(T):
        if rank == sharding.gather_root(first + t, world):
            np.save(os.path.join(outdir, f"bpkt_{t}.npy"), packets[t].numpy())
    dist.barrier()
    dist.destroy_process_group()


def _worker_a2a(rank, world, port, nsig, L, T, outdir):
    """The per-batch exchange shape of bench.py: dense slab buffer -> ONE all_to_all_single -> assemble.
    The oracle stands in for the plan (it fills the slab buffer the way crsdr_plan_bind_slab does) and plain
    tensor indexing stands in for crsdr_assemble_slabs (same index contract, tested on the GPU against it)."""
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sharding = importlib.import_module("coherent-rtlsdr_amd.sharding")
    synth = importlib.import_module("coherent-rtlsdr_amd.synth")
    import oracle_py as O

    nrows, B = nsig + 1, 2 * L
    slab = sharding.slab_for_rank(nrows, world, rank)
    per, Tg = slab.rows_per_rank, T // world
    params = synth.RowParams(nsig, L, 57, dmax=L // 8)
    eng = O.Engine(nrows, B, O.DIGITAL)
    mask = np.zeros(nrows, dtype=np.uint8)
    mask[slab.row_begin: slab.row_begin + slab.row_count] = 1
    mine = sharding.rooted_blocks(T, world, rank)
    send = torch.zeros((T, per, B), dtype=torch.int8)
    recv = torch.full((world, Tg, per, B), 55, dtype=torch.int8)
    packets = {}
    for t in range(T):
        rows, _ = synth.make_block(nsig, L, 57, t, params=params)
        out = eng.block(rows, seq=t, lag_mask=mask)
        m = sharding.matrix_view(torch.from_numpy(out["packet"].copy()), nrows, B)
        send[t] = m[slab.row_begin: slab.row_begin + slab.row_count]          # slab mode: owned rows only, dense
        if t in mine:                                                          # header + readcnt + row 0 only where rooted
            pkt = torch.from_numpy(out["packet"].copy())
            sharding.matrix_view(pkt, nrows, B)[1:] = 99
            packets[t] = pkt
        assert sharding.batch_root(t, T, world) == t // Tg
    w = sharding.exchange_batch(recv, send, async_op=True)
    w.wait()
    for j, t in enumerate(mine):
        m = sharding.matrix_view(packets[t], nrows, B)
        for src in range(world):
            m[1 + src * per: 1 + (src + 1) * per] = recv[src, j]
        np.save(os.path.join(outdir, f"apkt_{t}.npy"), packets[t].numpy())
    dist.barrier()
    dist.destroy_process_group()


def _worker_slots(rank, world, port, nsig, L, T, outdir):
    """The exchange bench.py runs: slots = owned rows + the 24 B/row tail {lag, mag, frac, phasor, readcnt} (what set_lag and the
    port-5557 payload need on the assembling rank, src/ccoherent.cc:232-233, src/cpacketizer.cc:127), ONE all-to-all with
    per-peer split sizes (a ragged batch ships only its own bytes), then assembly by the index contract of
    crsdr_assemble_slots (restated in numpy here; the kernel itself is tested on the GPU against the same contract)."""
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sharding = importlib.import_module("coherent-rtlsdr_amd.sharding")
    synth = importlib.import_module("coherent-rtlsdr_amd.synth")
    import oracle_py as O

    nrows, B = nsig + 1, 2 * L
    slab = sharding.slab_for_rank(nrows, world, rank)
    geo = sharding.slot_geometry(nrows, B, world)
    per, slot, toff = geo["per"], geo["slot_stride"], geo["tail_offset"]
    params = synth.RowParams(nsig, L, 58, dmax=L // 8)
    eng = O.Engine(nrows, B, O.DIGITAL)
    mask = np.zeros(nrows, dtype=np.uint8)
    own = slice(slab.row_begin, slab.row_begin + slab.row_count)
    mask[own] = 1
    mine = sharding.rooted_range(T, world, rank)
    send = np.zeros(T * slot, dtype=np.uint8)
    packets = {}
    for t in range(T):
        rows, _ = synth.make_block(nsig, L, 58, t, params=params)
        # this rank knows the read counters of row 0 and of its own rows only (a process that reads its own dongles)
        rc_true = (500 + 3 * t + 11 * np.arange(nrows)).astype(np.uint32)
        rc = np.full(nrows, 0xBAD00000 + rank, dtype=np.uint32)
        rc[0], rc[own] = rc_true[0], rc_true[own]
        out = eng.block(rows, seq=t, lag_mask=mask, readcnt=rc)
        s = send[t * slot: (t + 1) * slot]
        s[:per * B] = out["matrix"][own].reshape(-1).view(np.uint8)
        s[toff: toff + 4 * per] = out["lag"][own].view(np.uint8)
        s[toff + 4 * per: toff + 8 * per] = out["mag"][own].view(np.uint8)
        s[toff + 8 * per: toff + 12 * per] = out["frac"][own].view(np.uint8)
        s[toff + 12 * per: toff + 20 * per] = out["phasor"][own].view(np.uint8)
        s[toff + 20 * per: toff + 24 * per] = rc[own].view(np.uint8)
        if t in mine:                                                          # header + readcnt + row 0 only where rooted
            pkt = out["packet"].copy()
            pkt[16 + 4 * nrows + B:] = 99
            packets[t] = pkt
    recv = torch.full((max(1, world * len(mine) * slot),), 55, dtype=torch.uint8)
    w = sharding.exchange_slots(recv, torch.from_numpy(send), T, slot, async_op=True)
    w.wait()
    r = recv.numpy()
    for j, t in enumerate(mine):
        m = packets[t][16 + 4 * nrows:].reshape(nrows, B)
        lag, mag, frac = np.zeros(nrows, np.int32), np.zeros(nrows, np.float32), np.zeros(nrows, np.float32)
        ph = np.zeros(nrows, np.complex64)
        for src in range(world):
            sl = r[(src * len(mine) + j) * slot: (src * len(mine) + j + 1) * slot]
            rr = slice(1 + src * per, 1 + (src + 1) * per)
            m[rr] = sl[:per * B].view(np.int8).reshape(per, B)
            lag[rr], mag[rr] = sl[toff: toff + 4 * per].view(np.int32), sl[toff + 4 * per: toff + 8 * per].view(np.float32)
            frac[rr], ph[rr] = sl[toff + 8 * per: toff + 12 * per].view(np.float32), sl[toff + 12 * per: toff + 20 * per].view(np.complex64)
            packets[t][16:16 + 4 * nrows].view(np.uint32)[rr] = sl[toff + 20 * per: toff + 24 * per].view(np.uint32)      # the header's counters
        np.savez(os.path.join(outdir, f"slot_{t}.npz"), packet=packets[t], lag=lag, mag=mag, frac=frac, phasor=ph)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,T", [(2, 8), (4, 8), (4, 6), (2, 3)])
def test_slot_exchange_carries_rows_and_per_row_scalars(world, T, tmp_path, oracle, synth):
    # T = 6 over 4 ranks and T = 3 over 2: ragged batches (runs of 2 -> the last rank roots fewer blocks / none)
    nsig, L = 8, 256
    mp.spawn(_worker_slots, args=(world, _free_port(), nsig, L, T, str(tmp_path)), nprocs=world, join=True)
    params = synth.RowParams(nsig, L, 58, dmax=L // 8)
    eng = oracle.Engine(nsig + 1, 2 * L, oracle.DIGITAL)
    for t in range(T):
        rows, _ = synth.make_block(nsig, L, 58, t, params=params)
        exp = eng.block(rows, seq=t, readcnt=(500 + 3 * t + 11 * np.arange(nsig + 1)).astype(np.uint32))
        got = np.load(tmp_path / f"slot_{t}.npz")
        assert np.array_equal(got["packet"], exp["packet"]), t
        for k in ("lag", "mag", "frac", "phasor"):
            assert np.array_equal(got[k][1:], exp[k][1:]), (t, k)


@pytest.mark.parametrize("world", [2, 4])
def test_one_all_to_all_per_batch_reassembles_every_packet(world, tmp_path, oracle, synth):
    nsig, L, T = 8, 256, 8
    mp.spawn(_worker_a2a, args=(world, _free_port(), nsig, L, T, str(tmp_path)), nprocs=world, join=True)
    params = synth.RowParams(nsig, L, 57, dmax=L // 8)
    eng = oracle.Engine(nsig + 1, 2 * L, oracle.DIGITAL)
    for t in range(T):
        rows, _ = synth.make_block(nsig, L, 57, t, params=params)
        exp = eng.block(rows, seq=t)
        assert np.array_equal(np.load(tmp_path / f"apkt_{t}.npy"), exp["packet"]), t


@pytest.mark.parametrize("world", [2, 4])
def test_batched_rotating_root_exchange(world, tmp_path, oracle, synth):
    # one grouped point-to-point exchange per batch, block b assembled on rank b mod G
    nsig, L, T = 8, 256, 6
    mp.spawn(_worker_batch, args=(world, _free_port(), nsig, L, T, str(tmp_path)), nprocs=world, join=True)
    params = synth.RowParams(nsig, L, 56, dmax=L // 8)
    eng = oracle.Engine(nsig + 1, 2 * L, oracle.DIGITAL)    # fresh engine from block 3, like the workers' engines
    for t in range(T):
        rows, _ = synth.make_block(nsig, L, 56, 3 + t, params=params)
        exp = eng.block(rows, seq=3 + t)
        assert np.array_equal(np.load(tmp_path / f"bpkt_{t}.npy"), exp["packet"]), t


@pytest.mark.parametrize("world", [2, 4])
def test_rotating_root_gather_reassembles_the_packet(world, tmp_path, oracle, synth):
    nsig, L, nblocks = 8, 256, 5
    port = _free_port()
    mp.spawn(_worker, args=(world, port, nsig, L, nblocks, str(tmp_path)), nprocs=world, join=True)
    # single-rank truth
    params = synth.RowParams(nsig, L, 55, dmax=L // 8)
    eng = oracle.Engine(nsig + 1, 2 * L, oracle.DIGITAL)
    for t in range(nblocks):
        rows, _ = synth.make_block(nsig, L, 55, t, params=params)
        exp = eng.block(rows, seq=t)
        got = np.load(tmp_path / f"pkt_{t}.npy")
        assert np.array_equal(got, exp["packet"]), f"block {t} (root {t % world})"
        sc = np.load(tmp_path / f"scal_{t}.npy")
        assert np.array_equal(sc[1:, 0].astype(np.int32), exp["lag"][1:])
        assert np.array_equal(sc[1:, 1], exp["mag"][1:])


def test_slab_arithmetic():
    sharding = importlib.import_module("coherent-rtlsdr_amd.sharding")
    # cfg4: 1 + 1024 rows over 8 GPUs -> 128 signal rows each, contiguous, covering [1, 1025)
    slabs = [sharding.slab_for_rank(1025, 8, r) for r in range(8)]
    assert [s.row_begin for s in slabs] == [1 + 128 * r for r in range(8)]
    assert all(s.row_count == 128 for s in slabs)
    assert slabs[-1].row_begin + slabs[-1].row_count == 1025
    assert sharding.slab_for_rank(22, 1, 0) == sharding.Slab(1, 21, 21)
    with pytest.raises(ValueError):
        sharding.slab_for_rank(22, 8, 0)          # 21 signal rows do not split over 8
    with pytest.raises(ValueError):
        sharding.slab_for_rank(9, 2, 2)
    assert [sharding.gather_root(b, 8) for b in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]
