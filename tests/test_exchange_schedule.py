"""The exchange under the C ABI (include/crsdr.h: crsdr_exchange_*), host side, no GPU: the point-to-point schedule that
crsdr_exchange_batch issues through RCCL is pure arithmetic (crsdr_exchange_schedule), so the ranks of a node are simulated
here: every rank's sends to a peer are paired with that peer's receives from it, first in first out -- the matching rule of
ncclSend / ncclRecv inside one group -- and the bytes are moved between numpy buffers.  Checked for 1, 2, 4 and 8 ranks,
full, short and ragged batches, both modes:
  * every send has a receive of the same size (an unmatched or mis-sized pair is a hang or a fault on hardware),
  * no destination byte is written twice,
  * staged: the staging of rank q holds chunk r = rank r's slots of the blocks q roots, in the [src][block] order that
    crsdr_assemble_slots takes; in place: every remote row of every rooted packet arrived at its matrix offset and every
    tail in the tail staging,
  * and the geometry / rooting helpers of sharding.py agree with the C ABI's."""
import importlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def b():
    binding = importlib.import_module("coherent-rtlsdr_amd.binding")
    binding.build()
    return binding


def _simulate(b, G, nblocks, mode, nrows, B):
    geo = b.exchange_geometry(nrows, B, G)
    slot, tail_off, per = geo["slot_stride"], geo["tail_offset"], geo["per"]
    tail_bytes, tail_slot = 24 * per, (24 * per + 15) // 16 * 16
    moff = 16 + 4 * nrows
    pstride = (moff + nrows * B + 255) // 256 * 256
    rng = np.random.default_rng(G * 1000 + nblocks)
    roots = [b.rooted_blocks(nblocks, G, q) for q in range(G)]
    bufs = []
    for q in range(G):
        cnt = len(roots[q])
        bufs.append({0: rng.integers(1, 255, size=nblocks * slot, dtype=np.uint8),          # send slots (never zero: coverage check)
                     1: np.zeros(max(1, G * cnt * slot), dtype=np.uint8),                    # staging
                     2: np.zeros(max(1, cnt * pstride), dtype=np.uint8),                     # packets
                     3: np.zeros(max(1, G * cnt * tail_slot), dtype=np.uint8)})              # tail staging
    written = [{k: np.zeros(v.size, dtype=np.uint8) for k, v in bb.items()} for bb in bufs]
    ops = [b.exchange_schedule(G, q, nblocks, mode, nrows, B, pstride) for q in range(G)]
    for a in range(G):
        for c in range(G):
            sends = [o for o in ops[a] if o["peer"] == c and not o["is_recv"]]
            recvs = [o for o in ops[c] if o["peer"] == a and o["is_recv"]]
            assert len(sends) == len(recvs), (a, c, len(sends), len(recvs))
            assert a != c or not sends                                                        # nothing to itself
            for s, r in zip(sends, recvs):
                assert s["bytes"] == r["bytes"] and s["block"] == r["block"], (a, c, s, r)
                assert s["buffer"] == 0 and r["buffer"] in (1, 2, 3)
                src = bufs[a][0][s["offset"]: s["offset"] + s["bytes"]]
                assert src.size == s["bytes"]
                dst = bufs[c][r["buffer"]]
                assert r["offset"] + r["bytes"] <= dst.size
                dst[r["offset"]: r["offset"] + r["bytes"]] = src
                w = written[c][r["buffer"]]
                assert not w[r["offset"]: r["offset"] + r["bytes"]].any()                   # no byte written twice
                w[r["offset"]: r["offset"] + r["bytes"]] = 1
    for q in range(G):
        cnt = len(roots[q])
        for j, t in enumerate(roots[q]):
            for r in range(G):
                if r == q:
                    continue                                                                  # the own chunk never travels
                s_slot = bufs[r][0][t * slot: (t + 1) * slot]
                if mode == b.XCHG_STAGED:
                    got = bufs[q][1][(r * cnt + j) * slot: (r * cnt + j + 1) * slot]
                    assert np.array_equal(got, s_slot), (q, j, r)
                else:
                    o = j * pstride + moff + (1 + r * per) * B
                    assert np.array_equal(bufs[q][2][o: o + per * B], s_slot[:per * B]), (q, j, r)
                    to = (r * cnt + j) * tail_slot
                    assert np.array_equal(bufs[q][3][to: to + tail_bytes], s_slot[tail_off: tail_off + tail_bytes]), (q, j, r)
    return ops


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("G,nblocks", [(1, 3), (2, 4), (2, 5), (4, 8), (4, 6), (4, 3), (8, 64), (8, 20), (8, 4), (8, 1)])
def test_every_send_meets_its_receive(b, G, nblocks, mode):
    nrows, B = 1 + 8 * 3, 64
    ops = _simulate(b, G, nblocks, mode, nrows, B)
    if G == 8 and nblocks == 64 and mode == b.XCHG_STAGED:
        assert all(len(o) == 14 for o in ops)                       # one message per peer and direction
    if G == 8 and nblocks == 64 and mode == b.XCHG_INPLACE:
        assert all(len(o) == 4 * 8 * 7 for o in ops)                # rows + tail per (block, peer), both directions


def test_cfg4_geometry_and_rooting_match_sharding_py(b):
    sharding = importlib.import_module("coherent-rtlsdr_amd.sharding")
    for nrows, B, G in ((1025, 16384, 8), (1025, 16384, 1), (257, 16384, 4), (9, 512, 2)):
        g, s = b.exchange_geometry(nrows, B, G), sharding.slot_geometry(nrows, B, G)
        for k in ("per", "tail_offset", "slot_stride", "scalars_stride"):
            assert g[k] == s[k], (nrows, B, G, k)
    g = b.exchange_geometry(1025, 16384, 8)
    assert g["per"] == 128 and g["tail_offset"] == 128 * 16384 and g["slot_stride"] == 128 * 16384 + 3072 and g["scalars_stride"] >= 20 * 1025
    for nb in (1, 4, 8, 20, 64):
        for G in (1, 2, 4, 8):
            got = [b.rooted_blocks(nb, G, q) for q in range(G)]
            assert got == [sharding.rooted_range(nb, G, q) for q in range(G)]
            assert sorted(t for r in got for t in r) == list(range(nb))      # every block has exactly one root
    assert [len(b.rooted_blocks(20, 8, q)) for q in range(8)] == [3, 3, 3, 3, 3, 3, 2, 0]
    with pytest.raises(b.CrsdrError):
        b.exchange_geometry(1025, 16384, 3)                          # 1024 rows do not split over 3 ranks
    with pytest.raises(b.CrsdrError):
        b.exchange_schedule(4, 4, 8, 0, 9, 64)                       # rank out of range
