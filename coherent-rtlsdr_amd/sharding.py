"""Row sharding of the receive matrix across GPUs and its reassembly.

Every signal row depends only on itself and on row 0 (src/ccoherent.cc:177-179,189-234,262-283),
so rank g owns a contiguous slab of signal rows, the 2L-byte reference block is replicated,
and the only exchange step of the path is one gather of the int8 slabs (2 bytes per sample,
never cf32) plus 16 bytes per row of {lag, mag, phasor}.  torch.distributed is the transport:
backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.

The gather root rotates (a fixed root would ingest (G-1)/G of every block over its own xGMI
links and cap scaling; rotating spreads the ingest over all ranks).  Two exchange shapes:

* per block -- `gather_matrix` / `gather_batch`: block b is assembled on rank b mod G, slabs land
  in place in the root's packet (one send/recv per block and peer);
* per batch -- `exchange_batch` (what bench.py runs): the plans write their slabs densely
  (`crsdr_plan_bind_slab`), so a rank's batch of T blocks is one contiguous [T][rows_per_rank][B]
  buffer and the whole batch moves with ONE all-to-all: the root rotates in runs, rank q
  assembles blocks [q*T/G, (q+1)*T/G) of every batch (`batch_root`), and `crsdr_assemble_slabs`
  copies the received [G][T/G][rows_per_rank][B] chunks into the matrix rows of its T/G packets.
  Same bytes per xGMI link as the per-block shape, but 1 collective call per batch instead of
  ~1.75*T point-to-point operations -- at 8 GPUs a rank finishes a block every ~8 us, which
  per-operation host work cannot keep up with.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Slab:
    row_begin: int   # first owned signal row (global index, >= 1)
    row_count: int   # owned signal rows
    rows_per_rank: int


def slab_for_rank(nrows: int, world: int, rank: int) -> Slab:
    """Contiguous equal slabs of the nrows-1 signal rows; world must divide them."""
    nsig = nrows - 1
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} of {world}")
    if nsig % world:
        raise ValueError(f"{nsig} signal rows do not split evenly over {world} ranks")
    per = nsig // world
    return Slab(1 + rank * per, per, per)


def gather_root(block_index: int, world: int) -> int:
    return block_index % world


def matrix_view(packet, nrows: int, B: int):
    """[nrows][B] int8 view of the matrix inside a packet tensor (1-D int8/uint8, torch)."""
    off = 16 + 4 * nrows
    return packet[off: off + nrows * B].view(nrows, B)


def gather_matrix(packet, nrows: int, B: int, slab: Slab, root: int, group=None, async_op: bool = False):
    """Gather every rank's slab of matrix rows into `packet` on `root`.

    packet: this rank's full-size packet tensor; the plan has written rows
    [slab.row_begin, slab.row_begin + slab.row_count) (and row 0 + header, identical on every
    rank).  On the root the other ranks' slabs land in place; other ranks only send.
    Returns the torch.distributed work handle (or None).
    """
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return None
    m = matrix_view(packet, nrows, B)
    mine = m[slab.row_begin: slab.row_begin + slab.row_count]
    if rank == root:
        per = slab.rows_per_rank
        out = [m[1 + r * per: 1 + (r + 1) * per] for r in range(world)]
        return dist.gather(mine, gather_list=out, dst=root, group=group, async_op=async_op)
    return dist.gather(mine, gather_list=None, dst=root, group=group, async_op=async_op)


def gather_batch(packets, nrows: int, B: int, slab: Slab, first_block: int, group=None):
    """Reassemble a whole batch with ONE grouped point-to-point exchange.

    packets: list of T full-size packet tensors (block first_block + t each); this rank's plan
    has written its slab (and the replicated row 0 + header) into every one of them.  Block b is
    assembled on rank b mod G, so with T a multiple of G every GPU roots T/G blocks of the batch
    and every xGMI link carries the same 2L x rows_per_rank bytes in each direction -- there is no
    hot root, and the whole batch costs one RCCL group launch (ncclGroupStart/End under
    torch.distributed.batch_isend_irecv) instead of T gathers.
    Returns the list of work handles (empty for a single rank).
    """
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return []
    per = slab.rows_per_rank
    ops = []
    for t, pk in enumerate(packets):
        m = matrix_view(pk, nrows, B)
        root = gather_root(first_block + t, world)
        if rank == root:
            for r in range(world):
                if r != rank:
                    ops.append(dist.P2POp(dist.irecv, m[1 + r * per: 1 + (r + 1) * per], r, group))
        else:
            ops.append(dist.P2POp(dist.isend, m[slab.row_begin: slab.row_begin + slab.row_count], root, group))
    return dist.batch_isend_irecv(ops) if ops else []


def gather_scalars(local, slab: Slab, root: int, group=None):
    """Gather per-row scalars (a [nrows, k] tensor holding this rank's rows) onto root in place."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return None
    mine = local[slab.row_begin: slab.row_begin + slab.row_count]
    if rank == root:
        per = slab.rows_per_rank
        out = [local[1 + r * per: 1 + (r + 1) * per] for r in range(world)]
        return dist.gather(mine, gather_list=out, dst=root, group=group)
    return dist.gather(mine, gather_list=None, dst=root, group=group)


def batch_root(t: int, T: int, world: int) -> int:
    """Root of block t (index inside a batch of T blocks, T % world == 0) under `exchange_batch`."""
    if T % world:
        raise ValueError(f"batch of {T} blocks does not split over {world} ranks")
    return t // (T // world)


def rooted_blocks(T: int, world: int, rank: int) -> range:
    """The blocks of a batch this rank assembles: the hdr_first / hdr_count of crsdr_plan_bind_slab."""
    if T % world:
        raise ValueError(f"batch of {T} blocks does not split over {world} ranks")
    per = T // world
    return range(rank * per, (rank + 1) * per)


def slot_geometry(nrows: int, B: int, world: int) -> dict:
    """Send-slot geometry of include/crsdr.h (crsdr_exchange_geometry), restated so that CPU-only code can use it:
    a slot = [per][B] int8 rows | tail: int32 lag[per] | float mag[per] | float frac[per] | float phasor[per][2] | uint32 readcnt[per]."""
    per = (nrows - 1) // world
    if per * world != nrows - 1:
        raise ValueError(f"{nrows - 1} signal rows do not split evenly over {world} ranks")
    up16 = lambda v: (v + 15) // 16 * 16
    return {"per": per, "tail_offset": per * B, "slot_stride": up16(per * B + 24 * per), "scalars_stride": up16(20 * nrows)}


def rooted_range(nblocks: int, world: int, rank: int) -> range:
    """Blocks of a batch of nblocks that `rank` assembles (crsdr_exchange_rooted_blocks): runs of ceil(nblocks / world);
    the last ranks of a short batch get fewer blocks or none -- a ragged batch ships only its own bytes."""
    bpr = -(-nblocks // world)
    first = min(rank * bpr, nblocks)
    return range(first, first + min(bpr, nblocks - first))


_SPLITS = {}


def exchange_slots(recv, send, nblocks: int, slot_stride: int, group=None, async_op: bool = True):
    """ONE all-to-all for a batch of nblocks blocks whose slots carry rows + per-row scalars.

    send: this rank's slots [nblocks][slot_stride] bytes (written by its plan, crsdr_plan_bind_slab_ex); rank q gets the
    slots of the blocks it roots (`rooted_range`), so the split sizes differ per peer when the batch is ragged.
    recv: >= world * len(rooted_range(nblocks, world, rank)) * slot_stride bytes, filled [src][block][slot_stride]
    (the layout crsdr_assemble_slots takes).  Returns the work handle (async) or None.
    """
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    key = (world, rank, nblocks, slot_stride)
    splits = _SPLITS.get(key)
    if splits is None:                       # per batch shape, not per batch: this runs on the host's critical path at 8 GPUs
        counts = [len(rooted_range(nblocks, world, q)) for q in range(world)]
        in_splits = [c * slot_stride for c in counts]
        out_splits = [counts[rank] * slot_stride] * world
        splits = _SPLITS[key] = (in_splits, out_splits, sum(out_splits))
    in_splits, out_splits, total_out = splits
    return dist.all_to_all_single(recv.view(-1)[:total_out], send.view(-1)[: nblocks * slot_stride], out_splits, in_splits,
                                  group=group, async_op=async_op)


def exchange_batch(recv, send, group=None, async_op: bool = True):
    """One all-to-all for a whole batch.

    send: this rank's slab buffer, [T][rows_per_rank][B] int8/uint8 contiguous (block t of the batch at
    index t), written by its plan in slab mode.  recv: same size, receives [G][T/G][rows_per_rank][B]:
    chunk src = rank src's slabs of the T/G blocks rooted here.  Returns the work handle (async) or None.
    """
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if send.numel() != recv.numel() or send.numel() % world:
        raise ValueError("send / recv must have equal sizes divisible by the world size")
    return dist.all_to_all_single(recv.view(-1), send.view(-1), group=group, async_op=async_op)
