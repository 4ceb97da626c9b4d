#!/usr/bin/env python3
"""bench.py -- aligned IQ blocks/s of the coherent-alignment hot path on N MI355X GPUs.

Contract: `python bench.py --gpus N --steps K --warmup W` (N > 1 is launched by the driver
through torch.distributed.run, one rank per GPU over RCCL).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json metric "aligned IQ blocks/s (N ch x 8192)"; the target is quoted on
1024 ch x 8192 int8 IQ): one *step* = one aligned block = the full receive matrix for one time
step, (1 + 1024) rows x 8192 complex int8-IQ samples, taken from int8 input resident in HBM to
the int8 packet-layout matrix + {lag, mag, phasor} per row resident in HBM:
    K0 ref spectrum -> K1 fused int8->cf32, 16384-pt FFT, x conj(ref), IFFT, |.|^2, argmax
    -> K2 shift by lag, exact int dot -> phasor -> EMA -> rotate -> requantise
("track" cadence: every row is cross-correlated in every block, digital shift mode).
With N GPUs the 1024 signal rows are sharded (strong scaling: same total work), the ref row
is replicated, and the int8 slabs are gathered over xGMI onto a per-block rotating root.

Extra (reported, not `value`): the "locked" steady state of the reference (no FFT, phase path
only -- src/ccontrol.cc:117-120), which is the HBM-bound regime of this path.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VALU_PEAK_TF = 157.3      # MI355X_MICROARCH.md: peak FP32 vector


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--nsig", type=int, default=1024, help="signal rows (channels) in the receive matrix")
    ap.add_argument("--L", type=int, default=8192, help="complex samples per row per block")
    ap.add_argument("--nbuf", type=int, default=32, help="distinct resident input blocks rotated through")
    ap.add_argument("--batch", type=int, default=0, help="consecutive blocks per submit (one launch set per batch); "
                    "0 = 64, the plan's maximum (measured on one GPU: 16.5 k blocks/s at 8, 17.1 k at 16, 17.6 k at 64)")
    ap.add_argument("--mode", choices=["digital", "faithful"], default="digital")
    ap.add_argument("--cfg5", action="store_true", help="BASELINE config 5 instead: 1 + 21 rows x 2^20 samples (long-block path)")
    ap.add_argument("--frac-apply", action="store_true", help="fractional-delay correction on (crsdr_plan_set_frac_apply, D = this block's estimate): an extra, never the default workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the locked-mode / large-working-set extras")
    ap.add_argument("--skip-pcie", action="store_true", help="skip the host-buffer (PCIe-inclusive) extra: its submits carry 16 blocks, "
                    "which would mix launch sizes into a profile whose per-kernel averages are meant for 64-block launches")
    ap.add_argument("--cpu-blocks", type=int, default=0, help="oracle blocks to time (0 = auto, ~10-20 s)")
    ap.add_argument("--repeats", type=int, default=0, help="repeats of the timed region of exactly --steps steps, the median is reported "
                    "(0 = auto: 5 when the region is shorter than ~50 ms, else 1)")
    return ap.parse_args()


def main():
    args = parse()
    if args.cfg5:
        args.nsig, args.L, args.batch, args.nbuf, args.no_extras, args.no_cpu_baseline = 21, 1 << 20, 1, 2, True, True
    import numpy as np
    import torch
    import torch.distributed as dist

    pkg = importlib.import_module("coherent-rtlsdr_amd")
    b, synth, sharding = pkg.binding, pkg.synth, pkg.sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # CRSDR_BENCH_REHEARSAL=1: every rank on cuda:0, gloo transport with the slabs staged through host
    # memory -- exercises the sharded plans / exchange bookkeeping on a one-GPU box (numbers meaningless)
    rehearsal = os.environ.get("CRSDR_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # CRSDR_BENCH_FORCE_EXCHANGE=1 (one process): the N > 1 code path with a process group of ONE rank -- slots, the all-to-all
    # (to itself), the assembly on the side stream, the three ring-buffered sets -- to read its host cost per batch and its
    # stream ordering off a one-GPU box before the driver's node runs it with real peers
    force_x = world == 1 and os.environ.get("CRSDR_BENCH_FORCE_EXCHANGE") == "1"
    multi = world > 1 or force_x
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    nsig, L = args.nsig, args.L
    nrows, B = nsig + 1, 2 * L
    slab = sharding.slab_for_rank(nrows, world, rank)
    mode = b.MODE_DIGITAL if args.mode == "digital" else b.MODE_FAITHFUL

    # ---- synthetic resident input: nbuf distinct blocks, each rank fills row 0 + its slab ----------
    # blocks per submit: ONE batch per timed region when the region fits the plan's maximum of 64 blocks (the driver's
    # --steps 20: T = 20 on one GPU and on eight), else batches of 64 (several GPUs: the largest multiple of the world size,
    # so that every rank roots the same number of blocks).  Regions are issued back to back -- the series is fenced once, see
    # `series()` -- so the three ring-buffered output sets keep region r's exchange under region r + 1's compute whatever T is;
    # r02 sized T for ">= 4 batches inside one fenced region", which at --steps 20 on 8 GPUs meant 8 + 8 + 4 blocks and 91 us
    # of fixed cost per batch.  A ragged last batch ships only its own bytes.
    if args.batch > 0:
        T = max(1, min(args.batch, 64))
    elif args.steps <= 64:
        T = max(1, args.steps)
    else:
        T = 64 if world == 1 else max(world, 64 // world * world)
    nbuf = max(2 * T if multi else T, (args.nbuf // T) * T)   # whole batches, contiguous in HBM
    seed = synth.config_seed(4)
    params = synth.RowParams(nsig, L, seed)
    d_in = torch.empty((nbuf, nrows, B), dtype=torch.uint8, device=dev)
    for t in range(nbuf):
        rows, _ = synth.make_block(nsig, L, seed, t, params=params) if world == 1 else _make_slab(synth, params, nsig, L, seed, t, slab)
        d_in[t].copy_(torch.from_numpy(rows.view(np.uint8)))
    host_block0 = rows if world == 1 else None
    block_bytes = nrows * B

    plan = b.Plan(nrows, B, mode, device=local_rank, row_begin=slab.row_begin, row_count=slab.row_count, max_batch=T)
    # An explicit compute stream, made torch's current stream: the plan launches on it, torch.distributed's collectives order
    # themselves behind it, and the events below are recorded on it.  (torch's DEFAULT stream is the null stream, handle 0 --
    # which crsdr_plan_set_stream reads as "the plan's own stream": r01 / r02 ran the plan on its own non-blocking stream while
    # every wait_event / collective ordering of the N > 1 path went to the null stream, i.e. the all-to-all was not ordered behind
    # the submit it ships.  One GPU never noticed: its fences are device-wide.)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    torch.cuda.synchronize()                            # the input blocks above were filled on the null stream
    plan.set_stream(stream.cuda_stream)
    if args.frac_apply:
        plan.set_frac_apply(True, 1.0, None)
    # NSETS ring-buffered output sets (the exchange of batch i runs under the compute of batch i+1); the matrix
    # of every packet is 16-byte aligned.  One GPU: T packets per set.  Several GPUs: per set one slot buffer
    # [T][slot_stride] (rows + the 24 B/row {lag, mag, frac, phasor, readcnt} tail of the rank's slab: the all-to-all's send side),
    # one receive buffer [G][Tg][slot_stride], the Tg = ceil(T/G) packets this rank assembles and their scalars blocks.
    NSETS = 3 if multi else 2
    Tg = -(-T // world)
    pstride = (plan.packet_bytes + 255) // 256 * 256
    packets = [torch.zeros(pstride * Tg + 64, dtype=torch.uint8, device=dev) for _ in range(NSETS)]
    pk_off = [(-(p.data_ptr() + plan.matrix_offset)) % 16 for p in packets]
    pk_view = [[p[o + t * pstride: o + t * pstride + plan.packet_bytes] for t in range(Tg)] for p, o in zip(packets, pk_off)]
    flags = b.REFNOISE_ENABLED | b.INPUT_READY
    if multi:
        geo = b.exchange_geometry(nrows, B, world)
        slot, toff, sstride = geo["slot_stride"], geo["tail_offset"], geo["scalars_stride"]
        send = [torch.zeros(T * slot, dtype=torch.uint8, device=dev) for _ in range(NSETS)]
        recv = [torch.zeros(world * Tg * slot, dtype=torch.uint8, device=dev) for _ in range(NSETS)]
        scal = [torch.zeros(Tg * sstride, dtype=torch.uint8, device=dev) for _ in range(NSETS)]
        xstream = torch.cuda.Stream(device=dev)       # waits for the exchange, then assembles: never blocks the compute stream
        done = [None] * NSETS                          # event: set k's exchange + assembly finished
        done_pool = [torch.cuda.Event() for _ in range(NSETS)]      # re-recorded per batch (creating one costs the host ~5 us)
        ready_ev = torch.cuda.Event()
    xchg = {"c": None}                                 # crsdr_exchange (RCCL under the C ABI) when the C transport is selected

    gib = {"n": 0, "last_full": None}                  # batches issued so far: output sets, input blocks and seq keep rotating across regions

    def run_batch(nb, fl=flags):
        """the next nb blocks of the stream: one submit; with several GPUs then ONE exchange of the batch's slots
        (rank q assembles the q-th run of ceil(nb/G) blocks, rows AND per-row scalars) and the assembly on a side stream"""
        ib = gib["n"]
        gib["n"] = ib + 1
        if nb == T:
            gib["last_full"] = ib
        k = ib % NSETS
        first_buf = (ib * T) % nbuf
        if not multi:
            plan.bind_packet(pk_view[k][0].data_ptr(), pstride)
            plan.submit(d_in[first_buf].data_ptr(), seq=ib * T, flags=fl, nblocks=nb, block_stride=block_bytes)
            return
        if done[k] is not None:
            stream.wait_event(done[k])               # set k is free again: its exchange and assembly are complete
        mine = sharding.rooted_range(nb, world, rank)
        plan.bind_packet(pk_view[k][0].data_ptr(), pstride)
        plan.bind_slab_ex(send[k].data_ptr(), slot, mine.start, len(mine), toff)
        plan.submit(d_in[first_buf].data_ptr(), seq=ib * T, flags=fl, nblocks=nb, block_stride=block_bytes)
        if rehearsal:
            plan.sync()
            h_send, h_recv = send[k].cpu(), torch.empty(world * Tg * slot, dtype=torch.uint8)
            sharding.exchange_slots(h_recv, h_send, nb, slot, async_op=False)
            recv[k].copy_(h_recv)
            if len(mine):
                b.assemble_slots(pk_view[k][0].data_ptr(), pstride, scal[k].data_ptr(), sstride, nrows, B, recv[k].data_ptr(), world, len(mine), slot, toff,
                                 stream=stream.cuda_stream)
            return
        if xchg["c"] is not None:
            # the same exchange under the C ABI: grouped ncclSend / ncclRecv + assembly, all enqueued on the side stream
            ready_ev.record(stream)
            xstream.wait_event(ready_ev)
            xchg["c"].batch(b.XCHG_STAGED, send[k].data_ptr(), recv[k].data_ptr(), nb, pk_view[k][0].data_ptr(), pstride, scal[k].data_ptr(), sstride,
                            nrows, B, xstream.cuda_stream)
            done[k] = done_pool[k]
            done[k].record(xstream)
            return
        work = sharding.exchange_slots(recv[k], send[k], nb, slot, async_op=True)     # RCCL stream: ordered after the submit above
        with torch.cuda.stream(xstream):
            work.wait()                              # stream-level: xstream waits for the all-to-all
            if len(mine):
                b.assemble_slots(pk_view[k][0].data_ptr(), pstride, scal[k].data_ptr(), sstride, nrows, B, recv[k].data_ptr(), world, len(mine), slot, toff,
                                 stream=xstream.cuda_stream)
            done[k] = done_pool[k]
            done[k].record(xstream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(nsteps, fl=flags):
        """exactly nsteps blocks, in batches of T (the last batch may be shorter)"""
        left = nsteps
        while left > 0:
            nb = min(T, left)
            run_batch(nb, fl)
            left -= nb

    def series(nsteps, regions, fl=flags):
        """`regions` timed regions of exactly nsteps steps each, issued back to back.  The barrier (+ device sync) is BEFORE the
        clock starts only; at the end every rank waits for ITS OWN last exchange and assembly (the `done` events, then a local
        device sync) -- no collective inside the interval: r02 ended the interval with dist.barrier(), tens of microseconds of
        RCCL latency charged to a 0.2 ms region.  The MAX over ranks is taken after the clocks have stopped.
        Returns (host seconds, GPU-event seconds between the first submit and the last assembly, host seconds spent issuing)."""
        fence()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        for _ in range(regions):
            run_steps(nsteps, fl)
        t_issue = time.perf_counter() - t0              # host time to enqueue everything (no waiting on the GPU)
        if multi:
            for ev in done:
                if ev is not None:
                    stream.wait_event(ev)               # this rank's outstanding exchanges + assemblies
        e1.record(stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        dt_ev = 1e-3 * e0.elapsed_time(e1)
        if world > 1:
            t = torch.tensor([dt, dt_ev, t_issue], device="cpu" if rehearsal else dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, dt_ev, t_issue = (float(x) for x in t.tolist())
        return dt, dt_ev, t_issue

    def timed(nsteps, fl=flags):
        """one fenced region (what r01 / r02 reported the median of)"""
        return series(nsteps, 1, fl)[0]

    # communicator set-up (not a step): RCCL opens its point-to-point channels lazily, on the first exchange
    # between each rank pair -- do that here, on the still-empty buffers, so that connection set-up cannot land
    # inside the timed region whatever the warm-up length is
    if multi and not rehearsal:
        sharding.exchange_slots(recv[NSETS - 1], send[NSETS - 1], T, slot, async_op=False)
        torch.cuda.synchronize()
        dist.barrier()

    # ---- warm-up, fenced regions (extras), then the timed series with HIP events on K1's launch stream ----------
    run_steps(args.warmup)
    fence()
    # How much is timed: a device that has idled (the seconds of host-side input generation above) needs ~40 ms of sustained load
    # before its clocks settle -- measured r02, `region_ms` of a 60-repeat run: 1.17, 1.19, 1.28, 1.25, 1.20 ... 0.99 ms for the
    # same 20 steps.  So the region of exactly --steps steps is issued `repeats` times, ~250 ms of work (5 .. 400 regions):
    #   (1) each between its own fences -- `value_fenced_median`, `region_ms` (the series in order), `value_first5` (the clock
    #       ramp): what r01 / r02 reported, kept as extras; a fenced region pays its cold start (K0 exposed, launch gaps, the
    #       drain of the last exchange) once per --steps steps;
    #   (2) back to back, fenced ONCE around the whole series -- `value`: blocks per second of the stream the engine exists for,
    #       in which region r's exchange / K0 of region r + 1 run under their neighbours' compute.
    nbatch = -(-args.steps // T)
    est_ms = args.steps * (0.55 if args.cfg5 else 0.05 * (nsig + 1) / 1025) / world
    repeats = args.repeats if args.repeats > 0 else int(min(400, max(5, -(-250.0 // max(est_ms, 1e-3)))))
    repeats = min(repeats, max(1, 4096 // nbatch))                    # one HIP-event pair per K1 launch
    if rehearsal and args.repeats <= 0:
        repeats = min(repeats, 5)                                     # bookkeeping run through host memory: its timing means nothing
    dts = [timed(args.steps) for _ in range(repeats)]
    # hipEvent pairs around the dominant kernel (K1) only -- every pair costs stream time
    plan.enable_profiling(min(max(nbatch * repeats, 1), 4096), 1 << b.KERNEL_XCORR_LAG)
    dt_series, dt_series_ev, t_issue = series(args.steps, repeats)
    dt = dt_series / repeats                                          # per region of --steps steps
    host_issue = {"s": t_issue / repeats}
    # checked right here, before later (untimed) runs reuse the packet sets: the last full batch -- every packet this rank
    # assembled must hold every rank's rows under the header of the right block, and its scalars block every rank's lags
    assembled_ok = scalars_ok = True
    if multi:
        torch.cuda.synchronize()
        if gib["last_full"] is not None and gib["n"] - gib["last_full"] <= NSETS:
            ib = gib["last_full"]                    # the series' last full batch: its output set has not been reused (fewer than NSETS batches after it)
            mine = sharding.rooted_range(T, world, rank)
            for j, t_ in enumerate(mine):
                pkv = pk_view[ib % NSETS][j]
                m = sharding.matrix_view(pkv, nrows, B)
                for r in range(world):
                    assembled_ok &= bool(m[1 + r * slab.rows_per_rank].ne(0).any().item())
                assembled_ok &= int(pkv[:4].cpu().numpy().view(np.uint32)[0]) == ib * T + t_
                sc = b.parse_scalars(scal[ib % NSETS][j * sstride: (j + 1) * sstride].cpu().numpy(), nrows)
                scalars_ok &= bool(np.array_equal(sc["lag"][1:], params.d)) and bool(np.all(np.abs(sc["phasor"][1:]) > 0.2))
        flag = torch.tensor([1 if assembled_ok else 0, 1 if scalars_ok else 0], device="cpu" if rehearsal else dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        assembled_ok, scalars_ok = bool(flag[0].item()), bool(flag[1].item())

    host_ms_per_batch = 1e3 * host_issue["s"] / max(1, -(-args.steps // T))
    k_ms = {"xcorr_lag": plan.kernel_times_ms(b.KERNEL_XCORR_LAG)}
    # the other kernels: a short untimed loop with every pair recorded
    plan.enable_profiling(64, 0xF)
    run_steps(4 * T)
    fence()
    for name, k in (("ref_spectrum", b.KERNEL_REF_SPECTRUM), ("phase_dot", b.KERNEL_PHASE_DOT), ("align_quant", b.KERNEL_ALIGN_QUANT)):
        k_ms[name] = plan.kernel_times_ms(k)
    plan.enable_profiling(0)
    full_batches = args.steps // T      # launches per repeat that carried exactly T blocks
    blocks_per_s = args.steps / dt
    # K0 exposed: one cold batch (nothing in flight before it) with whole-submit events: what the batch takes beyond its K1 and
    # phase kernel is the reference-spectrum kernel it had to wait for plus launch gaps; in the steady state K0 of batch b+1
    # runs on the aux stream under batch b and the region's own "unhidden" remainder (below) is what is left of it
    cold = None
    if world == 1:
        plan.enable_profiling(4, 0xF | (1 << 31))
        fence()
        run_batch(T)
        fence()
        tot = plan.last_elapsed_ms()
        parts = [plan.kernel_times_ms(k) for k in (b.KERNEL_XCORR_LAG, b.KERNEL_PHASE_DOT, b.KERNEL_ALIGN_QUANT)]
        cold = {"submit_ms": tot, "k0_exposed_ms": max(0.0, tot - sum(float(x[-1]) for x in parts if len(x)))}
        plan.enable_profiling(0)

    # parity spot-check of the timed path against the injected delays (every rank, its slab)
    out = plan.fetch(want_packet=False)
    own = slice(slab.row_begin, slab.row_begin + slab.row_count)
    lags_ok = bool(np.array_equal(out["lag"][own], params.d[slab.row_begin - 1: slab.row_begin - 1 + slab.row_count]))

    result = None
    if rank == 0:
        A_block = nrows * B                                   # algorithmic bytes per block
        k1_all = k_ms["xcorr_lag"]
        k1s = [x for i, x in enumerate(k1_all) if ((i % nbatch) < full_batches or not full_batches)]   # the series' launches that carried T blocks
        k1 = float(np.mean(k1s)) if len(k1s) else float("nan")
        tb = T if full_batches else args.steps
        k1_bytes = tb * slab.row_count * B                    # int8 bytes one K1 launch consumes (tb blocks)
        achieved = k1_bytes / (k1 * 1e-3) / 1e9               # GB/s
        flop_row = 2 * 5 * B * np.log2(B) + 9 * B + 16 * L    # SURVEY 8d VALU view, per signal row
        result = {
            "metric": f"aligned IQ blocks/s ({nsig} ch x {L})", "value": blocks_per_s, "unit": "blocks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "repeats": repeats,
            "timing": {"clock": "host perf_counter around `repeats` back-to-back regions of --steps steps: barrier + device sync before t0 only, "
                                "each rank's own device sync (after its last exchange + assembly) at the end, MAX over ranks taken afterwards",
                       "regions": repeats, "steps_timed": args.steps * repeats, "series_ms": 1e3 * dt_series,
                       "gpu_event_ms": 1e3 * dt_series_ev, "value_gpu_events": args.steps * repeats / dt_series_ev,
                       "note": "gpu_event_ms: hipEvents on the compute stream from the first submit to after the last assembly"},
            "value_fenced_median": args.steps / float(np.median(dts)),                               # every region between its own fences (what r01 / r02 reported as value)
            "value_spread": [min(blocks_per_s, args.steps / max(dts)), max(blocks_per_s, args.steps / min(dts))],     # fenced regions and the series
            "region_ms": [round(1e3 * dts[i], 4) for i in sorted(set(np.linspace(0, len(dts) - 1, min(len(dts), 48)).astype(int).tolist()))],   # the fenced regions in order (evenly thinned to <= 48 entries)
            "value_first5": args.steps / float(np.median(dts[:5])),                                  # fenced, a device just out of idle: the ramp (see the comment at `repeats`)
            "batches_timed": {"per_repeat": nbatch, "whole": full_batches, "ragged_blocks": args.steps - full_batches * T, "blocks_per_batch": T},
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{'cfg5' if args.cfg5 else {21: 'cfg2', 256: 'cfg3', 1024: 'cfg4'}.get(nsig, 'custom') if L == 8192 else 'custom'}: 1 ref + {nsig} signal rows x {L} int8 IQ samples per block, track cadence "
                                   f"(FFT xcorr every block), {args.mode} mode, inputs resident in HBM, "
                                   f"{nbuf} rotating input blocks, {T} blocks per submit",
                       "rows": nrows, "L": L, "fft_len": B, "mode": args.mode, "batch": T,
                       "forced_exchange_path": force_x, "frac_apply": bool(args.frac_apply),
                       "parallelism": f"rows sharded x{world}, ref replicated, rotating-root gather of int8 rows + 24 B/row {{lag, mag, frac, phasor, readcnt}}: one RCCL all-to-all per {T}-block batch + local assembly, overlapped with the next batch" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "k_xcorr_lag", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": _committed_traffic("k_xcorr_lag", tb)[0],
                         "traffic_source": _committed_traffic("k_xcorr_lag", tb)[1],
                         "algorithmic_bytes_per_launch": k1_bytes, "avg_launch_ms": k1,
                         "valu_frac": (tb * slab.row_count * flop_row / (k1 * 1e-3)) / (FP32_VALU_PEAK_TF * 1e12),
                         "blocks_per_launch": tb,
                         "note": "track cadence is fp32-VALU/LDS bound (157 flop per input byte vs machine balance 20); "
                                 "the HBM-bound regime of this path is the locked mode below"},
            "whole_path": {"algorithmic_bytes_per_block": A_block,
                           "hbm_read_frac": A_block * blocks_per_s / (world * HBM_PEAK_GBS * 1e9)},
            "kernel_ms": {k: (k1 if k == "xcorr_lag" else float(np.mean(v)) if len(v) else None) for k, v in k_ms.items()},   # per launch of `batch` blocks
            "host_issue_ms_per_batch": host_ms_per_batch,
            "k0": {"launch_ms": float(np.mean(k_ms["ref_spectrum"])) if len(k_ms["ref_spectrum"]) else None, "cold_batch": cold,
                   "note": "K0 runs on the aux stream under the previous batch; cold_batch = one batch with nothing in flight before it"},
            "lags_exact": lags_ok, "matrix_assembled": assembled_ok if multi else None, "scalars_assembled": scalars_ok if multi else None,
            "env": _env(torch, dev, b, local_rank),
        }

    if rank == 0 and args.cfg5:
        # long blocks: "k_xcorr_lag" above is the whole four-step pipeline (stages A + B + C + finalize, events around all of it).
        # Beside the algorithmic figure (each int8 input byte once), the traffic MODEL of DESIGN.md section 4: per signal row
        # A reads B bytes and writes 8B, B reads and writes 8B, C reads 8B = 33 B bytes (66 MiB per 2 MiB row); the reference row
        # adds 17 B bytes.  model_GBs says how fast the stages stream what the four-step algorithm has to move.
        model = (33 * nsig + 17) * B
        result["roofline"].update({"kernel": "long-block stages A+B+C (k_long_fwd_cols, k_rows14_cf32p, k_long_inv_cols)",
                                   "traffic_model_bytes_per_block": model, "traffic_model_GBs": model / (k1 * 1e-3) / 1e9,
                                   "traffic_model_frac": model / (k1 * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_apply": bool(args.frac_apply),
                                   "traffic": None, "traffic_source": "per-stage PMC bytes of this configuration: profiles/r02_cfg5_traffic_notes.json "
                                                                      "(stage writes equal the model exactly; not rescaled into this line)",
                                   "note": "HBM-streaming regime: the four-step transform moves 33x the algorithmic bytes through HBM / the memory-side cache "
                                           "(cf32 intermediates); traffic_model_* prices the stages against that, achieved/frac against the int8 input alone"})

    # ---- extras: locked steady state (phase path only) ---------------------------------------------
    if not args.no_extras:
        fl_locked = flags | b.NO_LAG
        run_steps(2 * T, fl_locked)
        # the rate is timed without per-kernel events over >= 16 batches (a batch is ~0.6 ms: a shorter region would
        # mostly measure its own fences); the kernel's launch duration comes from a separate short profiled run
        n_l = max(args.steps, 16 * T) // T * T
        dts_l = [timed(n_l, fl_locked) for _ in range(int(min(40, max(3, 100.0 // (n_l * 0.009)))))]     # ~100 ms of timed work, median (as `value`)
        dt_l = float(np.median(dts_l))
        plan.enable_profiling(256, (1 << b.KERNEL_PHASE_DOT) | (1 << b.KERNEL_ALIGN_QUANT))
        run_steps(4 * T, fl_locked)
        fence()
        k2 = plan.kernel_times_ms(b.KERNEL_ALIGN_QUANT)
        k2a = plan.kernel_times_ms(b.KERNEL_PHASE_DOT)
        plan.enable_profiling(0)
        if rank == 0:
            k2m = float(np.mean(k2))
            k2am = float(np.mean(k2a)) if len(k2a) else None      # None: the fused phase kernel read every row once
            rd = T * (slab.row_count + 1) * B
            result["locked"] = {"blocks_per_s": n_l / dt_l, "ms_per_step": 1e3 * dt_l / n_l,
                                "roofline": {"bound": "hbm", "kernel": "k_align_quant" if k2am is not None else "k_align_fused", "achieved": rd / (k2m * 1e-3) / 1e9,
                                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rd / (k2m * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "read_plus_write_GBs": 2 * rd / (k2m * 1e-3) / 1e9, "avg_launch_ms": k2m,
                                             "phase_dot_avg_launch_ms": k2am, "blocks_per_launch": T,
                                             "traffic": _committed_traffic("k_align_quant" if k2am is not None else "k_align_fused", T)[0],
                                             "traffic_source": _committed_traffic("k_align_quant" if k2am is not None else "k_align_fused", T)[1]},
                                "hbm_read_frac": (nrows * B) * (n_l / dt_l) / (world * HBM_PEAK_GBS * 1e9)}

    # ---- extra: downstream covariance of one aligned matrix on the matrix cores ----------
    if not args.no_extras and world == 1:
        rxx = torch.empty((nsig, nsig, 2), dtype=torch.float32, device=dev)
        mptr = pk_view[0][0].data_ptr() + plan.matrix_offset
        b.covariance_device(rxx.data_ptr(), mptr, nrows, B)
        t0 = time.perf_counter()
        for _ in range(10):
            b.covariance_device(rxx.data_ptr(), mptr, nrows, B)
        dtc = (time.perf_counter() - t0) / 10
        result["covariance"] = {"ms": 1e3 * dtc, "int8_TOPS": 3 * 2 * nsig * nsig * B / dtc / 1e12,
                                "note": "Rxx = X^H X / L of the signal rows, host-synchronous call; int8_TOPS counts the full matrix at 3 int8 products "
                                        "per sample pair as in r01 / r02 (the r03 kernel forms the upper triangle with 2)"}

    # ---- extra: host-buffer (PCIe-inclusive) rate -- reported for DESIGN.md, never `value` -----------
    if not args.no_extras and not args.skip_pcie and world == 1:
        Tp = min(T, 16)
        hp = b.Plan(nrows, B, mode, device=local_rank, max_batch=Tp)
        hstride = hp.packet_stride
        pin_rows = [b.PinnedArray((Tp, nrows, B), np.int8) for _ in range(2)]           # crsdr_host_alloc: what a C host hands over
        pin_pk = [b.PinnedArray((Tp * hstride,), np.int8) for _ in range(2)]
        pin_lag = [b.PinnedArray((Tp, nrows), np.int32) for _ in range(2)]
        for pr in pin_rows:
            pr.array[:] = host_block0
        pageable = np.ascontiguousarray(np.broadcast_to(host_block0, (Tp, nrows, B)))
        res = {}
        # (a) synchronous: submit + fetch of the last packet per batch, pageable and page-locked rows (round-1 figure)
        for name, hrows in (("pageable", pageable), ("pinned_sync", pin_rows[0].array)):
            hp.submit(hrows, seq=0)
            hp.fetch()
            t0 = time.perf_counter()
            nh = 3
            for i in range(nh):
                hp.submit(hrows, seq=i * Tp)
                hp.fetch()
            res[name] = nh * Tp / (time.perf_counter() - t0)
        # (b) pipelined: EVERY packet comes back (crsdr_plan_fetch_batch_async), upload of batch i + 1 under the download of batch i
        def pipe(n):
            hp.submit(pin_rows[0].array, seq=0)
            hp.fetch_batch_async(pin_lag[0].array, None, None, None, pin_pk[0].array, hstride)
            for i in range(1, n):
                hp.submit(pin_rows[i & 1].array, seq=i * Tp)
                hp.fetch_batch_async(pin_lag[i & 1].array, None, None, None, pin_pk[i & 1].array, hstride)
                hp.fetch_wait()
            hp.fetch_wait()
        pipe(3)
        t0 = time.perf_counter()
        npipe = 12
        pipe(npipe)
        res["pipelined"] = npipe * Tp / (time.perf_counter() - t0)
        ok_pipe = bool(np.array_equal(pin_lag[(npipe - 1) & 1].array[Tp - 1][1:], params.d))
        result["pcie_inclusive"] = {"blocks_per_s": res["pipelined"], "sync_last_packet_only_blocks_per_s": res["pinned_sync"],
                                    "pageable_blocks_per_s": res["pageable"], "lags_exact": ok_pipe,
                                    "GBs_each_way": res["pipelined"] * nrows * B / 1e9,
                                    "note": f"host int8 rows in and EVERY host packet out ({Tp} blocks per submit, page-locked via crsdr_host_alloc; "
                                            "crsdr_plan_fetch_batch_async: the upload of a batch runs under the download of the previous one); "
                                            "PCIe Gen5 x16, 16.8 MB per block each way; the C++ engine's figure is coherent_demo --bench"}
        hp.close()
        for x in pin_rows + pin_pk + pin_lag:
            x.close()

    # ---- CPU baseline: the oracle (C port of the reference path) on this host's cores --------------
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        result["cpu_baseline"] = _cpu_baseline(args, host_block0, nrows, B, mode)
    elif rank == 0:
        result["cpu_baseline"] = None

    # ---- the ONE result line goes out here, before anything that has not run on real links yet ----
    if rank == 0:
        print(json.dumps(result), flush=True)

    # ---- opt-in extra, several GPUs (CRSDR_BENCH_C_EXCHANGE=1): the same exchange through the C ABI (crsdr_exchange_*:
    # grouped ncclSend / ncclRecv issued by libcrsdr.so itself, no Python collective in the loop), re-timing the same
    # series.  The N > 1 RCCL wiring of this leg has never run on more than one GPU (RCCL refuses two ranks on one device;
    # what is covered: the schedule for 1 .. 8 simulated ranks, the transport with one rank), so it is OFF by default, runs
    # after the result line has been printed, reports on stderr, and a hang is a FAILURE: the watchdog exits non-zero.
    if world > 1 and not rehearsal and os.environ.get("CRSDR_BENCH_C_EXCHANGE", "0") == "1":
        import threading

        def bail():
            print(json.dumps({"c_exchange": {"status": "timeout: the C-ABI RCCL leg did not finish within 90 s", "rank": rank}}), file=sys.stderr, flush=True)
            os._exit(3)                              # a hung collective is a finding, not an ok run

        wd = threading.Timer(90.0, bail)
        wd.daemon = True
        wd.start()
        c_res, c_fail = None, False
        try:
            ids = [b.exchange_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            xchg["c"] = b.Exchange(ids[0], world, rank, local_rank)
            run_steps(max(args.warmup, 2 * T))
            fence()
            dt_c, dt_c_ev, _ = series(args.steps, repeats)
            torch.cuda.synchronize()
            ok = True
            if gib["last_full"] is not None and gib["n"] - gib["last_full"] <= NSETS:
                ib = gib["last_full"]
                for j, t_ in enumerate(sharding.rooted_range(T, world, rank)):
                    sc = b.parse_scalars(scal[ib % NSETS][j * sstride: (j + 1) * sstride].cpu().numpy(), nrows)
                    ok &= bool(np.array_equal(sc["lag"][1:], params.d))
                    ok &= int(pk_view[ib % NSETS][j][:4].cpu().numpy().view(np.uint32)[0]) == ib * T + t_
            flag = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            c_res = {"status": "ok", "blocks_per_s": args.steps * repeats / dt_c, "ms_per_step": 1e3 * dt_c / (args.steps * repeats),
                     "assembled": bool(flag.item()), "transport": "crsdr_exchange_batch (staged): grouped ncclSend/ncclRecv + crsdr assembly, one side stream"}
            c_fail = not bool(flag.item())
        except Exception as e:
            c_res, c_fail = {"status": f"error: {e}"}, True
        wd.cancel()
        if rank == 0:
            print(json.dumps({"c_exchange": c_res}), file=sys.stderr, flush=True)
        xc, xchg["c"] = xchg["c"], None
        if xc is not None:
            xc.close()
        if c_fail:
            plan.close()
            dist.destroy_process_group()
            sys.exit(4)

    plan.close()
    if multi:
        dist.destroy_process_group()


def _make_slab(synth, params, nsig, L, seed, t, slab):
    """Only row 0 and this rank's slab are generated; other rows stay zero (never read)."""
    import numpy as np
    sub = synth.RowParams.__new__(synth.RowParams)
    lo, hi = slab.row_begin - 1, slab.row_begin - 1 + slab.row_count
    sub.nsig, sub.L, sub.seed, sub.dmax = slab.row_count, L, seed, params.dmax
    sub.d, sub.c, sub.s, sub.phi, sub.g = params.d[lo:hi], params.c[lo:hi], params.s[lo:hi], params.phi[lo:hi], params.g[lo:hi]
    part, _ = synth.make_block(slab.row_count, L, seed, t, params=sub)
    rows = np.zeros((nsig + 1, 2 * L), dtype=np.int8)
    rows[0] = part[0]
    rows[slab.row_begin: slab.row_begin + slab.row_count] = part[1:]
    return rows, None


def _env(torch, dev, b, local_rank):
    """GPU and host the numbers were taken on (clocks recorded)."""
    p = torch.cuda.get_device_properties(dev)
    cpu = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    info = b.device_info(local_rank)
    return {"gpu": info["name"], "arch": getattr(p, "gcnArchName", None), "cus": p.multi_processor_count,
            "gpu_max_clock_mhz": info["clock_mhz"], "hbm_clock_mhz": info["memory_clock_mhz"], "hbm_gib": round(p.total_memory / 2**30, 1),
            "cpu": cpu, "host_cores": len(os.sched_getaffinity(0)), "torch": torch.__version__}


def _committed_traffic(kernel, blocks_per_launch=None):
    """(HBM bytes per launch, where the figure comes from) from the committed rocprofv3 --pmc passes
    (profiles/*traffic*.json), else (None, None).  NOT measured in this run: the passes record how many blocks their
    launches carried and the figure is scaled to this run's blocks per launch (the kernels move the same bytes per block
    whatever the batch length); the source string names the file and both block counts."""
    import glob
    best, src = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json"))):
        try:
            d = json.load(open(f))
            if kernel in d:
                e = d[kernel]
                best = e["bytes_per_launch"] if isinstance(e, dict) else e
                src = f"{os.path.relpath(f, ROOT)} (committed rocprofv3 --pmc pass, not this run)"
                if isinstance(e, dict) and blocks_per_launch and e.get("blocks_per_launch"):
                    best = int(round(best * blocks_per_launch / e["blocks_per_launch"]))
                    src += f": {e['bytes_per_launch']} B per {e['blocks_per_launch']}-block launch, scaled to {blocks_per_launch} blocks"
        except Exception:
            pass
    return best, src


def _cpu_baseline(args, rows, nrows, B, mode):
    """Time oracle/ (the C restatement of the reference's path; kind "port") on this host:
    1 thread = the reference's configuration (one ccoherent thread, no FFTW threads), plus an
    all-cores row.  Bounded sample of the same workload (~10-20 s)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    refl = O.lib_refflags()                 # the reference's compiler flags (-O3, no -march: CMakeLists.txt:10); the -mavx2 build is timed beside it
    eng = O.Engine(nrows, B, mode, cdll=refl)
    t0 = time.perf_counter()
    eng.block(rows, want_packet=True)
    one = time.perf_counter() - t0
    nb = args.cpu_blocks or max(1, min(64, int(12.0 / max(one, 1e-3))))
    t0 = time.perf_counter()
    for i in range(nb):
        eng.block(rows, seq=i)
    dt1 = time.perf_counter() - t0
    ncpu = len(os.sched_getaffinity(0))
    nthreads = max(1, min(ncpu, 64))
    nbm = max(2, min(64, int(6.0 / max(one / nthreads, 1e-3))))
    t0 = time.perf_counter()
    engm = O.Engine(nrows, B, mode, cdll=refl)
    for i in range(nbm):
        engm.block(rows, seq=i, nthreads=nthreads)
    dtm = time.perf_counter() - t0
    # the reference as shipped: at most nfft = 8 rows (ref + 7) are cross-correlated per block (src/main.cc:165,
    # src/ccoherent.cc:124); every row still goes through the phase path
    engx = O.Engine(nrows, B, mode)        # same sources, -O3 -mavx2 (the checker's build)
    engx.block(rows)
    nbx = max(1, nb // 2)
    t0 = time.perf_counter()
    for i in range(nbx):
        engx.block(rows, seq=i)
    dtx = time.perf_counter() - t0
    eng8 = O.Engine(nrows, B, mode, nfft_cap=8, cdll=refl)
    eng8.block(rows)
    nb8 = max(2, min(16, int(4.0 / max(one / 8, 1e-3))))
    t0 = time.perf_counter()
    for i in range(nb8):
        eng8.block(rows, seq=i)
    dt8 = time.perf_counter() - t0
    # probe (never fetch) the reference's own numeric libraries on this host; they are absent from this image
    import ctypes.util
    genuine = {lib: bool(ctypes.util.find_library(lib)) for lib in ("fftw3f", "volk")}
    return {"value": nb / dt1, "unit": "blocks/s", "cores": 1, "kind": "port", "host_has_fftw3f_volk": genuine,
            "ref8": {"value": nb8 / dt8, "cores": 1, "sample": f"{nb8} blocks with the reference's nfft = 8 queue cap (7 of the {nrows - 1} "
                                                               f"signal rows get a lag per block, all get the phase path)"},
            "sample": f"{nb} blocks of the same {nrows} x {B // 2} workload, oracle/coherent_oracle.c built with the reference's flags (-O3, no -march), 1 thread "
                      f"(the reference runs one ccoherent thread); reference itself unbuildable here (VOLK/FFTW absent)",
            "flags": "-O3 (CMakeLists.txt:10)", "avx2": {"value": nbx / dtx, "cores": 1, "sample": f"{nbx} blocks, same sources built -O3 -mavx2"},
            "allcores": {"value": nbm / dtm, "cores": nthreads, "sample": f"{nbm} blocks, rows split over {nthreads} pthreads"}}


if __name__ == "__main__":
    main()
