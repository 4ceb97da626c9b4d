#!/usr/bin/env python3
"""A/B of the phase kernel (K2) in the locked cadence: children with different environments / builds, interleaved rounds.

    python tools/k2_ab.py "CRSDR_K2_NT=0" "CRSDR_K2_NT=1" "CRSDR_K2_NT=3 CRSDR_LIB=/path/other.so" [--rounds 2] [--track]

Each child builds a 1025 x 8192 plan, runs one track batch on synthetic rows (so that the carried lags are the injected
delays: the locked batches then shift every row, like the real steady state), then times locked batches (T = 64, 64 resident
input blocks = 1.07 GB) and prints the phase kernel's mean launch duration from hipEvents on its stream.
--track times the track cadence instead (K1 + K2 per batch) and reports both kernels.
"""
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(track):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    pkg = importlib.import_module("coherent-rtlsdr_amd")
    b, synth = pkg.binding, pkg.synth
    nsig, L, T, ndist = 1024, 8192, 64, 4
    nrows, B = nsig + 1, 2 * L
    dev = torch.device("cuda", 0)
    seed = synth.config_seed(4)
    params = synth.RowParams(nsig, L, seed, locked=os.environ.get("K2AB_ALIGNED") == "1")     # K2AB_ALIGNED=1: all delays 0 (aligned row loads)
    d_in = torch.empty((T, nrows, B), dtype=torch.uint8, device=dev)
    for t in range(ndist):
        rows, _ = synth.make_block(nsig, L, seed, t, params=params)
        d_in[t].copy_(torch.from_numpy(rows.view(np.uint8)))
    for t in range(ndist, T):
        d_in[t].copy_(d_in[t % ndist])
    plan = b.Plan(nrows, B, b.MODE_DIGITAL, max_batch=T)
    stream = torch.cuda.current_stream()
    plan.set_stream(stream.cuda_stream)
    pstride = (plan.packet_bytes + 255) // 256 * 256
    pk = [torch.zeros(pstride * T + 64, dtype=torch.uint8, device=dev) for _ in range(2)]
    off = [(-(p.data_ptr() + plan.matrix_offset)) % 16 for p in pk]
    fl = b.REFNOISE_ENABLED | b.INPUT_READY
    fl_run = fl if track else fl | b.NO_LAG
    if os.environ.get("K2AB_REFNOISE") == "0":       # no reference-row loads, no dot product, no phasor chain: rotate by the carried phasor only
        fl_run &= ~b.REFNOISE_ENABLED

    def batch(i, f):
        plan.bind_packet(pk[i % 2].data_ptr() + off[i % 2], pstride)
        plan.submit(d_in.data_ptr(), seq=i * T, flags=f, nblocks=T, block_stride=nrows * B)

    batch(0, fl)
    out = plan.fetch(want_packet=False)
    assert np.array_equal(out["lag"][1:], params.d)
    # a device just out of idle needs ~40 ms of load before its clocks settle (bench.py, `region_ms`): warm up for ~100 ms
    for i in range(32 if track else 200):
        batch(i, fl_run)
    torch.cuda.synchronize()
    n = 16 if track else 64
    plan.enable_profiling(n, 0xF)
    t0 = time.perf_counter()
    for i in range(n):
        batch(i, fl_run)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k2 = plan.kernel_times_ms(b.KERNEL_ALIGN_QUANT)
    k1 = plan.kernel_times_ms(b.KERNEL_XCORR_LAG)
    rw = 2 * T * nrows * B
    res = {"blocks_per_s": n * T / dt, "k2_ms": float(np.mean(k2)), "k2_min_ms": float(np.min(k2)), "k2_rw_TBs": rw / (float(np.mean(k2)) * 1e-3) / 1e12}
    if len(k1):
        res["k1_ms"] = float(np.mean(k1))
    print(json.dumps(res))


def main():
    if "--child" in sys.argv:
        return child("--track" in sys.argv)
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 2
    args = [a for a in args if not a.isdigit()]
    extra = ["--track"] if "--track" in sys.argv else []
    for r in range(rounds):
        for spec in args:
            env = dict(os.environ)
            for kv in spec.split():
                k, v = kv.split("=", 1)
                env[k] = v
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + extra, env=env, capture_output=True, text=True, timeout=600)
            line = p.stdout.strip().splitlines()[-1] if p.returncode == 0 and p.stdout.strip() else f"FAILED rc={p.returncode} {p.stderr[-400:]}"
            print(f"round {r} [{spec}] {line}", flush=True)


if __name__ == "__main__":
    main()
