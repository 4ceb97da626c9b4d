#!/usr/bin/env python3
"""Per-call times of the beamformer chain on the 7 x 3 array of the reference's client (beamformclient/heatmap2d2.cpp): covariance of a
22-row packet matrix, noise subspace (21 x 21), 100 x 100 MUSIC scan -- host pointers in and out, as cbeamformer calls them."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
b = importlib.import_module("coherent-rtlsdr_amd.binding")
import ura
rng = np.random.default_rng(1)
rows = ura.scene(8192, [(1.0, 1.3, 1.0)], rng)
def t(f, n=50):
    f(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    return 1e3 * (time.perf_counter() - t0) / n, r
ms_c, rxx = t(lambda: b.covariance(rows))
ms_s, (vec, sv) = t(lambda: b.noisesubspace(rxx))
ms_p, pm = t(lambda: b.pmusic2d(vec, 1, ura.D, ura.MX, ura.MY, 100, 100))
print(f"covariance 22 x 16384 (host in/out) {ms_c:.3f} ms | noise subspace 21 x 21 {ms_s:.3f} ms | MUSIC scan 100 x 100 {ms_p:.3f} ms | chain {ms_c + ms_s + ms_p:.3f} ms per frame")
