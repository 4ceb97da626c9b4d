"""Generates tests/golden/*.npz -- committed fixtures for the coherent-alignment path.

The reference holds no fixtures for this path ("parity unpinned"), and it
cannot be built here, so these vectors come from this repo's own fp64 numpy model
(oracle/model_fp64.py) on seeded synthetic blocks (coherent-rtlsdr_amd/synth.py).  They pin the
C oracle and the HIP path against a fixed, reviewable set of numbers:

  cfg1_{faithful,digital}.npz : config 1 shape (1 ref + 3 signal rows x 8192, four.cfg), 3 blocks
  small_digital.npz           : 1 + 5 rows x 512 samples (B = 1024), 4 blocks, digital mode
  long_digital.npz            : 1 + 2 rows x 32768 samples (B = 65536: the long-block path), 2 blocks, digital mode
  music_ura21.npz             : the beamformer chain on a 7 x 3 URA scene: rows, covariance, singular
                                values, noise-subspace projector and the 40 x 40 MUSIC map, fp64 numpy model (tests/ura.py)

Run:  python tests/golden/make_golden.py   (from the repo root)
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import model_fp64 as M  # noqa: E402

synth = importlib.import_module("coherent-rtlsdr_amd.synth")
HERE = os.path.dirname(os.path.abspath(__file__))


def gen(name, nsig, L, cfg, mode, nblocks, dmax=None):
    seed = synth.config_seed(cfg)
    params = synth.RowParams(nsig, L, seed, dmax=dmax)
    m = M.Model(nsig + 1, 2 * L, mode)
    rows_all, lag, mag, frac, ph, mat = [], [], [], [], [], []
    for t in range(nblocks):
        rows, _ = synth.make_block(nsig, L, seed, t, params=params)
        o = m.block(rows)
        rows_all.append(rows); lag.append(o[0]); mag.append(o[1]); frac.append(o[2]); ph.append(o[3]); mat.append(o[4])
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        rows=np.stack(rows_all), lag=np.stack(lag).astype(np.int32), mag=np.stack(mag), frac=np.stack(frac),
        phasor=np.stack(ph), matrix=np.stack(mat), mode=np.int32(mode), seed=np.int64(seed),
        d=params.d, phi=params.phi, g=params.g)
    print(name, "lags", lag[-1], "d", params.d)


if __name__ == "__main__":
    gen("cfg1_faithful", 3, 8192, 1, M.FAITHFUL, 3)
    gen("cfg1_digital", 3, 8192, 1, M.DIGITAL, 3)
    gen("small_digital", 5, 512, 101, M.DIGITAL, 4, dmax=100)
    gen("long_digital", 2, 1 << 15, 102, M.DIGITAL, 2, dmax=3000)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ura  # noqa: E402
    rng = np.random.default_rng(2024)
    rows = ura.scene(4096, [(0.8, 1.7, 1.0), (2.1, 0.9, 0.6)], rng)
    R, U, s, pm = ura.music_fp64(rows, 2, ncx=40, ncy=40)
    Un = U[:, 2:]
    np.savez_compressed(os.path.join(HERE, "music_ura21.npz"), rows=rows, rxx=R, sv=s, projector=Un @ Un.conj().T, pm=pm,
                        k=np.int32(2), d=np.float32(ura.D), mx=np.int32(ura.MX), my=np.int32(ura.MY))
    print("music_ura21 sv", s[:4])
