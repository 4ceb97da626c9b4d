"""The committed bench records and profile artefacts keep the shape the driver and the judge read
(one JSON object per bench.py run: the contract keys + `roofline` + `cpu_baseline`; HBM traffic per launch from
the PMC passes under profiles/).  CPU only: nothing here runs the benchmark."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")

CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}
ROOFLINE_KEYS = {"bound", "achieved", "peak", "unit", "frac", "traffic"}
CPU_KEYS = {"value", "unit", "cores", "kind", "sample"}


def _latest_record():
    # from round 2 on: one default-arguments record per round, profiles/rNN_bench_default.json (round 1 kept a numbered history)
    recs = sorted(glob.glob(os.path.join(PROFILES, "r*_bench_default.json")))
    if recs:
        return recs[-1]
    recs = glob.glob(os.path.join(PROFILES, "r*_bench_v*.json"))
    assert recs, "no committed bench record under profiles/"
    return max(recs, key=lambda p: int(re.search(r"_v(\d+)", os.path.basename(p)).group(1)))


def _latest_round():
    return os.path.basename(_latest_record())[:3]


def test_latest_bench_record_keeps_the_contract():
    d = json.load(open(_latest_record()))
    assert CONTRACT_KEYS <= set(d), CONTRACT_KEYS - set(d)
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["data"] == "synthetic"
    assert d["unit"] == "blocks/s" and d["dtype"] == "f32" and d["vs_baseline"] is None       # BASELINE.json: published = {}
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = steps / wall time of the timed region
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.0) < 1e-6
    r = d["roofline"]
    assert ROOFLINE_KEYS <= set(r)
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # achieved = algorithmic bytes per launch / average launch duration (HIP events in the timed region)
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    # one launch reads each int8 input byte of its blocks once: rows x 16384 B x blocks
    cfg = d["config"]
    assert r["algorithmic_bytes_per_launch"] == (cfg["rows"] - 1) * cfg["fft_len"] * r["blocks_per_launch"]
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    assert CPU_KEYS <= set(c) and c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    assert d["lags_exact"] is True
    if "repeats" in d:                                  # r02: the timed region is repeated, value is the median
        lo, hi = d["value_spread"]
        assert d["repeats"] >= 1 and lo <= d["value"] <= hi
        assert d["batches_timed"]["whole"] * d["batches_timed"]["blocks_per_batch"] + d["batches_timed"]["ragged_blocks"] == d["steps"]
        assert r["traffic"] is None or "profiles/" in r["traffic_source"]      # the traffic figure names the committed pass it comes from


def test_committed_traffic_matches_the_pmc_summary():
    rnd = _latest_round()
    t = json.load(open(os.path.join(PROFILES, f"{rnd}_traffic.json")))
    for key in ("k_xcorr_lag", "k_align_fused"):
        e = t[key]
        # FETCH_SIZE / WRITE_SIZE are KiB; the read side is doubled as MI355X_MICROARCH.md prescribes for gfx950
        assert abs(e["bytes_per_launch"] - (2 * e["fetch_size_raw_kib"] + e["write_size_raw_kib"]) * 1024) < 1024     # raw values are rounded to 0.1 KiB
        assert 0.9 < e["bytes_per_launch"] / e["algorithmic_bytes_per_launch"] < 1.25, key     # no wasted re-reads
    summary = open(os.path.join(PROFILES, rnd, "rocprofv3_summary_final.txt" if rnd == "r01" else "rocprofv3_summary.txt")).read()
    for kernel in ("k_xcorr_lag14", "k_align_fused"):
        assert kernel in summary
    # the kernel-trace average for K1 and the bench record's HIP-event average describe the same launches
    m = re.search(r"k_xcorr_lag14[pq]\s+calls=\s*\d+\s+total_ns=\s*\d+\s+avg_ns=\s*([0-9.]+)", summary)
    assert m
    d = json.load(open(_latest_record()))
    assert abs(float(m.group(1)) * 1e-6 / d["roofline"]["avg_launch_ms"] - 1.0) < 0.10
