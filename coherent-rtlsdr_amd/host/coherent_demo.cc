// coherent_demo.cc -- BASELINE config 1 plumbing on the reference's class surface: 1 ref + nsig
// synthetic channels (four.cfg shape by default) -> ccoherent::step() -> cpacketize sink.
// Exit code 0 iff every lag equals its injected delay and, in digital mode, the EMA phasors
// converge to exp(-j phi_k).  --dump writes the first generated block for the csynth/synth.py
// bit-exactness test; --cdsp runs a few class-cdsp identities through the per-op ABI.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>

#include "cbeamformer.h"
#include "ccoherent.h"
#include "ccontrol.h"

static int cdsp_selftest()
{
    const int n = 1024;
    std::vector<int8_t> i8(2 * n);
    for (int i = 0; i < 2 * n; ++i) i8[i] = (int8_t)((i * 37) % 251 - 125);
    std::vector<std::complex<float>> x(n), y(n);
    cdsp::convtofloat(x.data(), i8.data(), 2 * n);
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += (x[i].real() != (float)i8[2 * i] * (1.0f / 127.0f));
    cdsp::scalarmul(y.data(), x.data(), std::complex<float>(1.0f, 0.0f), n);          // identity rotation
    std::vector<std::complex<int8_t>> q(n);
    cdsp::convto8bit(q.data(), y.data(), n);
    for (int i = 0; i < 2 * n; ++i) bad += (reinterpret_cast<int8_t *>(q.data())[i] != i8[i]); // int8 -> float -> int8 round trip
    std::complex<float> e = cdsp::conj_dotproduct(x.data(), x.data(), n);
    float ref = 0; for (int i = 0; i < n; ++i) ref += std::norm(x[i]);
    bad += (std::fabs(e.real() - ref) > 1e-3f * ref) + (std::fabs(e.imag()) > 1e-3f * ref);
    crsdr_fft_scheme sch = {n, 1, -1, nullptr, nullptr}, isch = {n, 1, +1, nullptr, nullptr};
    fft_scheme f = &sch, fi = &isch;
    std::vector<std::complex<float>> X(n), xb(n);
    cdsp::fft(X.data(), x.data(), &f);
    cdsp::fft(xb.data(), X.data(), &fi);
    float err = 0, nrm = 0;
    for (int i = 0; i < n; ++i) { err += std::norm(xb[i] / (float)n - x[i]); nrm += std::norm(x[i]); }
    bad += (std::sqrt(err / nrm) > 1e-6f);
    std::vector<float> m(n);
    uint32_t idx = cdsp::indexofmax(m.data(), X.data(), n);
    uint32_t idx_ref = 0; for (int i = 1; i < n; ++i) if (std::norm(X[i]) > std::norm(X[idx_ref])) idx_ref = i;
    bad += (idx != idx_ref);
    std::printf("cdsp selftest: %s\n", bad ? "FAIL" : "ok");
    return bad;
}

int main(int argc, char **argv)
{
    int nsig = 3, L = 8192, blocks = 12, mode = CRSDR_MODE_DIGITAL, dmax = -1;
    std::string dump, zmqaddr;
    bool run_cdsp = false, servo = false, threads = false, music = false, servo_table = false, bench = false, batch_parity = false, batched = false;
    int batch = 16, engine_delay_ms = 0, pace_us_arg = -1;
    bool engine_batches = false;
    int ranks = 1, rank = 0, device = 0;
    std::string idfile;
    std::vector<double> table_lags;
    int table_fs = 2048000;
    std::string replay;
    int pace_ms = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto val = [&](int &v) { if (i + 1 < argc) v = std::atoi(argv[++i]); };
        if (a == "--nsig") val(nsig);
        else if (a == "--L") val(L);
        else if (a == "--blocks") val(blocks);
        else if (a == "--dmax") val(dmax);
        else if (a == "--faithful") mode = CRSDR_MODE_FAITHFUL;
        else if (a == "--dump" && i + 1 < argc) dump = argv[++i];
        else if (a == "--cdsp") run_cdsp = true;
        else if (a == "--servo") servo = true;      // closed loop: ccontrol model slews the synthetic delays to zero (f3)
        else if (a == "--threads") threads = true;  // streaming: producer thread per device, ccoherent thread, publish loop (f2)
        else if (a == "--zmq" && i + 1 < argc) zmqaddr = argv[++i];           // e.g. tcp://127.0.0.1:5555 (reference: tcp://*:5555)
        else if (a == "--zmq-debug" && i + 1 < argc) cpacketize::debugaddress = argv[++i];
        else if (a == "--pace-ms") val(pace_ms);
        else if (a == "--replay" && i + 1 < argc) { replay = argv[++i]; threads = true; }   // <prefix><row>.u8: raw uint8 IQ recordings (f2)
        else if (a == "--music") music = true;      // f4: the last packet through the beamformer chain (needs nsig = 7 x 3)
        else if (a == "--fs") val(table_fs);
        else if (a == "--bench") bench = true;      // pipelined batched engine over PCIe: blocks/s from C++, no Python in the loop
        else if (a == "--batch") val(batch);
        else if (a == "--batch-parity") batch_parity = true;   // the batched engine's packets == step()'s packets, bit for bit (own read counters, a masked row)
        else if (a == "--batched") batched = true;             // with --threads: the engine thread runs a batch at a time (ccoherent::start_batched)
        else if (a == "--engine-delay-ms") val(engine_delay_ms);
        else if (a == "--pace-us") val(pace_us_arg);
        // one process per GPU: --engine-batches --ranks G --rank r --device d --id-file <path>; start the G processes
        // yourself (a shell loop, mpirun, ...): rank 0 writes the exchange id to the file, the others wait for it
        else if (a == "--engine-batches") engine_batches = true;
        else if (a == "--ranks") val(ranks);
        else if (a == "--rank") val(rank);
        else if (a == "--device") val(device);
        else if (a == "--id-file" && i + 1 < argc) idfile = argv[++i];
        else if (a == "--servo-table" && i + 1 < argc) {      // comma-separated lags
            servo_table = true;
            for (char *tok = std::strtok(argv[++i], ","); tok; tok = std::strtok(nullptr, ",")) table_lags.push_back(std::atof(tok));
        }
    }
    int fails = 0;
    if (servo_table) {
        // f3, numbers only (no device): descent / hold time / hold blocks of the servo for every lag given after the flag,
        // printed with full precision for tests/test_host_cpp.py to compare with its restatement of src/ccontrol.cc
        for (double lg : table_lags) {
            const float lag = (float)lg;
            std::printf("servo lag %.9g correct %d p %.9g t %.17g blocks %d realfs %.17g\n", lag, (int)ccontrol::needs_correction(lag),
                        ccontrol::descent(lag), ccontrol::needs_correction(lag) ? ccontrol::hold_seconds(lag, (uint32_t)table_fs) : 0.0,
                        ccontrol::needs_correction(lag) ? ccontrol::hold_block_count(lag, (uint32_t)table_fs, 2 * (uint32_t)L) : 0,
                        ccontrol::realfs((uint32_t)table_fs));
        }
        return 0;
    }
    if (run_cdsp) fails += cdsp_selftest();

    const uint32_t B = 2 * (uint32_t)L;
    if (batch_parity) {
        // The batched engine against the per-block loop on the same synthetic stream: 3 batches of `batch` blocks.  Row 2's read
        // counter starts three ahead of the others (each device's OWN counter must reach the header, src/cpacketizer.cc:142,163)
        // and row 2 stops asking for a lag after the first batch (the per-device gate of src/ccoherent.cc:266).  Every published
        // message -- header, read counters, matrix, zero tail -- and every phase-factor payload must be equal bit for bit.
        typedef std::vector<std::vector<int8_t>> pkv;
        typedef std::vector<std::vector<std::complex<float>>> phv;
        const int T = batch, NB = 3;
        auto run = [&](bool use_batches, pkv &pk, phv &ph, std::vector<float> &lags) -> bool {
            csynthsource src(nsig, L, csynth_config_seed(1), dmax, false);
            crefsdr rdev(&src, B);
            lvector<csdrdevice *> dv;
            std::vector<std::unique_ptr<csyntheticsdr>> od;
            for (int k = 0; k < nsig; ++k) { od.emplace_back(new csyntheticsdr(&src, 1 + k, B)); dv.push_back(od.back().get()); }
            if (nsig >= 2) for (int i = 0; i < 3; ++i) od[1]->inc_readcnt();
            crefnoise rn;
            cpacketize::init("", false, 1 + nsig, B);
            cpacketize::sink = [&](const int8_t *p, size_t bytes, const std::complex<float> *phase, size_t n) {
                pk.emplace_back(p, p + bytes);
                ph.emplace_back(phase, phase + n);
            };
            ccoherent eng(&rdev, &dv, &rn, 8, mode, use_batches ? T : 1);
            auto request = [&](int block) { for (int k = 0; k < nsig; ++k) if (!(k == 1 && block >= T)) dv[k]->requestfft(); };
            bool ok = true;
            if (!use_batches) {
                for (int t = 0; t < NB * T && ok; ++t) { src.advance(); request(t); ok = eng.step(); cpacketize::send(); }
            } else {
                ok = eng.enable_batching(T);
                for (int bb = 0; bb < NB && ok; ++bb) {
                    request(bb * T);
                    ok = eng.fill_batch(0, T, [&](int) { src.advance(); }) && eng.submit_batch(0, T, 0) && eng.collect_batch(0);
                }
            }
            for (int k = 0; k < nsig; ++k) lags.push_back(dv[k]->get_lagp()->lag);
            cpacketize::sink = nullptr;
            cpacketize::cleanup();
            return ok;
        };
        pkv pa, pb; phv ha, hb; std::vector<float> la, lb;
        const bool oka = run(false, pa, ha, la), okb = run(true, pb, hb, lb);
        size_t bad = (!oka) + (!okb) + (pa.size() != (size_t)(NB * T)) + (pb.size() != pa.size());
        size_t own_cnt = 0, masked_ok = 0;
        for (size_t i = 0; i < pa.size() && i < pb.size(); ++i) {
            if (pa[i] != pb[i]) { if (bad < 5) std::printf("packet %zu differs (%zu vs %zu bytes)\n", i, pa[i].size(), pb[i].size()); ++bad; }
            if (ha[i].size() != hb[i].size() || std::memcmp(ha[i].data(), hb[i].data(), ha[i].size() * sizeof(std::complex<float>))) { if (bad < 5) std::printf("phase payload %zu differs\n", i); ++bad; }
            const uint32_t *rc = reinterpret_cast<const uint32_t *>(pb[i].data() + 16);
            if (nsig >= 2) own_cnt += (rc[2] == rc[1] + 3);              // row 2 carries its own, offset counter
        }
        if (nsig >= 2) bad += (own_cnt != pb.size());
        for (int k = 0; k < nsig; ++k) { bad += (la[k] != lb[k]); masked_ok += 1; }
        std::printf("batch parity: %zu + %zu packets of %zu bytes, %zu with row 2's own read counter, lags %s\n", pa.size(), pb.size(), pa.empty() ? 0 : pa[0].size(), own_cnt,
                    la == lb ? "equal" : "DIFFER");
        std::printf("%s\n", bad ? "DEMO FAILED" : "DEMO OK");
        return bad ? 1 : 0;
    }

    if (engine_batches) {
        // The batched engine as a program: synchronous synthetic source -> fill_batch -> submit_batch -> collect_batch, pipelined over
        // two slots, every block's packet published (sink: one digest line per packet).  With --id-file the signal rows are split over
        // --ranks processes, one per GPU: each owns a slab, the exchange (RCCL under the C ABI) assembles every block on its rotating
        // root, and each rank publishes the blocks it assembled -- the union over the ranks is the unsharded engine's output.
        // (Every rank generates the whole synthetic matrix; only row 0 and its slab go to its GPU.)
        unsigned char xid[CRSDR_EXCHANGE_ID_BYTES];
        ccoherent_shard sh;
        sh.ranks = ranks; sh.rank = rank; sh.device = device;
        if (!idfile.empty()) {
            if (rank == 0) {
                if (crsdr_exchange_unique_id(xid) != CRSDR_OK) { std::printf("exchange id: %s\nDEMO FAILED\n", crsdr_last_error()); return 1; }
                const std::string tmp = idfile + ".tmp";
                FILE *f = std::fopen(tmp.c_str(), "wb");
                if (!f || std::fwrite(xid, 1, sizeof(xid), f) != sizeof(xid)) { std::printf("cannot write %s\nDEMO FAILED\n", tmp.c_str()); return 1; }
                std::fclose(f);
                std::rename(tmp.c_str(), idfile.c_str());              // atomic: a reader never sees half an id
            } else {
                bool got = false;
                for (int spin = 0; spin < 1200 && !got; ++spin) {      // up to 60 s for rank 0 to come up
                    FILE *f = std::fopen(idfile.c_str(), "rb");
                    if (f) { got = std::fread(xid, 1, sizeof(xid), f) == sizeof(xid); std::fclose(f); }
                    if (!got) usleep(50 * 1000);
                }
                if (!got) { std::printf("rank %d: no exchange id in %s\nDEMO FAILED\n", rank, idfile.c_str()); return 1; }
            }
            sh.id = xid;
        }
        csynthsource src(nsig, L, csynth_config_seed(1), dmax, false);
        crefsdr rdev(&src, B);
        lvector<csdrdevice *> dv;
        std::vector<std::unique_ptr<csyntheticsdr>> od;
        for (int k = 0; k < nsig; ++k) { od.emplace_back(new csyntheticsdr(&src, 1 + k, B)); dv.push_back(od.back().get()); }
        crefnoise rn;
        cpacketize::init(zmqaddr, false, 1 + nsig, B);
        size_t npk = 0, badhdr = 0;
        cpacketize::sink = [&](const int8_t *p, size_t bytes, const std::complex<float> *phase, size_t n) {
            const hdr0 *h = reinterpret_cast<const hdr0 *>(p);
            uint64_t fnv = 1469598103934665603ull;
            for (size_t i = 0; i < bytes; ++i) { fnv ^= (uint8_t)p[i]; fnv *= 1099511628211ull; }
            for (size_t i = 0; i < n * sizeof(std::complex<float>); ++i) { fnv ^= reinterpret_cast<const uint8_t *>(phase)[i]; fnv *= 1099511628211ull; }
            std::printf("packet seq %u N %u L %u bytes %zu digest %016llx\n", h->globalseqn, h->N, h->L, bytes, (unsigned long long)fnv);
            badhdr += (h->N != (uint32_t)(1 + nsig)) + (h->L != (uint32_t)L);
            ++npk;
        };
        ccoherent eng(&rdev, &dv, &rn, 8, mode, batch, &sh);
        if (!eng.ok() || !eng.enable_batching(batch)) { std::printf("engine unavailable\nDEMO FAILED\n"); return 1; }
        const int nb = std::max(1, blocks / batch);
        bool ok = true;
        auto fill = [&](int b) { for (auto *d : dv) d->requestfft(); return eng.fill_batch(b & 1, batch, [&](int) { src.advance(); }); };
        ok = fill(0) && eng.submit_batch(0, batch, 0);
        for (int b = 1; b < nb && ok; ++b) {
            ok = fill(b) && eng.submit_batch(b & 1, batch, 0);          // batch b computes while batch b - 1 is exchanged and fetched
            ok = ok && eng.collect_batch((b - 1) & 1);
        }
        ok = ok && eng.collect_batch((nb - 1) & 1);
        const csynth_params *pp = src.get_params();
        int badlag = 0;
        const int per = nsig / ranks, lo = sh.id ? rank * per : 0, hi = sh.id ? lo + per : nsig;
        for (int k = lo; k < hi; ++k) badlag += ((long)dv[k]->get_lagp()->lag != (long)pp->d[k]);      // this process's own devices got their lags
        // blocks this rank assembled: runs of ceil(batch / ranks) per batch
        const int bpr = (batch + ranks - 1) / ranks, mine = std::max(0, std::min(bpr, batch - rank * bpr));
        const size_t expect = (size_t)nb * (size_t)(sh.id ? mine : batch);
        std::printf("engine batches: rank %d of %d on device %d, %d batches x %d blocks, published %zu packets (expected %zu), own lags %s\n", rank, ranks, device, nb, batch,
                    npk, expect, badlag ? "MISMATCH" : "ok");
        const bool fail = !ok || badlag || badhdr || npk != expect;
        cpacketize::sink = nullptr;
        cpacketize::cleanup();
        std::printf("%s\n", fail ? "DEMO FAILED" : "DEMO OK");
        return fail ? 1 : 0;
    }

    csynthsource source(nsig, L, csynth_config_seed(1), dmax, false);
    crefsdr ref(&source, B);
    lvector<csdrdevice *> devs;
    std::vector<std::unique_ptr<csyntheticsdr>> own;
    for (int k = 0; k < nsig; ++k) { own.emplace_back(new csyntheticsdr(&source, 1 + k, B)); devs.push_back(own.back().get()); }
    crefnoise refnoise;
    cpacketize::init(zmqaddr, false, 1 + nsig, B);             // src/main.cc:261 binds tcp://*:5555
    if (!zmqaddr.empty()) { std::printf("zmq publish on %s: %s\n", zmqaddr.c_str(), cpacketize::publishing() ? "bound" : "unavailable"); usleep(300 * 1000); }
    size_t packets = 0, last_bytes = 0; uint32_t last_seq = 0, last_N = 0, last_L = 0;
    std::vector<int8_t> last_packet;
    cpacketize::sink = [&](const int8_t *p, size_t bytes, const std::complex<float> *, size_t) {
        const hdr0 *h = reinterpret_cast<const hdr0 *>(p);
        last_seq = h->globalseqn; last_N = h->N; last_L = h->L; last_bytes = bytes; ++packets;
        if (music) last_packet.assign(p, p + bytes);
    };
    ccoherent coherent(&ref, &devs, &refnoise, 8, mode, (bench || batched) ? batch : 1);

    if (bench) {
        // The engine a batch at a time, pipelined: page-locked slots of `batch` blocks go over PCIe while the previous batch's
        // packets come back (crsdr_plan_submit_batch + crsdr_plan_fetch_batch_async; src/ccoherent.cc:245-294 is the per-block
        // loop this replaces).  Host rows in, host packets out for EVERY block -- the PCIe-inclusive rate, bounded by
        // (1 + nsig) * B bytes each way per block.  The synthetic source is far slower than the link, so four distinct blocks
        // are generated once and cycled through the slots; the transfers and the device work are the real ones.
        if (!coherent.enable_batching(batch)) { std::printf("bench: batching unavailable\nDEMO FAILED\n"); return 1; }
        const size_t blk = (size_t)(1 + nsig) * B;
        std::vector<int8_t> distinct(4 * blk);
        for (int t = 0; t < 4; ++t) { source.advance(); std::memcpy(distinct.data() + t * blk, source.row(0), blk); }
        for (int sl = 0; sl < 2; ++sl)
            for (int t = 0; t < batch; ++t) std::memcpy(coherent.batch_rows(sl) + (size_t)t * blk, distinct.data() + (size_t)((sl * batch + t) % 4) * blk, blk);
        const int nb = std::max(3, blocks / batch);
        const uint32_t fl = CRSDR_REFNOISE_ENABLED;
        bool ok = true;
        auto run = [&](int n) {
            ok = ok && coherent.submit_batch(0, batch, fl);
            for (int b = 1; b < n && ok; ++b) {
                ok = ok && coherent.submit_batch(b & 1, batch, fl);       // upload of batch b overlaps the download of batch b - 1
                ok = ok && coherent.collect_batch((b - 1) & 1);
            }
            ok = ok && coherent.collect_batch((n - 1) & 1);
        };
        run(3);                                                           // warm-up: allocations, code objects, link
        const auto t0 = std::chrono::steady_clock::now();
        run(nb);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (!ok) { std::printf("bench: engine error\nDEMO FAILED\n"); return 1; }
        const csynth_params *p = source.get_params();
        const int last = (nb - 1) & 1;
        int bad = 0;
        for (int t = 0; t < batch; ++t)
            for (int k = 0; k < nsig; ++k) bad += (coherent.batch_lag(last, t)[1 + k] != (int32_t)p->d[k]);
        const hdr0 *h = reinterpret_cast<const hdr0 *>(coherent.batch_packet(last, batch - 1));
        bad += (h->N != (uint32_t)(1 + nsig)) + (h->L != (uint32_t)L);
        const double bps = (double)nb * batch / dt, gbs = bps * (double)blk / 1e9;
        std::printf("bench: %d batches x %d blocks of (1 + %d) x %d in %.3f s: %.1f blocks/s, %.2f GB/s each way over PCIe, lags %s\n", nb, batch, nsig, L, dt,
                    bps, gbs, bad ? "MISMATCH" : "ok");
        cpacketize::cleanup();
        std::printf("%s\n", bad ? "DEMO FAILED" : "DEMO OK");
        return bad ? 1 : 0;
    }

    if (threads) {
        // f2: every device streams from its own producer thread into a ring (crtlsdr::asynch_threadf /
        // swapbuffer / read / consume), the engine runs in its thread (ccoherent::start) and this thread
        // is the publish loop of src/main.cc:277-279.  Per-row readcnt must be continuous in the packets.
        {
            // Real-time sources start below: warm the process first (code object, streams, page-locked staging, the kernels' first
            // launches) on a throwaway plan of the same shape, or a cold first step costs more than the rings' 8 blocks of slack and
            // the run starts with overruns that have nothing to do with the engine.
            crsdr_plan_desc wd;
            std::memset(&wd, 0, sizeof(wd));
            wd.nrows = 1 + nsig; wd.blocksize = B; wd.mode = mode; wd.device = device; wd.max_batch = std::max(1, batch);
            crsdr_plan *w = nullptr;
            if (crsdr_plan_create(&w, &wd) == CRSDR_OK) {
                std::vector<int8_t> z((size_t)(1 + nsig) * B * (size_t)wd.max_batch, 0);
                for (int i = 0; i < 2; ++i) {
                    (void)crsdr_plan_submit_batch(w, z.data(), CRSDR_MEM_HOST, wd.max_batch, 0, nullptr, nullptr, 0, CRSDR_REFNOISE_ENABLED);
                    (void)crsdr_plan_sync(w);
                }
                crsdr_plan_destroy(w);
            }
        }
        std::vector<uint32_t> lastcnt(1 + nsig, 0);
        size_t gaps = 0;
        std::atomic<size_t> npk{0};
        cpacketize::sink = [&](const int8_t *p, size_t, const std::complex<float> *, size_t) {
            const uint32_t *rc = reinterpret_cast<const uint32_t *>(p + 16);
            for (int c = 0; c <= nsig; ++c) { if (npk && rc[c] != lastcnt[c] + 1) ++gaps; lastcnt[c] = rc[c]; }
            ++npk;
        };
        const int pace_us = pace_us_arg >= 0 ? pace_us_arg : pace_ms ? pace_ms * 1000 : 3000;
        if (batched) {
            // the engine thread a batch at a time (ccoherent::start_batched): every block's packet is published from there with
            // each device's own read counter.  With --engine-delay-ms the reader is slower than the producers: rings overrun,
            // blocks are lost, and the ONLY trace of that is a jump in that row's read counter (README.md:42) -- which must
            // therefore show in the headers of the batched engine exactly as it does in step()'s.
            std::vector<uint32_t> first(1 + nsig, 0), last(1 + nsig, 0);
            size_t jumps = 0, backwards = 0, npk2 = 0;
            cpacketize::sink = [&](const int8_t *p, size_t, const std::complex<float> *, size_t) {
                const uint32_t *rc = reinterpret_cast<const uint32_t *>(p + 16);
                for (int c = 0; c <= nsig; ++c) {
                    if (!npk2) first[c] = rc[c];
                    else { jumps += rc[c] > last[c] + 1; backwards += rc[c] <= last[c]; }
                    last[c] = rc[c];
                }
                ++npk2;
            };
            ref.start(pace_us, blocks);
            for (auto &d : own) d->start(pace_us, blocks);
            coherent.start_batched(batch, engine_delay_ms * 1000);
            // wait until the producers are done and less than a batch is left in some ring (the engine cannot fill another one)
            for (int spin = 0; spin < 60000; ++spin) {
                bool done = ref.produced_all(blocks) && ref.backlog() < (uint32_t)batch;
                for (auto &d : own) done = done || (d->produced_all(blocks) && d->backlog() < (uint32_t)batch);
                bool all_prod = ref.produced_all(blocks);
                for (auto &d : own) all_prod = all_prod && d->produced_all(blocks);
                if (all_prod && done) break;
                usleep(1000);
            }
            usleep(200 * 1000);                          // the batch in flight
            coherent.request_exit();
            ref.stop();
            for (auto &d : own) d->stop();
            coherent.join();
            uint32_t over = ref.get_overruns();
            for (auto &d : own) over += d->get_overruns();
            size_t skipped = 0;
            for (int c = 0; c <= nsig; ++c) skipped += npk2 ? (last[c] - first[c] + 1) - npk2 : 0;
            std::printf("streaming batched: %zu packets, %zu read-counter jumps (%zu blocks skipped), %zu backwards, %u ring overruns\n", npk2, jumps, skipped, backwards, over);
            // every skipped count is a block a ring dropped; drops behind the last published block do not show
            fails += (npk2 == 0) + (backwards != 0) + (skipped > over) + (engine_delay_ms > 0 ? (over == 0 || jumps == 0) : (over != 0 || jumps != 0));
            cpacketize::cleanup();
            std::printf("%s\n", fails ? "DEMO FAILED" : "DEMO OK");
            return fails ? 1 : 0;
        }
        if (!replay.empty()) {
            // recorded streams: one raw offset-binary uint8 file per channel, row 0 = the reference-noise channel
            bool ok = ref.start_replay((replay + "0.u8").c_str(), pace_us, blocks);
            for (int k = 0; k < nsig; ++k) ok = own[k]->start_replay((replay + std::to_string(1 + k) + ".u8").c_str(), pace_us, blocks) && ok;
            if (!ok) { std::printf("DEMO FAILED\n"); return 1; }
        } else {
            ref.start(pace_us, blocks);
            for (auto &d : own) d->start(pace_us, blocks);
        }
        coherent.start();
        // a ring that overran has dropped a block: the engine then waits for a block that never comes and this loop for a packet that
        // never comes.  Watch for it -- every producer done, every ring empty, no packet for a second -- and let go, so that a lost block
        // is a FAILED line below, not a hang
        std::atomic<bool> fin{false};
        std::thread guard([&] {
            size_t seen = 0;
            int idle = 0;
            while (!fin.load()) {
                usleep(50 * 1000);
                bool starving = ref.drained((uint32_t)blocks);
                for (auto &d : own) starving = starving && d->drained((uint32_t)blocks);
                if (starving && npk.load() == seen) { if (++idle >= 20) { ref.packetize.request_exit(); break; } }
                else idle = 0;
                seen = npk.load();
            }
        });
        for (int t = 0; t < blocks; ++t)
            if (cpacketize::send() < 0) break;
        fin = true;
        guard.join();
        coherent.request_exit();
        ref.stop();
        for (auto &d : own) d->stop();
        coherent.join();
        const csynth_params *p = source.get_params();
        for (int k = 0; k < nsig; ++k) {
            if (!replay.empty()) std::printf("row %d: lag %ld\n", 1 + k, (long)devs[k]->get_lagp()->lag);   // the recording's truth is the caller's
            else fails += ((long)devs[k]->get_lagp()->lag != (long)p->d[k]);
        }
        uint32_t over = ref.get_overruns();
        for (auto &d : own) over += d->get_overruns();
        std::printf("streaming: %zu packets, %zu readcnt gaps, %u ring overruns, lags %s\n", npk.load(), gaps, over, fails ? "MISMATCH" : "ok");
        fails += (npk.load() != (size_t)blocks) + (gaps != 0) + (over != 0);
        cpacketize::cleanup();
        std::printf("%s\n", fails ? "DEMO FAILED" : "DEMO OK");
        return fails ? 1 : 0;
    }
    if (servo) {
        // f3: closed loop.  Each device's ccontrol reads the lag the engine reports, slews the (modelled)
        // resampler, re-measures, and marks the device synchronized at |lag| <= sync_threshold -- after which
        // nobody requests lags and the engine runs its locked cadence.
        std::vector<ccontrol> ctl;
        for (auto &d : own) ctl.emplace_back(d.get());
        int t = 0, synced_at = -1;
        for (; t < blocks; ++t) {
            source.advance();
            if (!coherent.step()) { std::fprintf(stderr, "step failed\n"); return 2; }
            cpacketize::send();
            int nsync = 0;
            for (size_t k = 0; k < ctl.size(); ++k) { ctl[k].on_block(); nsync += own[k]->get_synchronized(); }
            if (nsync == nsig && synced_at < 0) synced_at = t;
            if (synced_at >= 0 && t >= synced_at + 12) break;   // a dozen locked blocks for the EMA to settle
        }
        const csynth_params *p = source.get_params();
        for (int k = 0; k < nsig; ++k) {
            std::complex<float> ph = devs[k]->get_phasecorrect();
            const double resid = std::arg(std::complex<double>(ph.real(), ph.imag()) * std::polar(1.0, std::atan2(p->s[k], p->c[k])));
            const bool ok = own[k]->get_synchronized() && p->d[k] == 0 && devs[k]->get_lagp()->lag == 0.0f && std::fabs(resid) < 0.02;
            std::printf("row %d: synchronized %d, residual delay %ld, last lag %.0f, residual phase %+.5f rad  %s\n", 1 + k,
                        (int)own[k]->get_synchronized(), (long)p->d[k], devs[k]->get_lagp()->lag, resid, ok ? "ok" : "MISMATCH");
            fails += !ok;
        }
        std::printf("servo: all synchronized after %d blocks, %u locked (phase-only) blocks of %d\n", synced_at, coherent.get_locked_steps(), t + 1);
        fails += (synced_at < 0) + (coherent.get_locked_steps() < 10);
        cpacketize::cleanup();
        std::printf("%s\n", fails ? "DEMO FAILED" : "DEMO OK");
        return fails ? 1 : 0;
    }

    for (int t = 0; t < blocks; ++t) {
        source.advance();
        if (t == 0 && !dump.empty()) {
            FILE *f = std::fopen(dump.c_str(), "wb");
            if (f) { std::fwrite(source.row(0), 1, (size_t)(1 + nsig) * B, f); std::fclose(f); }
        }
        for (auto *d : devs) d->requestfft();                 // "track" cadence: every row asks for a lag every block
        if (!coherent.step()) { std::fprintf(stderr, "step failed\n"); return 2; }
        cpacketize::send();                                    // main-thread publish loop, src/main.cc:277-279
        if (pace_ms) usleep(pace_ms * 1000);
    }
    const csynth_params *p = source.get_params();
    for (int k = 0; k < nsig; ++k) {
        const lagpoint *lp = devs[k]->get_lagp();
        std::complex<float> ph = devs[k]->get_phasecorrect();
        const double phi = std::atan2(p->s[k], p->c[k]);
        const double resid = std::arg(std::complex<double>(ph.real(), ph.imag()) * std::polar(1.0, phi));
        const bool lag_ok = ((long)lp->lag == (long)p->d[k]);
        const bool ph_ok = (mode != CRSDR_MODE_DIGITAL) || (std::fabs(resid) < 0.02);
        std::printf("row %d: lag %6.0f (injected %6ld) mag %10.1f  phasor %+.4f%+.4fj  residual phase %+.5f rad  %s\n", 1 + k,
                    lp->lag, (long)p->d[k], lp->mag, ph.real(), ph.imag(), resid, (lag_ok && ph_ok) ? "ok" : "MISMATCH");
        fails += !(lag_ok && ph_ok);
    }
    if (music) {
        // what the reference's beamformer client does with a packet (beamformclient/heatmap2d2.cpp:185-203).  The
        // synthetic channels all carry the reference noise, so once aligned the array sees one source at
        // broadside: steering vector of all ones, alpha = beta = pi/2, grid point (50, 50).
        cmatrix Rxx, U;
        std::vector<float> S, pm;
        int M = 0;
        if (nsig != cbeamformer::MX * cbeamformer::MY || cbeamformer::covariance(last_packet.data(), Rxx, M) ||
            cbeamformer::noisesubspace(Rxx, M, U, &S) ||
            cbeamformer::pmusic2dvec(U, M, 1, cbeamformer::D, cbeamformer::MX, cbeamformer::MY, cbeamformer::CX, cbeamformer::CY, pm)) {
            std::printf("music: chain failed (needs --nsig 21)\n");
            ++fails;
        } else {
            size_t best = 0;
            for (size_t i = 1; i < pm.size(); ++i) if (pm[i] > pm[best]) best = i;
            const int cx = (int)(best / cbeamformer::CY), cy = (int)(best % cbeamformer::CY);
            std::printf("music: singular values %.4f %.4f .. %.4f, peak (%d, %d) of %d x %d\n", S[0], S[1], S[M - 1], cx, cy, cbeamformer::CX, cbeamformer::CY);
            fails += (mode == CRSDR_MODE_DIGITAL) && !(cx == 50 && cy == 50 && S[0] > 10 * S[1]);
        }
    }
    std::printf("packets sent %zu, last hdr {seq %u, N %u, L %u}, %zu bytes\n", packets, last_seq, last_N, last_L, last_bytes);
    fails += (packets != (size_t)blocks) + (last_N != (uint32_t)(1 + nsig)) + (last_L != (uint32_t)L) + (last_seq != (uint32_t)(blocks - 1));
    cpacketize::cleanup();
    std::printf("%s\n", fails ? "DEMO FAILED" : "DEMO OK");
    return fails ? 1 : 0;
}
