// ccoherent.cc -- see ccoherent.h.
#include "ccoherent.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>

ccoherent::ccoherent(crefsdr *refdev_, lvector<csdrdevice *> *devvec_, crefnoise *refnoise_, int nfft_, int mode_, int batch_, const ccoherent_shard *shard)
    : devices(devvec_), refdev(refdev_), refnoise(refnoise_), nfft(nfft_), mode(mode_), plan(nullptr), rows(nullptr), packet(nullptr), packet_bytes(0), seq(0), do_exit(false)
{
    blocksize = (int)refdev->get_blocksize();                 // src/ccoherent.cc:43
    nrows = 1 + (int)devices->size();
    crsdr_plan_desc d;
    std::memset(&d, 0, sizeof(d));
    d.nrows = nrows; d.blocksize = blocksize; d.mode = mode; d.device = shard ? shard->device : 0;
    d.max_batch = batch_ > 1 ? batch_ : 1;
    batch = d.max_batch;
    per = nrows - 1;
    if (shard && shard->id) {                                  // one process per GPU: this one owns rank's slab of the signal rows
        ranks = shard->ranks; rank = shard->rank;
        if (ranks < 1 || rank < 0 || rank >= ranks || (nrows - 1) % ranks) {
            std::fprintf(stderr, "ccoherent: %d signal rows do not split over %d ranks (rank %d)\n", nrows - 1, ranks, rank);
            return;
        }
        per = (nrows - 1) / ranks;
        row_begin = 1 + rank * per;
        d.row_begin = row_begin; d.row_count = per;
    }
    if (crsdr_plan_create(&plan, &d) != CRSDR_OK) {
        // same convention as the reference's backend-init failure (src/ccoherent.cc:54-61): print and continue
        std::fprintf(stderr, "ccoherent: %s\n", crsdr_last_error());
        plan = nullptr;
    }
    if (plan && shard && shard->id &&
        (crsdr_exchange_create(&xchg, shard->id, ranks, rank, shard->device) != CRSDR_OK || crsdr_exchange_bind_plan(xchg, plan, shard->xmode) != CRSDR_OK)) {
        std::fprintf(stderr, "ccoherent: %s\n", crsdr_last_error());
        if (xchg) crsdr_exchange_destroy(xchg);
        xchg = nullptr;
        crsdr_plan_destroy(plan);
        plan = nullptr;
    }
    readcnt.resize(nrows); mask.resize(nrows); lag.resize(nrows); mag.resize(nrows); frac.resize(nrows);
    phasor.resize(2 * (size_t)nrows);
    packet_bytes = plan ? crsdr_plan_packet_bytes(plan) : 0;
    // volk_malloc'd / fftwf_alloc'd in the reference (src/ccoherent.cc:44-47,66-69); page-locked here so that the
    // block goes over PCIe asynchronously at full rate
    if (plan && (crsdr_host_alloc((void **)&rows, (size_t)nrows * blocksize) != CRSDR_OK ||
                 crsdr_host_alloc((void **)&packet, packet_bytes) != CRSDR_OK)) {
        std::fprintf(stderr, "ccoherent: %s\n", crsdr_last_error());
        crsdr_plan_destroy(plan);
        plan = nullptr;
    }
}

ccoherent::~ccoherent()
{
    if (thread.joinable()) thread.join();
    if (xchg) crsdr_exchange_destroy(xchg);
    if (plan) crsdr_plan_destroy(plan);
    for (int i = 0; i < 2; ++i) { crsdr_host_free(bscal[i]); crsdr_host_free(btails[i]); }
    crsdr_host_free(rows);
    crsdr_host_free(packet);
    for (int i = 0; i < 2; ++i) {
        crsdr_host_free(brows[i]); crsdr_host_free(bpackets[i]); crsdr_host_free(blag[i]);
        crsdr_host_free(bmag[i]); crsdr_host_free(bfrac[i]); crsdr_host_free(bphasor[i]);
    }
}

bool ccoherent::enable_batching(int T)
{
    if (!plan || T < 1 || T > batch) return false;
    if (brows[0]) return true;
    // host packets sit a whole reference-length message apart -- (16 + 4N) + 2 N blocksize by default (src/cpacketizer.cc:91-96) --
    // and the memory starts out zero, so that a block's packet is published where it landed, zero tail included, without a copy
    const size_t moff = crsdr_plan_matrix_offset(plan);
    const size_t msg = cpacketize::refpadding ? moff + 2 * (size_t)nrows * (size_t)blocksize : packet_bytes;
    bstride = (std::max(msg, packet_bytes) + 255) / 256 * 256;
    const size_t n = (size_t)nrows, Tz = (size_t)batch;
    for (int i = 0; i < 2; ++i) {
        if (crsdr_host_alloc((void **)&brows[i], Tz * n * (size_t)blocksize) != CRSDR_OK ||
            crsdr_host_alloc((void **)&bpackets[i], Tz * bstride) != CRSDR_OK ||
            crsdr_host_alloc((void **)&blag[i], Tz * n * sizeof(int32_t)) != CRSDR_OK ||
            crsdr_host_alloc((void **)&bmag[i], Tz * n * sizeof(float)) != CRSDR_OK ||
            crsdr_host_alloc((void **)&bfrac[i], Tz * n * sizeof(float)) != CRSDR_OK ||
            crsdr_host_alloc((void **)&bphasor[i], Tz * n * 2 * sizeof(float)) != CRSDR_OK) {
            std::fprintf(stderr, "ccoherent: %s\n", crsdr_last_error());
            return false;
        }
        std::memset(bpackets[i], 0, Tz * bstride);
        breadcnt[i].assign(Tz * n, 0);
        bmask[i].assign(n, 0);
        if (xchg) {
            sstride = (20 * n + 15) / 16 * 16;
            tstride = (24 * (size_t)per + 15) / 16 * 16;      // {lag, mag, frac, phasor, readcnt} of this rank's rows
            if (crsdr_host_alloc((void **)&bscal[i], Tz * sstride) != CRSDR_OK || crsdr_host_alloc((void **)&btails[i], Tz * tstride) != CRSDR_OK) {
                std::fprintf(stderr, "ccoherent: %s\n", crsdr_last_error());
                return false;
            }
        }
    }
    return true;
}

bool ccoherent::fill_batch(int slot, int nblocks, const std::function<void(int)> &before_block)
{
    if (!brows[0] || nblocks < 1 || nblocks > batch) return false;
    std::fill(bmask[slot].begin(), bmask[slot].end(), (uint8_t)0);
    for (int t = 0; t < nblocks; ++t) {
        if (before_block) before_block(t);
        int8_t *dst = brows[slot] + (size_t)t * nrows * blocksize;
        uint32_t *rc = breadcnt[slot].data() + (size_t)t * nrows;
        std::memcpy(dst, refdev->read(), blocksize);                          // src/ccoherent.cc:250
        if (do_exit) return false;                                            // woken by shutdown, not by data: the partial batch is dropped
        rc[0] = refdev->get_readcntbuf();                                     // :253 -- the reference row's own counter
        refdev->consume();
        int c = 1;
        for (auto *d : *devices) {                                            // :262-283
            // sharded: this process hands the plan row 0 and ITS rows only (the plan uploads nothing else, and the other rows' read
            // counters reach the packet header with their owners' tails); every ring is still drained
            const bool own = !xchg || (c >= row_begin && c < row_begin + per);
            const int8_t *src = d->read();
            if (own) {
                std::memcpy(dst + (size_t)c * blocksize, src, blocksize);
                rc[c] = d->get_readcntbuf();                                  // :278 -- what clients detect dropped blocks by (README.md:42)
                if (d->is_lagrequested()) bmask[slot][c] = 1;                 // :266 -- a request made during the batch is served for the whole batch
            }
            d->consume();                                                     // :281
            ++c;
        }
    }
    bool any = false;
    for (int c = 1; c < nrows; ++c) any |= bmask[slot][c] != 0;
    bflags[slot] = refnoise->isenabled() ? CRSDR_REFNOISE_ENABLED : 0;        // :271
    if (refdev->is_streaming_raw()) bflags[slot] |= CRSDR_OFFSET_BINARY;      // ring holds raw uint8: XOR fused into the loads
    if (!any) { bflags[slot] |= CRSDR_NO_LAG; locked_steps += (uint32_t)nblocks; }   // :284 -- nobody asked: "locked" cadence
    bfilled[slot] = true;
    return true;
}

bool ccoherent::submit_batch(int slot, int nblocks, uint32_t flags)
{
    if (!plan || !brows[0] || nblocks < 1 || nblocks > batch) return false;
    const bool f = bfilled[slot];
    if (xchg) {
        // sharded: the plan writes this rank's rows + tails into the exchange's send slots, the exchange assembles on the rotating root
        if (crsdr_exchange_submit_batch(xchg, brows[slot], CRSDR_MEM_HOST, nblocks, 0, f ? breadcnt[slot].data() : nullptr, f ? bmask[slot].data() : nullptr, seq,
                                        flags | (f ? bflags[slot] : 0u)) != CRSDR_OK) {
            std::fprintf(stderr, "ccoherent: %s\n", crsdr_last_error());
            return false;
        }
        seq += (uint32_t)nblocks;
        bcount[slot] = nblocks;
        return true;
    }
    // H2D of this batch shares the link with the D2H of the previous one (crsdr_plan_fetch_batch_async); its kernels queue behind both
    if (crsdr_plan_submit_batch(plan, brows[slot], CRSDR_MEM_HOST, nblocks, 0, f ? breadcnt[slot].data() : nullptr, f ? bmask[slot].data() : nullptr, seq,
                                flags | (f ? bflags[slot] : 0u)) != CRSDR_OK ||
        crsdr_plan_fetch_batch_async(plan, blag[slot], bmag[slot], bfrac[slot], bphasor[slot], bpackets[slot], bstride) != CRSDR_OK) {
        std::fprintf(stderr, "ccoherent: %s\n", crsdr_last_error());
        return false;
    }
    seq += (uint32_t)nblocks;
    bcount[slot] = nblocks;
    return true;
}

bool ccoherent::collect_batch(int slot)
{
    if (!plan || bcount[slot] < 1) return false;
    if (xchg) {
        int first = 0, count = 0, nb = 0;
        if (crsdr_exchange_fetch_rooted(xchg, bpackets[slot], bstride, bscal[slot], sstride, btails[slot], tstride, &first, &count, &nb) != CRSDR_OK) {
            std::fprintf(stderr, "ccoherent: %s\n", crsdr_last_error());
            return false;
        }
        brooted_first[slot] = first; brooted_count[slot] = count;
        const bool f = bfilled[slot];
        // every block: the lags of THIS rank's rows to the devices this process reads (src/ccoherent.cc:232-233) ...
        for (int t = 0; t < nb; ++t) {
            const int8_t *tl = btails[slot] + (size_t)t * tstride;
            const int32_t *lg = reinterpret_cast<const int32_t *>(tl);
            const float *mg = reinterpret_cast<const float *>(tl + 4 * (size_t)per), *ph = reinterpret_cast<const float *>(tl + 12 * (size_t)per);
            for (int i = 0; i < per; ++i) {
                csdrdevice *d = (*devices)[(size_t)(row_begin - 1 + i)];
                if (!f || bmask[slot][row_begin + i]) d->set_lag((float)lg[i], mg[i]);
                d->set_phasecorrect(std::complex<float>(ph[2 * i], ph[2 * i + 1]));
            }
        }
        // ... and the blocks assembled HERE go out whole (:288): all N rows, all N phase factors
        const size_t msg_bytes = cpacketize::packetlength((uint32_t)nrows, (uint32_t)blocksize);
        for (int j = 0; j < count && f && bpublish; ++j) {
            const std::complex<float> *ph = reinterpret_cast<const std::complex<float> *>(bscal[slot] + (size_t)j * sstride + 12 * (size_t)nrows);
            cpacketize::publish(bpackets[slot] + (size_t)j * bstride, msg_bytes, ph, (size_t)nrows);
            ++published;
        }
        bfilled[slot] = false;
        return true;
    }
    if (crsdr_plan_fetch_wait(plan) != CRSDR_OK) { std::fprintf(stderr, "ccoherent: %s\n", crsdr_last_error()); return false; }
    // (fetch_wait waits for the oldest outstanding fetch only: the batch submitted after this one keeps flying)
    const bool f = bfilled[slot];
    const size_t msg_bytes = cpacketize::packetlength((uint32_t)nrows, (uint32_t)blocksize);
    for (int t = 0; t < bcount[slot]; ++t) {                  // block by block, in order: what the per-block loop would have done
        const int32_t *lg = batch_lag(slot, t);
        const float *mg = bmag[slot] + (size_t)t * nrows, *ph = batch_phasor(slot, t);
        int c = 1;
        for (auto *d : *devices) {
            if (!f || bmask[slot][c]) d->set_lag((float)lg[c], mg[c]);                                   // src/ccoherent.cc:232-233
            d->set_phasecorrect(std::complex<float>(ph[2 * c], ph[2 * c + 1]));
            ++c;
        }
        if (f && bpublish)                                     // :288 -- the plan assembled hdr0 + readcnt + matrix; the zero tail is in place
            { cpacketize::publish(bpackets[slot] + (size_t)t * bstride, msg_bytes, reinterpret_cast<const std::complex<float> *>(ph), (size_t)nrows); ++published; }
    }
    bfilled[slot] = false;
    return true;
}

void ccoherent::clearlagqueue() { lagqueue.clear(); }
size_t ccoherent::lagqueuesize() { return lagqueue.size(); }
void ccoherent::queuelag(csdrdevice *d) { lagqueue.push_back(d); }   // src/ccoherent.cc:123-142 (no copy, no cap)

void ccoherent::computelag()
{
    // src/ccoherent.cc:154-239: the xcorr itself ran inside the plan; hand the results to the devices
    int c = 0;
    for (auto *d : *devices) {
        ++c;
        for (auto *q : lagqueue)
            if (q == d) { d->set_lag((float)lag[c], mag[c]); break; }  // :232-233
    }
    lagqueue.clear();
}

bool ccoherent::step()
{
    if (!plan) return false;
    clearlagqueue();                                           // src/ccoherent.cc:249
    int8_t *refsptr = refdev->read();                          // :250
    std::memcpy(rows, refsptr, blocksize);
    readcnt[0] = refdev->get_readcntbuf();
    queuelag(refdev);                                          // :252
    mask[0] = 0;
    int c = 1;
    for (auto *d : *devices) {                                 // :262-283
        int8_t *ptr = d->read();
        std::memcpy(rows + (size_t)c * blocksize, ptr, blocksize);
        readcnt[c] = d->get_readcntbuf();
        mask[c] = d->is_lagrequested() ? 1 : 0;
        if (mask[c]) queuelag(d);                              // :266-267
        ++c;
    }
    uint32_t flags = refnoise->isenabled() ? CRSDR_REFNOISE_ENABLED : 0;   // :271
    if (refdev->is_streaming_raw()) flags |= CRSDR_OFFSET_BINARY;          // ring holds raw uint8: XOR fused into the loads
    if (lagqueuesize() <= 1) { flags |= CRSDR_NO_LAG; ++locked_steps; }   // :284 -- nobody asked: "locked" cadence
    if (crsdr_plan_submit(plan, rows, CRSDR_MEM_HOST, readcnt.data(), mask.data(), seq++, flags) != CRSDR_OK ||
        crsdr_plan_fetch(plan, lag.data(), mag.data(), frac.data(), phasor.data(), packet) != CRSDR_OK) {
        std::fprintf(stderr, "ccoherent: %s\n", crsdr_last_error());
        return false;
    }
    const size_t moff = crsdr_plan_matrix_offset(plan);
    refdev->packetize.write(0, readcnt[0], packet + moff);          // :253 raw ref row
    refdev->consume();
    c = 1;
    for (auto *d : *devices) {
        d->set_phasecorrect(std::complex<float>(phasor[2 * c], phasor[2 * c + 1]));
        d->packetize.write(c, readcnt[c], packet + moff + (size_t)c * blocksize); // :278 (already rotated + requantised)
        d->packetize.writedebug(c, d->get_phasecorrect());                 // :279
        d->consume();                                                      // :281
        ++c;
    }
    if (lagqueuesize() > 1) computelag();                      // :284-286
    refdev->packetize.notifysend();                            // :288
    return true;
}

void ccoherent::threadf(ccoherent *ctx)
{
    while (!ctx->do_exit)
        if (!ctx->step()) break;
}
void ccoherent::threadf_batched(ccoherent *ctx, int T, int delay_us)
{
    if (!ctx->enable_batching(T)) return;
    int b = 0;
    bool pending = false;
    while (!ctx->do_exit) {
        if (!ctx->fill_batch(b & 1, T) || !ctx->submit_batch(b & 1, T, 0)) break;     // upload of batch b beside the download of batch b - 1
        if (pending && !ctx->collect_batch((b - 1) & 1)) { pending = false; break; }
        pending = true;
        ++b;
        if (delay_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(delay_us));
    }
    if (pending) ctx->collect_batch((b - 1) & 1);
}
void ccoherent::start() { thread = std::thread(&ccoherent::threadf, this); }
void ccoherent::start_batched(int T, int delay_us) { thread = std::thread(&ccoherent::threadf_batched, this, T, delay_us); }
void ccoherent::request_exit() { do_exit = true; }
void ccoherent::join() { thread.join(); }
