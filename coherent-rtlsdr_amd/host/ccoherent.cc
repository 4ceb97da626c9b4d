// ccoherent.cc -- see ccoherent.h.
#include "ccoherent.h"
#include <cstdio>
#include <cstring>

ccoherent::ccoherent(crefsdr *refdev_, lvector<csdrdevice *> *devvec_, crefnoise *refnoise_, int nfft_, int mode_, int batch_)
    : devices(devvec_), refdev(refdev_), refnoise(refnoise_), nfft(nfft_), mode(mode_), plan(nullptr), rows(nullptr), packet(nullptr), packet_bytes(0), seq(0), do_exit(false)
{
    blocksize = (int)refdev->get_blocksize();                 // src/ccoherent.cc:43
    nrows = 1 + (int)devices->size();
    crsdr_plan_desc d;
    std::memset(&d, 0, sizeof(d));
    d.nrows = nrows; d.blocksize = blocksize; d.mode = mode; d.device = 0;
    d.max_batch = batch_ > 1 ? batch_ : 1;
    batch = d.max_batch;
    if (crsdr_plan_create(&plan, &d) != CRSDR_OK) {
        // same convention as the reference's backend-init failure (src/ccoherent.cc:54-61): print and continue
        std::fprintf(stderr, "ccoherent: %s\n", crsdr_last_error());
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
    if (plan) crsdr_plan_destroy(plan);
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
    bstride = crsdr_plan_packet_stride(plan);
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
    }
    return true;
}

bool ccoherent::fill_batch(int slot, int nblocks)
{
    // the gather of step() (src/ccoherent.cc:250-283), nblocks times into one page-locked slot
    for (int t = 0; t < nblocks; ++t) {
        int8_t *dst = brows[slot] + (size_t)t * nrows * blocksize;
        std::memcpy(dst, refdev->read(), blocksize);
        refdev->consume();
        int c = 1;
        for (auto *d : *devices) {
            std::memcpy(dst + (size_t)c * blocksize, d->read(), blocksize);
            d->consume();
            ++c;
        }
    }
    return true;
}

bool ccoherent::submit_batch(int slot, int nblocks, uint32_t flags)
{
    if (!plan || !brows[0] || nblocks < 1 || nblocks > batch) return false;
    // H2D of this batch shares the link with the D2H of the previous one (crsdr_plan_fetch_batch_async); its kernels queue behind both
    if (crsdr_plan_submit_batch(plan, brows[slot], CRSDR_MEM_HOST, nblocks, 0, nullptr, nullptr, seq, flags) != CRSDR_OK ||
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
    if (crsdr_plan_fetch_wait(plan) != CRSDR_OK) { std::fprintf(stderr, "ccoherent: %s\n", crsdr_last_error()); return false; }
    // (fetch_wait waits for the oldest outstanding fetch only: the batch submitted after this one keeps flying)
    const int t = bcount[slot] - 1;                            // what get_lagp() / get_phasecorrect() show afterwards: the last block
    int c = 1;
    for (auto *d : *devices) {
        d->set_lag((float)batch_lag(slot, t)[c], bmag[slot][(size_t)t * nrows + c]);          // src/ccoherent.cc:232-233
        d->set_phasecorrect(std::complex<float>(batch_phasor(slot, t)[2 * c], batch_phasor(slot, t)[2 * c + 1]));
        ++c;
    }
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
void ccoherent::start() { thread = std::thread(&ccoherent::threadf, this); }
void ccoherent::request_exit() { do_exit = true; }
void ccoherent::join() { thread.join(); }
