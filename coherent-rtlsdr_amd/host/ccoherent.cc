// ccoherent.cc -- see ccoherent.h.
#include "ccoherent.h"
#include <cstdio>
#include <cstring>

ccoherent::ccoherent(crefsdr *refdev_, lvector<csdrdevice *> *devvec_, crefnoise *refnoise_, int nfft_, int mode_)
    : devices(devvec_), refdev(refdev_), refnoise(refnoise_), nfft(nfft_), mode(mode_), plan(nullptr), rows(nullptr), packet(nullptr), packet_bytes(0), seq(0), do_exit(false)
{
    blocksize = (int)refdev->get_blocksize();                 // src/ccoherent.cc:43
    nrows = 1 + (int)devices->size();
    crsdr_plan_desc d;
    std::memset(&d, 0, sizeof(d));
    d.nrows = nrows; d.blocksize = blocksize; d.mode = mode; d.device = 0;
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
