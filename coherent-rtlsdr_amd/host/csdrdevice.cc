// csdrdevice.cc -- see csdrdevice.h.
#include "csdrdevice.h"
#include <cmath>
#include <cstring>

csdrdevice::csdrdevice(uint32_t asyncbufn_, uint32_t blocksize_, uint32_t samplerate_, uint32_t fcenter_)
{
    asyncbufn = asyncbufn_; blocksize = blocksize_; samplerate = samplerate_; fcenter = fcenter_;
    readcnt = 0;
    lagrequested = true;                                     // src/csdrdevice.cc:30
    lagready = false; synced = false; streaming = false;
    phasecorr = std::complex<float>(1.0f, 0.0f);             // :39-40
    phasecorrprev = std::complex<float>(1.0f, 0.0f);
    sfloat = new std::complex<float>[blocksize]();           // :43-45 zeroed ("important especially for circ. convolution")
}
csdrdevice::~csdrdevice() { delete[] sfloat; }

std::complex<float> csdrdevice::est_phasecorrect(const lv_32fc_t *ref)
{
    const float alpha = 0.5f;                                // src/csdrdevice.cc:58-69
    std::complex<float> correlation = cdsp::conj_dotproduct(sfloat, ref, (blocksize >> 1));
    const float a = std::abs(correlation);
    if (a == 0.0f || a != a) return phasecorr;               // defined policy: hold (the reference goes NaN)
    phasecorr = std::conj(correlation) * (1.0f / a);
    phasecorr = alpha * phasecorr + (1 - alpha) * phasecorrprev;
    phasecorrprev = phasecorr;
    return phasecorr;
}

std::complex<float> *csdrdevice::phasecorrect()
{
    cdsp::scalarmul(sfloat, sfloat, phasecorr, (blocksize >> 1)); // src/csdrdevice.cc:80-84
    return sfloat;
}

void csdrdevice::requestfftblocking()
{
    lagrequested = true;
    lagready = false;
    std::unique_lock<std::mutex> lock(fftmtx);
    fftcv.wait(lock, [this] { return lagready.load(); });
}

void csdrdevice::set_lag(float lag, float mag)
{
    lagp.lag = lag;                                          // include/csdrdevice.h:138-151
    lagp.mag = mag;
    lagp.ts = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
                  std::chrono::high_resolution_clock::now().time_since_epoch()).count();
    {
        std::lock_guard<std::mutex> lock(fftmtx);
        lagready = true;
        lagrequested = false;
    }
    fftcv.notify_all();
}

csynthsource::csynthsource(int nsig_, int L_, uint64_t seed, int dmax, bool locked) : nsig(nsig_), L(L_), block(-1)
{
    params = csynth_params_create(nsig, L, seed, dmax, locked ? 1 : 0);
    rows.resize((size_t)(1 + nsig) * 2 * L);
}
csynthsource::~csynthsource() { csynth_params_destroy(params); }
void csynthsource::advance() { ++block; csynth_make_block(params, block, -1.0, rows.data()); }

csyntheticsdr::csyntheticsdr(csynthsource *s, int row, uint32_t blocksize_, uint32_t samplerate_, uint32_t fcenter_)
    : csdrdevice(0, blocksize_, samplerate_, fcenter_), src(s), rowindex(row), cur(nullptr)
{
    streaming = true;
    devname = "synthetic " + std::to_string(row);
}
csyntheticsdr::~csyntheticsdr() { stop(); }

int8_t *csyntheticsdr::read()
{
    if (ring) {                                              // crtlsdr::read src/crtlsdr.cc:195-203
        std::unique_lock<std::mutex> lock(mtx);
        cv.wait(lock, [this] { return newdata.load() > 0 || do_exit.load(); });
        if (newdata.load() > 0) newdata--;
        cur = reinterpret_cast<int8_t *>(ring->getbufferptr());
        cur_rcnt = ring->get_rcnt();
        held = true;                                         // the engine copies this slot outside the lock: the producer must not reuse it
        return cur;
    }
    cur = src->row(rowindex);
    inc_readcnt();
    return cur;
}

void csyntheticsdr::consume()
{
    if (ring) {
        { std::lock_guard<std::mutex> lock(mtx); if (held) { held = false; ring->consume(); } }
        cv_space.notify_all();
    }
}

void csyntheticsdr::asynch_threadf(csyntheticsdr *d)
{
    // the librtlsdr callback thread: asynch_callback -> swapbuffer -> cbuffer::setbufferptr
    // (src/crtlsdr.cc:61-68,173-193, include/common.h:114-122), here fed by the synthetic source
    std::vector<int8_t> tmp(d->blocksize);
    for (int t = 0; (d->max_blocks == 0 || t < d->max_blocks) && !d->do_exit; ++t) {
        if (d->replay) {
            if (std::fread(tmp.data(), 1, d->blocksize, d->replay) != d->blocksize) break;   // end of the recording
            for (uint32_t i = 0; i < d->blocksize; ++i) tmp[i] = (int8_t)((uint8_t)tmp[i] ^ 0x80u); // the ring store below re-applies it
        } else {
            csynth_make_row(d->src->get_params(), t, d->rowindex, -1.0, tmp.data());
        }
        {
            std::unique_lock<std::mutex> lock(d->mtx);
            if (d->lossless) d->cv_space.wait(lock, [d] { return d->ring->backlog() < d->ring->capacity() || d->do_exit.load(); });
            if (d->do_exit) break;
            if (d->ring->backlog() >= d->ring->capacity()) {  // consumer too slow: a block is lost (README.md:42)
                d->overruns++;
                if (d->held) {
                    // the oldest slot is the one the engine is copying right now (read() handed it out, consume() has
                    // not come yet): overwriting it would tear that block under a valid readcnt -- the NEW block is
                    // dropped instead (its readcnt is skipped, which is how clients see the gap)
                    d->inc_readcnt();
                    lock.unlock();
                    if (d->pace_us) std::this_thread::sleep_for(std::chrono::microseconds(d->pace_us));
                    continue;
                }
                d->ring->consume();                            // nobody holds it: the oldest unread block goes
                if (d->newdata.load() > 0) d->newdata--;
            }
            uint8_t *w = d->ring->writeptr();
            for (uint32_t i = 0; i < d->blocksize; ++i) w[i] = (uint8_t)tmp[i] ^ 0x80u;   // what the dongle delivers: offset binary
            d->ring->commit(d->get_readcnt());
            d->inc_readcnt();
            d->newdata++;
        }
        d->cv.notify_all();
        if (d->pace_us) std::this_thread::sleep_for(std::chrono::microseconds(d->pace_us));
    }
}

void csyntheticsdr::start(int pace_us_, int max_blocks_)
{
    pace_us = pace_us_; max_blocks = max_blocks_;
    ring.reset(new cbuffer(8, blocksize));                    // asyncbufn = 8, include/common.h:30
    do_exit = false;
    streaming = true;
    producer = std::thread(asynch_threadf, this);
}

bool csyntheticsdr::start_replay(const char *path, int pace_us_, int max_blocks_)
{
    replay = std::fopen(path, "rb");
    if (!replay) { std::fprintf(stderr, "%s: cannot open %s\n", devname.c_str(), path); return false; }
    lossless = true;
    start(pace_us_, max_blocks_);
    return true;
}

void csyntheticsdr::stop()
{
    do_exit = true;
    cv.notify_all();
    cv_space.notify_all();
    if (producer.joinable()) producer.join();
    if (replay) { std::fclose(replay); replay = nullptr; }
}

void csyntheticsdr::advance_resampler()
{
    // resampler running at fs (1 + p) for one block: the stream slips p * L samples against the reference
    if (correction == 0.0f || rowindex == 0) return;
    slip += (double)correction * (double)(blocksize >> 1);
    const long whole = (long)(slip >= 0 ? std::floor(slip + 0.5) : -std::floor(-slip + 0.5));
    if (whole != 0) {
        src->set_delay(rowindex - 1, src->get_delay(rowindex - 1) - whole);   // a positive lag (late) is pulled forward
        slip -= (double)whole;
    }
}
const std::complex<float> *csyntheticsdr::convtofloat() { return cdsp::convtofloat(sfloat, cur, blocksize); }
const std::complex<float> *csyntheticsdr::convtofloat(const std::complex<float> *p) { return cdsp::convtofloat(p, cur, blocksize); }
const std::complex<float> *crefsdr::convtofloat()
{
    cdsp::convtofloat(sfloat + (blocksize >> 1), cur, blocksize); // src/crtlsdr.cc:215-218
    return sfloat + (blocksize >> 1);
}
const std::complex<float> *crefsdr::convtofloat(const std::complex<float> *p)
{
    cdsp::convtofloat(p + (blocksize >> 1), cur, blocksize);
    return p + (blocksize >> 1);
}
