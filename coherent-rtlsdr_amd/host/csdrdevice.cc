// csdrdevice.cc -- see csdrdevice.h.
#include "csdrdevice.h"
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
int8_t *csyntheticsdr::read()
{
    cur = src->row(rowindex);                                // crtlsdr::read src/crtlsdr.cc:195-203 without the cv wait
    inc_readcnt();
    return cur;
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
