// cdsp.cc -- see cdsp.h.  Every method forwards to the per-op C ABI (include/crsdr.h); complex data crosses the
// ABI as interleaved (re, im) floats.
#include "cdsp.h"
#include "../../include/crsdr.h"
#include <cmath>
#include <cstdio>

namespace {
inline void check(int rc, const char *what)
{
    if (rc != CRSDR_OK) std::fprintf(stderr, "cdsp::%s: %s\n", what, crsdr_last_error());
}
inline const float *fl(const cf32 *p) { return reinterpret_cast<const float *>(p); }
inline float *flw(const cf32 *p) { return reinterpret_cast<float *>(const_cast<cf32 *>(p)); }   // upstream writes through const out
} // namespace

// ---- sample format ----------------------------------------------------------------------------------
void cdsp::convtosigned(const uint8_t *in, const uint8_t *out, int n)
{
    check(crsdr_convtosigned(in, const_cast<uint8_t *>(out), n), "convtosigned");   // src/cdsp.cc:21-34 writes through `out`
}
const float *cdsp::convtofloat(const float *out, const int8_t *s8bit, int n)
{
    check(crsdr_convtofloat(const_cast<float *>(out), s8bit, n), "convtofloat");
    return out;
}
const cf32 *cdsp::convtofloat(const cf32 *out, const int8_t *s8bit, int n)
{
    check(crsdr_convtofloat(flw(out), s8bit, n), "convtofloat");
    return out;
}
const ci8 *cdsp::convto8bit(ci8 *out, cf32 *in, int n)
{
    check(crsdr_convto8bit(reinterpret_cast<int8_t *>(out), fl(in), n), "convto8bit");
    return out;
}

// ---- element-wise and reductions ------------------------------------------------------------------------
const cf32 *cdsp::scalarmul(const cf32 *out, const cf32 *in, const cf32 scalar_in, int n)
{
    check(crsdr_scalarmul(flw(out), fl(in), scalar_in.real(), scalar_in.imag(), n), "scalarmul");
    return out;
}
const cf32 *cdsp::conjugatemul(cf32 *out, cf32 *in1, cf32 *in2, int n)
{
    check(crsdr_conjugatemul(flw(out), fl(in1), fl(in2), n), "conjugatemul");
    return out;
}
const cf32 cdsp::conj_dotproduct(const cf32 *a, const cf32 *b, int n)
{
    float r[2] = {0.f, 0.f};
    check(crsdr_conj_dotproduct(r, fl(a), fl(b), n), "conj_dotproduct");
    return cf32(r[0], r[1]);
}
const float *cdsp::magsquared(float *out, const cf32 *in, int n)
{
    check(crsdr_magsquared(out, fl(in), n), "magsquared");
    return out;
}

// ---- argmax ---------------------------------------------------------------------------------------------
const uint32_t cdsp::indexofmax(float *in, int n)
{
    uint32_t idx = 0;
    check(crsdr_indexofmax(&idx, in, n), "indexofmax");
    return idx;
}
const uint32_t cdsp::indexofmax(float *out, cf32 *in, int n)
{
    magsquared(out, in, n);                                // src/cdsp.cc:141-146
    return indexofmax(out, n);
}

// ---- batched transform ----------------------------------------------------------------------------------
const cf32 *cdsp::fft(cf32 *out, cf32 *in, fft_scheme *scheme)
{
    const crsdr_fft_scheme *s = *scheme;
    check(crsdr_fft(flw(out), fl(in), s->n, s->sign, s->howmany), "fft");
    return out;
}
const cf32 *cdsp::fft(fft_scheme *scheme)
{
    crsdr_fft_scheme *s = *scheme;                         // src/cdsp.cc:122-133 executes the plan's own buffers, returns NULL
    check(crsdr_fft(flw(s->out), fl(s->in), s->n, s->sign, s->howmany), "fft");
    return nullptr;
}

// ---- diagnostics (host arithmetic over the device dot product, like upstream over VOLK's) ------------------
const float cdsp::rms(const float *in, int n)
{
    // src/cdsp.cc:68-73: sqrt(dot(in, in) / n); the real dot product is the re part of the complex one over
    // n/2 interleaved pairs: sum (a^2 + b^2)
    if (n & 1) {
        double s = 0;
        for (int i = 0; i < n; ++i) s += (double)in[i] * in[i];
        return (float)std::sqrt(s / n);
    }
    const cf32 *c = reinterpret_cast<const cf32 *>(in);
    return std::sqrt(conj_dotproduct(c, c, n / 2).real() / n);
}
const float cdsp::rms(const cf32 *s, int n) { return std::sqrt(conj_dotproduct(s, s, n).real() / n); }   // src/cdsp.cc:75-78
const float cdsp::crestfactor(const float *in, float peak, int n) { return peak / rms(in, n); }            // src/cdsp.cc:80-83
const float cdsp::crestfactor(const float *in, int n)
{
    return in[indexofmax(const_cast<float *>(in), n)] / rms(in, n);                                         // src/cdsp.cc:90-98
}
const float cdsp::PAPR(const cf32 *, const cf32 *, int) { return 0; }                                      // src/cdsp.cc:85-88
