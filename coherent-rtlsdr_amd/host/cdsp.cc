// cdsp.cc -- see cdsp.h.  Every method forwards to the per-op C ABI (include/crsdr.h).
#include "cdsp.h"
#include "../../include/crsdr.h"
#include <cmath>
#include <cstdio>

static inline void check(int rc, const char *what)
{
    if (rc != CRSDR_OK) std::fprintf(stderr, "cdsp::%s: %s\n", what, crsdr_last_error());
}

void cdsp::convtosigned(const uint8_t *in, const uint8_t *out, int n)
{
    check(crsdr_convtosigned(in, const_cast<uint8_t *>(out), n), "convtosigned");  // src/cdsp.cc:21-34 writes through `out`
}
const float *cdsp::convtofloat(const float *out, const int8_t *s8bit, int n)
{
    check(crsdr_convtofloat(const_cast<float *>(out), s8bit, n), "convtofloat");
    return out;
}
const std::complex<float> *cdsp::convtofloat(const std::complex<float> *out, const int8_t *s8bit, int n)
{
    check(crsdr_convtofloat(reinterpret_cast<float *>(const_cast<std::complex<float> *>(out)), s8bit, n), "convtofloat");
    return out;
}
const std::complex<float> *cdsp::scalarmul(const std::complex<float> *out, const std::complex<float> *in,
                                           const std::complex<float> scalar_in, int n)
{
    check(crsdr_scalarmul(reinterpret_cast<float *>(const_cast<std::complex<float> *>(out)), reinterpret_cast<const float *>(in),
                          scalar_in.real(), scalar_in.imag(), n), "scalarmul");
    return out;
}
const std::complex<int8_t> *cdsp::convto8bit(std::complex<int8_t> *out, std::complex<float> *in, int n)
{
    check(crsdr_convto8bit(reinterpret_cast<int8_t *>(out), reinterpret_cast<const float *>(in), n), "convto8bit");
    return out;
}
const std::complex<float> cdsp::conj_dotproduct(const std::complex<float> *a, const std::complex<float> *b, int n)
{
    float r[2] = {0.f, 0.f};
    check(crsdr_conj_dotproduct(r, reinterpret_cast<const float *>(a), reinterpret_cast<const float *>(b), n), "conj_dotproduct");
    return std::complex<float>(r[0], r[1]);
}
const float cdsp::rms(const float *in, int n)
{
    // src/cdsp.cc:68-73: sqrt(dot(in,in)/n); the real dot product is the re part of the complex
    // one over n/2 interleaved pairs: sum (a^2 + b^2)
    if (n & 1) { double s = 0; for (int i = 0; i < n; ++i) s += (double)in[i] * in[i]; return (float)std::sqrt(s / n); }
    std::complex<float> r = conj_dotproduct(reinterpret_cast<const std::complex<float> *>(in),
                                            reinterpret_cast<const std::complex<float> *>(in), n / 2);
    return std::sqrt(r.real() / n);
}
const float cdsp::rms(const std::complex<float> *s, int n)
{
    std::complex<float> res = conj_dotproduct(s, s, n);   // src/cdsp.cc:75-78
    return std::sqrt(res.real() / n);
}
const float cdsp::PAPR(const std::complex<float> *, const std::complex<float> *, int) { return 0; } // src/cdsp.cc:85-88
const float cdsp::crestfactor(const float *in, float peak, int n) { return peak / rms(in, n); }      // src/cdsp.cc:80-83
const float cdsp::crestfactor(const float *in, int n)
{
    uint32_t idx = indexofmax(const_cast<float *>(in), n);                                           // src/cdsp.cc:90-98
    return in[idx] / rms(in, n);
}
const float *cdsp::magsquared(float *out, const std::complex<float> *in, int n)
{
    check(crsdr_magsquared(out, reinterpret_cast<const float *>(in), n), "magsquared");
    return out;
}
const std::complex<float> *cdsp::conjugatemul(std::complex<float> *out, std::complex<float> *in1, std::complex<float> *in2, int n)
{
    check(crsdr_conjugatemul(reinterpret_cast<float *>(out), reinterpret_cast<const float *>(in1), reinterpret_cast<const float *>(in2), n),
          "conjugatemul");
    return out;
}
const std::complex<float> *cdsp::fft(std::complex<float> *out, std::complex<float> *in, fft_scheme *scheme)
{
    const crsdr_fft_scheme *s = *scheme;
    check(crsdr_fft(reinterpret_cast<float *>(out), reinterpret_cast<const float *>(in), s->n, s->sign, s->howmany), "fft");
    return out;
}
const std::complex<float> *cdsp::fft(fft_scheme *scheme)
{
    crsdr_fft_scheme *s = *scheme;                        // src/cdsp.cc:122-133 executes the plan's own buffers, returns NULL
    check(crsdr_fft(reinterpret_cast<float *>(s->out), reinterpret_cast<const float *>(s->in), s->n, s->sign, s->howmany), "fft");
    return nullptr;
}
const uint32_t cdsp::indexofmax(float *in, int n)
{
    uint32_t idx = 0;
    check(crsdr_indexofmax(&idx, in, n), "indexofmax");
    return idx;
}
const uint32_t cdsp::indexofmax(float *out, std::complex<float> *in, int n)
{
    magsquared(out, in, n);                               // src/cdsp.cc:141-146
    return indexofmax(out, n);
}
