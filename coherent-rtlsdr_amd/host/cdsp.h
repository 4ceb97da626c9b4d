// cdsp.h -- the reference's DSP operator facade (include/cdsp.h:36-71 of mlaaks/coherent-rtlsdr): same class
// name, same static methods, same argument meaning and the same "returns its out pointer, cannot fail"
// behaviour, re-targeted from VOLK / FFTW to the C ABI of libcrsdr.so (HIP kernels for gfx950).  The reference
// invites exactly this: "currently wraps to volk kernels. In future, these could be mapped to custom code"
// (src/cdsp.cc:19).  A failing device call prints the ABI's error text to stderr and returns `out` untouched
// (the reference's methods have no error path either).
#ifndef CDSPH
#define CDSPH
#include <complex>
#include <cstdint>

typedef std::complex<float> cf32;      // layout-identical to lv_32fc_t / fftwf_complex (src/ccoherent.cc:65)
typedef std::complex<int8_t> ci8;

// Stands in for `typedef fftwf_plan fft_scheme` (include/cdsp.h:23-32): the batched 1-D plan geometry of
// src/ccoherent.cc:78-93 (rank 1, n, howmany, stride 1, dist n, sign).
struct crsdr_fft_scheme {
    int n, howmany, sign;              // sign -1 = FFTW_FORWARD, +1 = FFTW_BACKWARD
    cf32 *in, *out;                    // used by the fft(scheme*) overload
};
typedef crsdr_fft_scheme *fft_scheme;

class cdsp {
public:
    // -- sample format (n units differ, as upstream: bytes / int8 count / complex count) -- include/cdsp.h:40-45
    static void        convtosigned(const uint8_t *in, const uint8_t *out, int n);     // x ^ 0x80, n bytes, n % 8 == 0
    static const float *convtofloat(const float *out, const int8_t *s8bit, int n);     // n = number of int8 values
    static const cf32  *convtofloat(const cf32 *out, const int8_t *s8bit, int n);      // n = number of int8 values
    static const ci8   *convto8bit(ci8 *out, cf32 *in, int n);                         // n = complex count

    // -- element-wise and reductions, n = complex count -- include/cdsp.h:47-63
    static const cf32  *scalarmul(const cf32 *out, const cf32 *in, const cf32 scalar_in, int n);
    static const cf32  *conjugatemul(cf32 *out, cf32 *in1, cf32 *in2, int n);           // in1 * conj(in2)
    static const cf32   conj_dotproduct(const cf32 *a, const cf32 *b, int n);           // sum a * conj(b)
    static const float *magsquared(float *out, const cf32 *in, int n);

    // -- argmax, first strict maximum -- include/cdsp.h:68-69
    static const uint32_t indexofmax(float *in, int n);
    static const uint32_t indexofmax(float *out, cf32 *in, int n);                      // magsquared into out, then argmax

    // -- batched transform -- include/cdsp.h:65-66
    static const cf32  *fft(cf32 *out, cf32 *in, fft_scheme *scheme);
    static const cf32  *fft(fft_scheme *scheme);

    // -- diagnostics the hot path never calls (host arithmetic, like upstream) -- include/cdsp.h:52-58
    static const float  rms(const float *in, int n);
    static const float  rms(const cf32 *in, int n);
    static const float  crestfactor(const float *in, float peak, int n);
    static const float  crestfactor(const float *in, int n);
    static const float  PAPR(const cf32 *s, const cf32 *ref, int n);                    // returns 0 upstream too
};
#endif
