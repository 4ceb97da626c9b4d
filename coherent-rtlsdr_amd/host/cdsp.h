// cdsp.h -- the reference's DSP operator facade (include/cdsp.h:36-71 of mlaaks/coherent-rtlsdr),
// same class name, static methods, argument meaning and "returns its out pointer, cannot fail"
// behaviour, re-targeted from VOLK / FFTW to the C ABI of libcrsdr.so (HIP kernels for gfx950).
// The reference invites exactly this: "currently wraps to volk kernels. In future, these could
// be mapped to custom code" (src/cdsp.cc:19).  A failing device call prints the ABI's error text
// to stderr and returns `out` untouched (the reference's methods have no error path either).
#ifndef CDSPH
#define CDSPH
#include <complex>
#include <cstdint>

// stands in for `typedef fftwf_plan fft_scheme` (include/cdsp.h:23-32): the batched 1-D plan
// geometry of src/ccoherent.cc:78-93 (rank 1, n, howmany, stride 1, dist n, sign)
struct crsdr_fft_scheme {
    int n, howmany, sign;                 // sign -1 = FFTW_FORWARD, +1 = FFTW_BACKWARD
    std::complex<float> *in, *out;        // used by the fft(scheme*) overload
};
typedef crsdr_fft_scheme *fft_scheme;

class cdsp {
public:
    static void convtosigned(const uint8_t *in, const uint8_t *out, int n);                                   // include/cdsp.h:40
    static const float *convtofloat(const float *out, const int8_t *s8bit, int n);                          // :41
    static const std::complex<int8_t> *convto8bit(std::complex<int8_t> *out, std::complex<float> *in, int n); // :43
    static const std::complex<float> *convtofloat(const std::complex<float> *out, const int8_t *s8bit, int n); // :45
    static const std::complex<float> *scalarmul(const std::complex<float> *out, const std::complex<float> *in,
                                                const std::complex<float> scalar_in, int n);                  // :47
    static const std::complex<float> conj_dotproduct(const std::complex<float> *a, const std::complex<float> *b, int n); // :49
    static const float rms(const float *in, int n);                                                          // :52
    static const float rms(const std::complex<float> *in, int n);                                            // :53
    static const float PAPR(const std::complex<float> *s, const std::complex<float> *ref, int n);            // :55 (returns 0 upstream too)
    static const float crestfactor(const float *in, float peak, int n);                                      // :57
    static const float crestfactor(const float *in, int n);                                                  // :58
    static const float *magsquared(float *out, const std::complex<float> *in, int n);                        // :61
    static const std::complex<float> *conjugatemul(std::complex<float> *out, std::complex<float> *in1, std::complex<float> *in2, int n); // :63
    static const std::complex<float> *fft(std::complex<float> *out, std::complex<float> *in, fft_scheme *scheme); // :65
    static const std::complex<float> *fft(fft_scheme *scheme);                                               // :66
    static const uint32_t indexofmax(float *in, int n);                                                      // :68
    static const uint32_t indexofmax(float *out, std::complex<float> *in, int n);                            // :69
};
#endif
