/*
 * coherent_oracle.c -- CPU restatement (plain C, fp32) of the ccoherent/cdsp hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see coherent_oracle.h).  PARITY UNPINNED: the reference has no
 * tests/fixtures for this path and cannot be built here (VOLK / FFTW3f absent, unpinned).
 *
 * Every function cites the reference lines (relative to /root/reference) it follows.
 * VOLK semantics are the published *generic* kernels (one scalar op per element, no SIMD
 * re-association); FFTW semantics are its documented definition: forward sign -1, backward
 * sign +1, both unnormalised.  The FFT itself is this file's own Stockham radix-4/2
 * autosort transform (FFTW's codelets are not restated -- only the transform they compute).
 *
 * Build with -ffp-contract=off so that every fp32 operation below rounds exactly once, in
 * the order written (the HIP kernels are written to the same order where bit parity is
 * claimed: convtofloat, scalarmul, convto8bit).
 */
#include "coherent_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================================
 * per-op restatements
 * ==================================================================================== */

/* src/cdsp.cc:21-34 -- 64-bit XOR with 0x80 in every byte; n>>3 words. */
void orc_convtosigned(const uint8_t *in, uint8_t *out, int n)
{
    int N = n >> 3;
    for (int i = 0; i < N; i++) {
        uint64_t w;
        memcpy(&w, in + 8 * (size_t)i, 8);
        w ^= 0x8080808080808080ull;
        memcpy(out + 8 * (size_t)i, &w, 8);
    }
}

/* src/cdsp.cc:36-44 -> volk_8i_s32f_convert_32f(out, in, 127.0f, n), generic kernel:
 *   const float iScalar = 1.0 / scalar;  out[i] = ((float)in[i]) * iScalar;          */
void orc_convtofloat(float *out, const int8_t *in, int n)
{
    const float iScalar = 1.0f / 127.0f;
    for (int i = 0; i < n; i++) out[i] = ((float)in[i]) * iScalar;
}

/* src/cdsp.cc:46-49 -> volk_32fc_s32fc_multiply_32fc generic: c[i] = a[i] * scalar
 * (complex product, (ar*sr - ai*si) + j(ar*si + ai*sr), each op rounded once). */
void orc_scalarmul(float *out, const float *in, float sr, float si, int n)
{
    for (int i = 0; i < n; i++) {
        float ar = in[2 * i], ai = in[2 * i + 1];
        float re = ar * sr - ai * si;
        float im = ar * si + ai * sr;
        out[2 * i] = re;
        out[2 * i + 1] = im;
    }
}

/* src/cdsp.cc:51-54 -> volk_32f_s32f_convert_8i(out, in, 127.0f, n<<1) generic:
 *   r = in[i]*scalar; if (r > 127) 127; else if (r < -128) -128; else (int8) rintf(r).
 * rintf = round-half-even in the default rounding mode.  NaN fails both compares and the
 * reference then casts rintf(NaN) (undefined; 0 on x86 after truncation): restated as 0. */
void orc_convto8bit(int8_t *out, const float *in, int n)
{
    int m = n << 1;
    for (int i = 0; i < m; i++) {
        float r = in[i] * 127.0f;
        int8_t v;
        if (r > 127.0f) v = 127;
        else if (r < -128.0f) v = -128;
        else if (r != r) v = 0;
        else v = (int8_t)rintf(r);
        out[i] = v;
    }
}

/* src/cdsp.cc:61-66 -> volk_32fc_x2_conjugate_dot_prod_32fc generic:
 *   res = 0; for i: res += a[i] * conj(b[i]);   (one sequential accumulator)          */
void orc_conj_dotproduct(float *res, const float *a, const float *b, int n)
{
    float accr = 0.0f, acci = 0.0f;
    for (int i = 0; i < n; i++) {
        float ar = a[2 * i], ai = a[2 * i + 1];
        float br = b[2 * i], bi = b[2 * i + 1];
        float pr = ar * br + ai * bi;   /* re(a*conj(b)) */
        float pi = ai * br - ar * bi;   /* im(a*conj(b)) */
        accr += pr;
        acci += pi;
    }
    res[0] = accr;
    res[1] = acci;
}

/* src/cdsp.cc:100-103 -> volk_32fc_magnitude_squared_32f generic: re*re + im*im. */
void orc_magsquared(float *out, const float *in, int n)
{
    for (int i = 0; i < n; i++) {
        float re = in[2 * i], im = in[2 * i + 1];
        out[i] = re * re + im * im;
    }
}

/* src/cdsp.cc:105-108 -> volk_32fc_x2_multiply_conjugate_32fc generic: a * conj(b). */
void orc_conjugatemul(float *out, const float *a, const float *b, int n)
{
    for (int i = 0; i < n; i++) {
        float ar = a[2 * i], ai = a[2 * i + 1];
        float br = b[2 * i], bi = b[2 * i + 1];
        float re = ar * br + ai * bi;
        float im = ai * br - ar * bi;
        out[2 * i] = re;
        out[2 * i + 1] = im;
    }
}

/* src/cdsp.cc:135-139 -> volk_32f_index_max_32u generic:
 *   max = src[0]; index = 0; for i>=1: if (src[i] > max) { index = i; max = src[i]; }  */
uint32_t orc_indexofmax(const float *in, int n)
{
    float max = in[0];
    uint32_t index = 0;
    for (int i = 1; i < n; i++) {
        if (in[i] > max) { index = (uint32_t)i; max = in[i]; }
    }
    return index;
}

/* ======================================================================================
 * FFT: what fftwf_plan_many_dft(rank 1, n=B, howmany, stride 1, dist B, sign) computes
 * (src/ccoherent.cc:78-93): X[k] = sum_n x[n] exp(sign * 2*pi*i*n*k/B), unnormalised.
 * Stockham autosort, radix-4 passes plus one radix-2 pass when log2(n) is odd.
 * Twiddles are generated in double and rounded once to float.
 * ==================================================================================== */

typedef struct {
    int n;
    float *w; /* w[2k], w[2k+1] = cos(2 pi k/n), -sin(2 pi k/n), k in [0,n) (forward) */
} orc_twiddle;

static int is_pow2(int n) { return n >= 2 && (n & (n - 1)) == 0; }

static int twiddle_init(orc_twiddle *t, int n)
{
    t->n = n;
    t->w = (float *)malloc(sizeof(float) * 2 * (size_t)n);
    if (!t->w) return -1;
    for (int k = 0; k < n; k++) {
        double a = 2.0 * M_PI * (double)k / (double)n;
        t->w[2 * k] = (float)cos(a);
        t->w[2 * k + 1] = (float)(-sin(a));
    }
    return 0;
}

static void twiddle_free(orc_twiddle *t) { free(t->w); t->w = NULL; }

/* one length-n transform; x is destroyed, result lands in out (work is scratch, n complex) */
static void fft_one(const orc_twiddle *t, float *out, float *x, float *work, int sign)
{
    const int N = t->n;
    const float sg = (sign < 0) ? 1.0f : -1.0f; /* multiplies the stored -sin */
    float *src = x, *dst = work;
    int n = N, s = 1;
    while (n >= 4) {
        const int n1 = n >> 2;
        const int tw = N / n; /* W_n^p = W_N^(p*tw) */
        for (int p = 0; p < n1; p++) {
            const float w1r = t->w[2 * (p * tw)],     w1i = sg * t->w[2 * (p * tw) + 1];
            const float w2r = t->w[2 * (2 * p * tw)], w2i = sg * t->w[2 * (2 * p * tw) + 1];
            const float w3r = t->w[2 * (3 * p * tw)], w3i = sg * t->w[2 * (3 * p * tw) + 1];
            const float *pa = src + 2 * (size_t)s * (p);
            const float *pb = src + 2 * (size_t)s * (p + n1);
            const float *pc = src + 2 * (size_t)s * (p + 2 * n1);
            const float *pd = src + 2 * (size_t)s * (p + 3 * n1);
            float *y0 = dst + 2 * (size_t)s * (4 * p);
            float *y1 = y0 + 2 * (size_t)s;
            float *y2 = y1 + 2 * (size_t)s;
            float *y3 = y2 + 2 * (size_t)s;
            for (int q = 0; q < s; q++) {
                float ar = pa[2 * q], ai = pa[2 * q + 1];
                float br = pb[2 * q], bi = pb[2 * q + 1];
                float cr = pc[2 * q], ci = pc[2 * q + 1];
                float dr = pd[2 * q], di = pd[2 * q + 1];
                float apcr = ar + cr, apci = ai + ci;
                float amcr = ar - cr, amci = ai - ci;
                float bpdr = br + dr, bpdi = bi + di;
                float bmdr = br - dr, bmdi = bi - di;
                /* forward: -j*(b-d) = (bmdi, -bmdr); backward: +j*(b-d) = (-bmdi, bmdr) */
                float jr = sg * bmdi, ji = -sg * bmdr;
                float t1r = amcr + jr, t1i = amci + ji; /* k=1 */
                float t2r = apcr - bpdr, t2i = apci - bpdi; /* k=2 */
                float t3r = amcr - jr, t3i = amci - ji; /* k=3 */
                y0[2 * q] = apcr + bpdr;
                y0[2 * q + 1] = apci + bpdi;
                y1[2 * q] = t1r * w1r - t1i * w1i;
                y1[2 * q + 1] = t1r * w1i + t1i * w1r;
                y2[2 * q] = t2r * w2r - t2i * w2i;
                y2[2 * q + 1] = t2r * w2i + t2i * w2r;
                y3[2 * q] = t3r * w3r - t3i * w3i;
                y3[2 * q + 1] = t3r * w3i + t3i * w3r;
            }
        }
        float *tmp = src; src = dst; dst = tmp;
        n >>= 2;
        s <<= 2;
    }
    if (n == 2) {
        /* last radix-2 pass: twiddle W_2^0 = 1 */
        const float *pa = src, *pb = src + 2 * (size_t)s;
        for (int q = 0; q < s; q++) {
            float ar = pa[2 * q], ai = pa[2 * q + 1];
            float br = pb[2 * q], bi = pb[2 * q + 1];
            dst[2 * q] = ar + br;
            dst[2 * q + 1] = ai + bi;
            dst[2 * (q + s)] = ar - br;
            dst[2 * (q + s) + 1] = ai - bi;
        }
        float *tmp = src; src = dst; dst = tmp;
    }
    if (src != out) memcpy(out, src, sizeof(float) * 2 * (size_t)N);
}

int orc_fft(float *out, const float *in, int n, int sign, int howmany)
{
    if (!is_pow2(n) || howmany < 1 || (sign != 1 && sign != -1)) return -1;
    orc_twiddle t;
    if (twiddle_init(&t, n)) return -1;
    float *x = (float *)malloc(sizeof(float) * 4 * (size_t)n);
    if (!x) { twiddle_free(&t); return -1; }
    for (int h = 0; h < howmany; h++) {
        memcpy(x, in + 2 * (size_t)n * h, sizeof(float) * 2 * (size_t)n);
        fft_one(&t, out + 2 * (size_t)n * h, x, x + 2 * (size_t)n, sign);
    }
    free(x);
    twiddle_free(&t);
    return 0;
}

/* ======================================================================================
 * engine: one ccoherent::threadf iteration per orc_engine_block call
 * ==================================================================================== */

struct orc_engine {
    int nrows, B, L, mode, nfft_cap;
    orc_twiddle tw;
    /* per-row persistent state */
    float *phasecorr;     /* [nrows][2]  csdrdevice::phasecorr      src/csdrdevice.cc:39 */
    float *phasecorrprev; /* [nrows][2]  csdrdevice::phasecorrprev  src/csdrdevice.cc:40 */
    int32_t *lag;         /* lagpoint::lag  include/csdrdevice.h:42-54 */
    float *mag, *frac;
    /* per-block shared */
    float *ref_f;   /* [B] crefsdr sfloat: zeros in [0,L), samples in [L,2L) src/crtlsdr.cc:215-218 */
    float *ref_fft; /* [B] sfft row 0 */
    /* fractional-delay correction (orc_engine_set_frac_apply), off by default */
    int frac_apply;
    float frac_gain;
    float *frac_override; /* [nrows] or NULL */
};

typedef struct {
    float *s;     /* [B] csdrdevice::sfloat, samples in [0,L), zeros in [L,2L) src/crtlsdr.cc:205-207 */
    float *a, *b, *c; /* [B] complex scratch each (sfft/sconv/sifft slots) */
    float *m;     /* [B] smagsqr slot */
} orc_scratch;

static int scratch_init(orc_scratch *w, int B)
{
    size_t cb = sizeof(float) * 2 * (size_t)B;
    w->s = (float *)malloc(cb); w->a = (float *)malloc(cb);
    w->b = (float *)malloc(cb); w->c = (float *)malloc(cb);
    w->m = (float *)malloc(sizeof(float) * (size_t)B);
    return (w->s && w->a && w->b && w->c && w->m) ? 0 : -1;
}
static void scratch_free(orc_scratch *w)
{
    free(w->s); free(w->a); free(w->b); free(w->c); free(w->m);
}

size_t orc_packet_matrix_offset(int nrows) { return 16 + 4 * (size_t)nrows; }
size_t orc_packet_bytes(int nrows, int B) { return orc_packet_matrix_offset(nrows) + (size_t)nrows * (size_t)B; }

void orc_engine_reset(orc_engine *e)
{
    for (int r = 0; r < e->nrows; r++) {
        /* src/csdrdevice.cc:39-40 */
        e->phasecorr[2 * r] = 1.0f; e->phasecorr[2 * r + 1] = 0.0f;
        e->phasecorrprev[2 * r] = 1.0f; e->phasecorrprev[2 * r + 1] = 0.0f;
        e->lag[r] = 0; e->mag[r] = 0.0f; e->frac[r] = 0.0f; /* src/csdrdevice.cc:36-37 */
    }
}

orc_engine *orc_engine_create(int nrows, int B, int mode, int nfft_cap)
{
    if (nrows < 1 || B < 16 || !is_pow2(B)) return NULL;
    if (mode != ORC_MODE_FAITHFUL && mode != ORC_MODE_DIGITAL) return NULL;
    orc_engine *e = (orc_engine *)calloc(1, sizeof(*e));
    if (!e) return NULL;
    e->nrows = nrows; e->B = B; e->L = B >> 1; e->mode = mode; e->nfft_cap = nfft_cap;
    if (twiddle_init(&e->tw, B)) { free(e); return NULL; }
    e->phasecorr = (float *)malloc(sizeof(float) * 2 * (size_t)nrows);
    e->phasecorrprev = (float *)malloc(sizeof(float) * 2 * (size_t)nrows);
    e->lag = (int32_t *)malloc(sizeof(int32_t) * (size_t)nrows);
    e->mag = (float *)malloc(sizeof(float) * (size_t)nrows);
    e->frac = (float *)malloc(sizeof(float) * (size_t)nrows);
    e->ref_f = (float *)malloc(sizeof(float) * 2 * (size_t)B);
    e->ref_fft = (float *)malloc(sizeof(float) * 2 * (size_t)B);
    if (!e->phasecorr || !e->phasecorrprev || !e->lag || !e->mag || !e->frac || !e->ref_f || !e->ref_fft) {
        orc_engine_destroy(e);
        return NULL;
    }
    orc_engine_reset(e);
    return e;
}

void orc_engine_destroy(orc_engine *e)
{
    if (!e) return;
    twiddle_free(&e->tw);
    free(e->phasecorr); free(e->phasecorrprev); free(e->lag); free(e->mag); free(e->frac);
    free(e->ref_f); free(e->ref_fft); free(e->frac_override);
    free(e);
}

int orc_engine_set_frac_apply(orc_engine *e, int enable, float gain, const float *frac_override)
{
    if (!e) return -1;
    e->frac_apply = enable ? 1 : 0;
    e->frac_gain = gain;
    free(e->frac_override);
    e->frac_override = NULL;
    if (enable && frac_override) {
        e->frac_override = (float *)malloc(sizeof(float) * (size_t)e->nrows);
        if (!e->frac_override) return -1;
        memcpy(e->frac_override, frac_override, sizeof(float) * (size_t)e->nrows);
    }
    return 0;
}

/* Fractional-delay correction of one row (SURVEY 8 note "Fractional delay": applied as a linear phase ramp in the
 * frequency domain): the zero-padded row s (B complex) is advanced by delta = lag + D samples,
 *     y = IFFT( FFT(s)[f] * exp(+2 pi i f_s delta / B) ) / B,   f_s = f for f < B/2, f - B otherwise,
 * rotated by the row's phasor, and its first L samples are quantised (cdsp::convto8bit).  For D = 0 and |lag| <= L this
 * is the zero-filled integer shift of the digital mode.  The ramp is formed in double and rounded once. */
static void frac_apply_row(const orc_engine *e, orc_scratch *w, int row, int8_t *out)
{
    const int B = e->B, L = e->L;
    const double D = e->frac_override ? (double)e->frac_override[row] : (double)(e->frac_gain * e->frac[row]);
    const double delta = (double)e->lag[row] + D;
    const float pr = e->phasecorr[2 * row], pi = e->phasecorr[2 * row + 1];
    memcpy(w->c, w->s, sizeof(float) * 2 * (size_t)B);
    fft_one(&e->tw, w->a, w->c, w->b, -1);
    const float invB = 1.0f / (float)B;
    for (int f = 0; f < B; f++) {
        const int fs = f < B / 2 ? f : f - B;
        /* reduce f_s * lag mod B in integers, keep the small f_s * D part in double */
        const long long il = ((long long)fs * (long long)e->lag[row]) % (long long)B;
        const double rev = (double)il / (double)B + (double)fs * D / (double)B;
        const double ang = 2.0 * 3.14159265358979323846 * rev;
        const float hr0 = (float)cos(ang), hi0 = (float)sin(ang);
        /* h = (p / B) * e^{i ang} */
        const float pbr = pr * invB, pbi = pi * invB;
        const float hr = pbr * hr0 - pbi * hi0, hi = pbr * hi0 + pbi * hr0;
        const float xr = w->a[2 * f], xi = w->a[2 * f + 1];
        w->b[2 * f] = xr * hr - xi * hi;
        w->b[2 * f + 1] = xr * hi + xi * hr;
    }
    (void)delta;
    fft_one(&e->tw, w->c, w->b, w->a, +1);
    orc_convto8bit(out, w->c, L);
}

/* ccoherent::computelag for one queued row (src/ccoherent.cc:154-239), row-at-a-time:
 * forward FFT :174, conjugatemul with the ref spectrum :177-179, backward FFT :182,
 * magsquared :185, indexofmax :192, mag :204, lag :232.  The fractional estimate the
 * reference computes and discards (:206-219) is replaced by the standard parabola and
 * reported separately (SURVEY 8 note, "Fractional delay"). */
static void xcorr_row(const orc_engine *e, orc_scratch *w, const float *s_unrotated,
                      int32_t *lag, float *mag, float *frac)
{
    const int B = e->B, L = e->L;
    memcpy(w->c, s_unrotated, sizeof(float) * 2 * (size_t)B);  /* queuelag memcpy :137 */
    fft_one(&e->tw, w->a, w->c, w->b, -1);                      /* sfft  */
    orc_conjugatemul(w->b, w->a, e->ref_fft, B);                /* sconv */
    fft_one(&e->tw, w->c, w->b, w->a, +1);                      /* sifft */
    orc_magsquared(w->m, w->c, B);                              /* smagsqr */
    uint32_t idx = orc_indexofmax(w->m, B);
    float peak = w->m[idx];
    *mag = sqrtf(peak / (float)L);                              /* :204 */
    *lag = (int32_t)idx - (int32_t)L;                           /* :218,221,232 */
    float D = 0.0f;
    if (idx > 0 && idx < (uint32_t)(B - 1)) {
        float ym = w->m[idx - 1], yp = w->m[idx + 1];
        float den = (ym - 2.0f * peak) + yp;
        if (den != 0.0f) D = (0.5f * (ym - yp)) / den;
    }
    *frac = D;
}

/* csdrdevice::est_phasecorrect (src/csdrdevice.cc:58-69) on y (L complex) against the ref
 * samples r = ref sfloat + L (src/ccoherent.cc:272).
 * Defined zero-correlation policy (deviation, SURVEY 7 "hard parts"): the reference divides
 * by abs(correlation)==0 and poisons phasecorr with NaN for ever; here the estimate is
 * skipped for that block (phasecorr and phasecorrprev keep their values). */
static void est_phasecorrect(orc_engine *e, int row, const float *y, const float *r)
{
    const float alpha = 0.5f;
    float corr[2];
    orc_conj_dotproduct(corr, y, r, e->L);              /* :62 */
    float a = hypotf(corr[0], corr[1]);                 /* std::abs(complex<float>) */
    if (a == 0.0f || a != a) return;
    float inv = 1.0f / a;                               /* :63 */
    float pr = corr[0] * inv, pi = (-corr[1]) * inv;    /* conj(correlation) * (1/abs) */
    float *p = e->phasecorr + 2 * row, *pp = e->phasecorrprev + 2 * row;
    float one_m_alpha = 1.0f - alpha;
    p[0] = alpha * pr + one_m_alpha * pp[0];            /* :66 */
    p[1] = alpha * pi + one_m_alpha * pp[1];
    pp[0] = p[0]; pp[1] = p[1];                         /* :67 */
}

static int row_requested(const orc_engine *e, const uint8_t *lag_mask, int row, int *budget)
{
    if (lag_mask && !lag_mask[row]) return 0;
    if (e->nfft_cap > 0) {
        /* src/ccoherent.cc:124: the queue holds the ref plus at most nfft-1 signal rows */
        if (*budget <= 0) return 0;
        (*budget)--;
    }
    return 1;
}

static void process_row(orc_engine *e, orc_scratch *w, int row, const int8_t *rows,
                        int requested, int refnoise_enabled, int8_t *matrix)
{
    const int B = e->B, L = e->L;
    const int8_t *in = rows + (size_t)row * B;
    /* crtlsdr::convtofloat src/crtlsdr.cc:205-207: B int8 -> sfloat[0..L); [L..2L) stays 0 */
    orc_convtofloat(w->s, in, B);
    memset(w->s + 2 * (size_t)L, 0, sizeof(float) * 2 * (size_t)L);
    /* queuelag precedes phasecorrect (src/ccoherent.cc:266-267 vs :275): un-rotated samples */
    if (requested) xcorr_row(e, w, w->s, &e->lag[row], &e->mag[row], &e->frac[row]);

    float *y = w->s;
    if (e->mode == ORC_MODE_DIGITAL) {
        /* SURVEY 8 note (2): y[n] = s[n + lag], zero outside [0,L) */
        int d = e->lag[row];
        float *t = w->a;
        for (int n = 0; n < L; n++) {
            int m = n + d;
            if (m >= 0 && m < L) { t[2 * n] = w->s[2 * m]; t[2 * n + 1] = w->s[2 * m + 1]; }
            else { t[2 * n] = 0.0f; t[2 * n + 1] = 0.0f; }
        }
        y = t;
    }
    if (refnoise_enabled) est_phasecorrect(e, row, y, e->ref_f + 2 * (size_t)L); /* :271-273 */
    if (e->frac_apply && e->mode == ORC_MODE_DIGITAL) {
        /* the matrix row comes from the frequency-domain resampling (shift + fractional delay + rotation in one) */
        if (matrix) frac_apply_row(e, w, row, matrix + (size_t)row * B);
        return;
    }
    /* csdrdevice::phasecorrect src/csdrdevice.cc:80-84, applied every block (:275) */
    orc_scalarmul(y, y, e->phasecorr[2 * row], e->phasecorr[2 * row + 1], L);
    /* cpacketize::write(complex<float>*) src/cpacketizer.cc:158-172 */
    if (matrix) orc_convto8bit(matrix + (size_t)row * B, y, L);
}

static void prepare_ref(orc_engine *e, orc_scratch *w, const int8_t *rows, int8_t *matrix)
{
    const int B = e->B, L = e->L;
    /* crefsdr::convtofloat src/crtlsdr.cc:215-218: samples land in sfloat[L..2L) */
    memset(e->ref_f, 0, sizeof(float) * 2 * (size_t)L);
    orc_convtofloat(e->ref_f + 2 * (size_t)L, rows, B);
    memcpy(w->c, e->ref_f, sizeof(float) * 2 * (size_t)B);   /* queuelag(refdev) :252 */
    fft_one(&e->tw, e->ref_fft, w->c, w->b, -1);              /* sfft slot 0 */
    /* cpacketize::write(int8*) src/cpacketizer.cc:137-156: raw ref block, verbatim */
    if (matrix) memcpy(matrix, rows, (size_t)B);
}

static void write_header(const orc_engine *e, int8_t *packet, const uint32_t *readcnt, uint32_t seq)
{
    /* hdr0 include/cpacketizer.h:32-37 filled in cpacketize::send src/cpacketizer.cc:112-116 */
    uint32_t h[4] = { seq, (uint32_t)e->nrows, (uint32_t)e->L, 0u };
    memcpy(packet, h, 16);
    for (int r = 0; r < e->nrows; r++) {
        uint32_t v = readcnt ? readcnt[r] : seq;              /* src/cpacketizer.cc:142,163 */
        memcpy(packet + 16 + 4 * (size_t)r, &v, 4);
    }
}

static void copy_outputs(const orc_engine *e, int32_t *lag, float *mag, float *frac, float *phasor)
{
    for (int r = 0; r < e->nrows; r++) {
        if (lag) lag[r] = r ? e->lag[r] : 0;
        if (mag) mag[r] = r ? e->mag[r] : 0.0f;
        if (frac) frac[r] = r ? e->frac[r] : 0.0f;
        if (phasor) {
            /* pcorrection[0] is never written: stays 0 (src/cpacketizer.cc:72,131-134) */
            phasor[2 * r] = r ? e->phasecorr[2 * r] : 0.0f;
            phasor[2 * r + 1] = r ? e->phasecorr[2 * r + 1] : 0.0f;
        }
    }
}

int orc_engine_block(orc_engine *e, const int8_t *rows, const uint32_t *readcnt,
                     const uint8_t *lag_mask, int refnoise_enabled, uint32_t seq,
                     int32_t *lag, float *mag, float *frac, float *phasor, int8_t *packet)
{
    return orc_engine_block_mt(e, rows, readcnt, lag_mask, refnoise_enabled, seq,
                               lag, mag, frac, phasor, packet, 1);
}

typedef struct {
    orc_engine *e;
    const int8_t *rows;
    const uint8_t *req; /* resolved per-row request flags */
    int refnoise_enabled;
    int8_t *matrix;
    int row_begin, row_end;
    int rc;
} orc_job;

static void *job_main(void *arg)
{
    orc_job *j = (orc_job *)arg;
    orc_scratch w;
    if (scratch_init(&w, j->e->B)) { scratch_free(&w); j->rc = -1; return NULL; }
    for (int r = j->row_begin; r < j->row_end; r++)
        process_row(j->e, &w, r, j->rows, j->req[r], j->refnoise_enabled, j->matrix);
    scratch_free(&w);
    j->rc = 0;
    return NULL;
}

int orc_engine_block_mt(orc_engine *e, const int8_t *rows, const uint32_t *readcnt,
                        const uint8_t *lag_mask, int refnoise_enabled, uint32_t seq,
                        int32_t *lag, float *mag, float *frac, float *phasor, int8_t *packet,
                        int nthreads)
{
    if (!e || !rows) return -1;
    int8_t *matrix = packet ? packet + orc_packet_matrix_offset(e->nrows) : NULL;
    if (packet) write_header(e, packet, readcnt, seq);

    orc_scratch w0;
    if (scratch_init(&w0, e->B)) { scratch_free(&w0); return -1; }
    prepare_ref(e, &w0, rows, matrix);
    scratch_free(&w0);

    uint8_t *req = (uint8_t *)malloc((size_t)e->nrows);
    if (!req) return -1;
    int budget = e->nfft_cap - 1;
    req[0] = 0;
    for (int r = 1; r < e->nrows; r++) req[r] = (uint8_t)row_requested(e, lag_mask, r, &budget);

    int nsig = e->nrows - 1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > nsig) nthreads = nsig > 0 ? nsig : 1;
    int rc = 0;
    if (nthreads == 1) {
        orc_job j = { e, rows, req, refnoise_enabled, matrix, 1, e->nrows, 0 };
        job_main(&j);
        rc = j.rc;
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
        orc_job *jobs = (orc_job *)malloc(sizeof(orc_job) * (size_t)nthreads);
        if (!th || !jobs) { free(th); free(jobs); free(req); return -1; }
        for (int t = 0; t < nthreads; t++) {
            int b = 1 + (int)((long long)nsig * t / nthreads);
            int en = 1 + (int)((long long)nsig * (t + 1) / nthreads);
            orc_job j = { e, rows, req, refnoise_enabled, matrix, b, en, 0 };
            jobs[t] = j;
            pthread_create(&th[t], NULL, job_main, &jobs[t]);
        }
        for (int t = 0; t < nthreads; t++) { pthread_join(th[t], NULL); if (jobs[t].rc) rc = -1; }
        free(th); free(jobs);
    }
    free(req);
    copy_outputs(e, lag, mag, frac, phasor);
    return rc;
}
