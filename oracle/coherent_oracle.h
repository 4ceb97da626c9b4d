/*
 * coherent_oracle.h -- CPU restatement (plain C, fp32) of the ccoherent/cdsp hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it, and
 * only as the checker / reported CPU baseline -- never as a fallback for the HIP path.
 *
 * PARITY UNPINNED: the reference (mlaaks/coherent-rtlsdr) ships no tests, fixtures or golden
 * vectors for this path, and its arithmetic lives in third-party libraries that are absent
 * here and not version-pinned by the reference (VOLK "apt-get install volk" README.md:32-33,
 * FFTW3f README.md:29-30), so the reference itself cannot be compiled in this image
 * (include/cdsp.h:21 needs volk/volk.h, include/ccoherent.h:21 needs fftw3.h).  This file
 * restates the reference's call sites line by line and VOLK's published *generic* kernels /
 * FFTW's documented transform definition; it is cross-checked against an independent fp64
 * numpy model (oracle/model_fp64.py) and scipy's pocketfft, not against reference output.
 *
 * Units follow the reference (SURVEY.md conventions): B = blocksize = int8 values per row per
 * block = FFT length in complex points; L = B/2 complex samples per row; row 0 = reference
 * noise channel.  Complex data is interleaved (re,im) float, layout-identical to
 * std::complex<float> / fftwf_complex / lv_32fc_t (src/ccoherent.cc:65).
 */
#ifndef COHERENT_ORACLE_H
#define COHERENT_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-op restatements of class cdsp (include/cdsp.h:36-71, src/cdsp.cc) ------------- */

/* cdsp::convtosigned src/cdsp.cc:21-34: out = in ^ 0x80 per byte (n bytes, n % 8 == 0). */
void orc_convtosigned(const uint8_t *in, uint8_t *out, int n);
/* cdsp::convtofloat src/cdsp.cc:36-44 (volk_8i_s32f_convert_32f, scalar 127): n = #int8. */
void orc_convtofloat(float *out, const int8_t *in, int n);
/* cdsp::scalarmul src/cdsp.cc:46-49 (volk_32fc_s32fc_multiply_32fc): n = #complex. */
void orc_scalarmul(float *out, const float *in, float s_re, float s_im, int n);
/* cdsp::convto8bit src/cdsp.cc:51-54 (volk_32f_s32f_convert_8i, scalar 127): n = #complex. */
void orc_convto8bit(int8_t *out, const float *in, int n);
/* cdsp::conj_dotproduct src/cdsp.cc:61-66: sum a[i]*conj(b[i]); res[0]=re, res[1]=im. */
void orc_conj_dotproduct(float *res, const float *a, const float *b, int n);
/* cdsp::magsquared src/cdsp.cc:100-103: out[i] = re^2 + im^2. */
void orc_magsquared(float *out, const float *in, int n);
/* cdsp::conjugatemul src/cdsp.cc:105-108: out = in1 * conj(in2). */
void orc_conjugatemul(float *out, const float *in1, const float *in2, int n);
/* cdsp::indexofmax src/cdsp.cc:135-139 (volk_32f_index_max_32u generic: first strict max). */
uint32_t orc_indexofmax(const float *in, int n);
/* cdsp::fft src/cdsp.cc:110-120 with the plans of src/ccoherent.cc:78-93:
 * `howmany` contiguous length-n complex DFTs, sign = -1 forward / +1 backward, unnormalised,
 * out of place.  n must be a power of two >= 2.  Returns 0, or -1 on bad arguments. */
int orc_fft(float *out, const float *in, int n, int sign, int howmany);

/* ---- engine: ccoherent::threadf per-block loop (src/ccoherent.cc:245-294) --------------- */

enum {
    ORC_MODE_FAITHFUL = 0, /* SURVEY 8 note (1): lag reported, no shift, phase at zero offset */
    ORC_MODE_DIGITAL  = 1  /* SURVEY 8 note (2): shift row by its lag, then phase/rotate     */
};

typedef struct orc_engine orc_engine;

/* nrows = 1 reference row + nsig signal rows; B = blocksize (power of two >= 16).
 * nfft_cap <= 0: cross-correlate every requested row (the build's behaviour);
 * nfft_cap  > 0: emulate the reference's queue cap `lagqueue.size() < nfft`
 *                (src/ccoherent.cc:124), i.e. at most nfft_cap-1 signal rows per block. */
orc_engine *orc_engine_create(int nrows, int B, int mode, int nfft_cap);
void orc_engine_destroy(orc_engine *e);
/* restore the construction-time state of src/csdrdevice.cc:36-40 (phasecorr=prev=1+0j, lag 0) */
void orc_engine_reset(orc_engine *e);

/* One block.
 *   rows      [nrows][B] int8, row 0 = reference noise channel.
 *   readcnt   [nrows] per-row block counters copied into the packet, or NULL (every row: seq).
 *   lag_mask  [nrows] non-zero = csdrdevice::is_lagrequested() for that row
 *             (src/ccoherent.cc:266), entry 0 ignored; NULL = every signal row.
 *   refnoise_enabled  crefnoise::isenabled() gate of src/ccoherent.cc:271.
 * Outputs (any may be NULL):
 *   lag   [nrows] int32  idx - L          (src/ccoherent.cc:232); row 0 = 0
 *   mag   [nrows] float  sqrt(peak / L)   (src/ccoherent.cc:204); row 0 = 0
 *   frac  [nrows] float  standard 3-point parabolic vertex offset in samples, (-0.5,0.5)
 *   phasor[nrows][2] float  csdrdevice::get_phasecorrect() after this block; row 0 = 0
 *                           (the debug side channel, src/cpacketizer.cc:127,131-134)
 *   packet: hdr0{seq,N,L,0} + readcnt[N] + int8 [N][B]  (include/cpacketizer.h:32-37,
 *           src/cpacketizer.cc:137-172), orc_packet_bytes(nrows,B) bytes.
 * Rows whose lag was not requested in this block keep their previous lag/mag/frac. */
int orc_engine_block(orc_engine *e, const int8_t *rows, const uint32_t *readcnt,
                     const uint8_t *lag_mask, int refnoise_enabled, uint32_t seq,
                     int32_t *lag, float *mag, float *frac, float *phasor, int8_t *packet);

/* Fractional-delay correction of the matrix rows in ORC_MODE_DIGITAL (the build's crsdr_plan_set_frac_apply; the reference
 * computes a fractional estimate and discards it, src/ccoherent.cc:206-219; matlabclient/notes.m:9-40 studies applying one):
 * row k is advanced by lag_k + D_k samples through a linear phase ramp on its zero-padded spectrum, rotated by its phasor,
 * quantised.  D_k = frac_override[k] if given ([nrows], copied), else gain * frac_k.  Off by default. */
int orc_engine_set_frac_apply(orc_engine *e, int enable, float gain, const float *frac_override);

size_t orc_packet_bytes(int nrows, int B);   /* 16 + 4*nrows + nrows*B */
size_t orc_packet_matrix_offset(int nrows);  /* 16 + 4*nrows */

/* Multi-threaded driver for the "CPU-allcores" baseline row of BASELINE.md section 3:
 * same result as orc_engine_block, signal rows partitioned over nthreads pthreads. */
int orc_engine_block_mt(orc_engine *e, const int8_t *rows, const uint32_t *readcnt,
                        const uint8_t *lag_mask, int refnoise_enabled, uint32_t seq,
                        int32_t *lag, float *mag, float *frac, float *phasor, int8_t *packet,
                        int nthreads);

/* ---- beamformer_oracle.c: the first consumer of the matrix (beamformclient/heatmap2d2.cpp) ---- */
void orc_covariance(float *rxx, const int8_t *matrix, int nrows, int B);                   /* :185-199 */
int orc_noisesubspace(float *vec, float *sv, const float *rxx, int M);                     /* :69-79   */
void orc_pmusic2d(float *pm, const float *vec, int M, int k, float d, int Mx, int My, int Cx, int Cy); /* :103-147 */

#ifdef __cplusplus
}
#endif
#endif
