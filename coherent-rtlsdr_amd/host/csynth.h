/* csynth.h -- seeded synthetic int8 IQ block source, C implementation of ../synth.py.
 * Stands in for the RTL-SDR dongles behind csdrdevice (librtlsdr ring buffers,
 * src/crtlsdr.cc:54,61-68,173-203 of the reference).  Integer / correctly-rounded double
 * arithmetic only, so the bytes equal the numpy generator's (tests/test_gpu_host_cpp.py). */
#ifndef CSYNTH_H
#define CSYNTH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct csynth_params {
    int nsig, L, dmax;
    uint64_t seed;
    int64_t *d;         /* [nsig] integer delays: s_k[n] = r[n - d_k] */
    double *c, *s, *g;  /* [nsig] rotation cos/sin, gain */
} csynth_params;

csynth_params *csynth_params_create(int nsig, int L, uint64_t seed, int dmax /* <0: L/4 */, int locked);
void csynth_params_destroy(csynth_params *p);
/* rows: [1 + nsig][2L] int8, row 0 = reference noise.  noise_sigma < 0: default 10 LSB. */
void csynth_make_block(const csynth_params *p, int block, double noise_sigma, int8_t *rows);
/* one row of that block only (row 0 = reference noise, row k >= 1 = signal k-1): [2L] int8 */
void csynth_make_row(const csynth_params *p, int block, int row, double noise_sigma, int8_t *out);
uint64_t csynth_config_seed(int cfg); /* 0xC0FFEE + cfg */

#ifdef __cplusplus
}
#endif
#endif
