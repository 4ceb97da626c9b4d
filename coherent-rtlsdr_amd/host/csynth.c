/* csynth.c -- see csynth.h.  Build with -ffp-contract=off (one rounding per double op, like numpy). */
#include "csynth.h"
#include <math.h>
#include <stdlib.h>

#define GOLD 0x9E3779B97F4A7C15ull
#define M1 0xBF58476D1CE4E5B9ull
#define M2 0x94D049BB133111EBull
static const double INV_SIGMA16 = 1.0 / 37837.22723004292;
static const double SIGMA_REF = 30.0, SIGMA_NOISE = 10.0;

static uint64_t splitmix64(uint64_t seed, uint64_t idx)
{
    uint64_t z = seed + (idx + 1) * GOLD;
    z = (z ^ (z >> 30)) * M1;
    z = (z ^ (z >> 27)) * M2;
    return z ^ (z >> 31);
}
static uint64_t mix(uint64_t seed, uint64_t a, uint64_t b) { return splitmix64(splitmix64(seed, a), b); }
static double gauss16(uint64_t seed, uint64_t counter)
{
    uint64_t z = splitmix64(seed, counter);
    int64_t acc = (int64_t)((z & 0xFFFF) + ((z >> 16) & 0xFFFF) + ((z >> 32) & 0xFFFF) + (z >> 48));
    return (double)(acc - 131070) * INV_SIGMA16;
}
static int8_t quant(double x)
{
    double r = rint(x); /* round-half-even in the default rounding mode */
    if (r > 127.0) r = 127.0;
    if (r < -128.0) r = -128.0;
    return (int8_t)r;
}

uint64_t csynth_config_seed(int cfg) { return 0xC0FFEEull + (uint64_t)cfg; }

csynth_params *csynth_params_create(int nsig, int L, uint64_t seed, int dmax, int locked)
{
    csynth_params *p = (csynth_params *)calloc(1, sizeof(*p));
    if (!p) return NULL;
    p->nsig = nsig; p->L = L; p->seed = seed; p->dmax = dmax < 0 ? L / 4 : dmax;
    p->d = (int64_t *)malloc(sizeof(int64_t) * (size_t)nsig);
    p->c = (double *)malloc(sizeof(double) * (size_t)nsig);
    p->s = (double *)malloc(sizeof(double) * (size_t)nsig);
    p->g = (double *)malloc(sizeof(double) * (size_t)nsig);
    if (!p->d || !p->c || !p->s || !p->g) { csynth_params_destroy(p); return NULL; }
    const uint64_t ps = mix(seed, 0xD1, 0);
    const uint64_t span = (uint64_t)(2 * p->dmax + 1);
    for (int k = 0; k < nsig; ++k) {
        uint64_t zd = splitmix64(ps, 3 * (uint64_t)k), zp = splitmix64(ps, 3 * (uint64_t)k + 1), zg = splitmix64(ps, 3 * (uint64_t)k + 2);
        p->d[k] = (int64_t)(zd % span) - p->dmax;
        double a = (double)((int64_t)(zp & 0xFFFF) - 32768), b = (double)((int64_t)((zp >> 16) & 0xFFFF) - 32768);
        if (a == 0.0 && b == 0.0) a = 1.0;
        double hyp = sqrt(a * a + b * b);
        p->c[k] = a / hyp;
        p->s[k] = b / hyp;
        p->g[k] = 0.5 + 0.5 * ((double)(zg & 0xFFFF) / 65536.0);
    }
    if (nsig >= 4) { /* forced cases 0, +1, -1 */
        const int forced[3] = {0, 1, -1};
        for (int i = 0; i < 3; ++i) if (abs(forced[i]) <= p->dmax) p->d[i] = forced[i];
    }
    if (locked) for (int k = 0; k < nsig; ++k) p->d[k] = 0;
    return p;
}

void csynth_params_destroy(csynth_params *p)
{
    if (!p) return;
    free(p->d); free(p->c); free(p->s); free(p->g); free(p);
}

void csynth_make_row(const csynth_params *p, int block, int row, double noise_sigma, int8_t *out)
{
    const int L = p->L, dm = p->dmax, next = L + 2 * dm;
    const double sn = noise_sigma < 0 ? SIGMA_NOISE : noise_sigma;
    double *re = (double *)malloc(sizeof(double) * (size_t)next), *im = (double *)malloc(sizeof(double) * (size_t)next);
    const uint64_t rs = mix(p->seed, 0xA0, (uint64_t)block);
    for (int m = 0; m < next; ++m) { re[m] = gauss16(rs, 2 * (uint64_t)m); im[m] = gauss16(rs, 2 * (uint64_t)m + 1); }
    if (row == 0) {
        for (int n = 0; n < L; ++n) { out[2 * n] = quant(SIGMA_REF * re[dm + n]); out[2 * n + 1] = quant(SIGMA_REF * im[dm + n]); }
    } else {
        const int k = row - 1;
        const uint64_t ws = mix(p->seed, 0xB000 + (uint64_t)k, (uint64_t)block);
        const double gs = p->g[k] * SIGMA_REF, c = p->c[k], s = p->s[k];
        for (int n = 0; n < L; ++n) {
            const int m = n - (int)p->d[k] + dm;
            const double rot_re = re[m] * c - im[m] * s, rot_im = re[m] * s + im[m] * c;
            const double wr = gauss16(ws, 2 * (uint64_t)n), wi = gauss16(ws, 2 * (uint64_t)n + 1);
            out[2 * n] = quant(gs * rot_re + sn * wr);
            out[2 * n + 1] = quant(gs * rot_im + sn * wi);
        }
    }
    free(re); free(im);
}

void csynth_make_block(const csynth_params *p, int block, double noise_sigma, int8_t *rows)
{
    const int L = p->L, dm = p->dmax, next = L + 2 * dm;
    const double sn = noise_sigma < 0 ? SIGMA_NOISE : noise_sigma;
    double *re = (double *)malloc(sizeof(double) * (size_t)next), *im = (double *)malloc(sizeof(double) * (size_t)next);
    const uint64_t rs = mix(p->seed, 0xA0, (uint64_t)block);
    for (int m = 0; m < next; ++m) { re[m] = gauss16(rs, 2 * (uint64_t)m); im[m] = gauss16(rs, 2 * (uint64_t)m + 1); }
    for (int n = 0; n < L; ++n) {
        rows[2 * n] = quant(SIGMA_REF * re[dm + n]);
        rows[2 * n + 1] = quant(SIGMA_REF * im[dm + n]);
    }
    for (int k = 0; k < p->nsig; ++k) {
        int8_t *out = rows + (size_t)(1 + k) * 2 * (size_t)L;
        const uint64_t ws = mix(p->seed, 0xB000 + (uint64_t)k, (uint64_t)block);
        const double gs = p->g[k] * SIGMA_REF, c = p->c[k], s = p->s[k];
        for (int n = 0; n < L; ++n) {
            const int m = n - (int)p->d[k] + dm;
            const double rot_re = re[m] * c - im[m] * s, rot_im = re[m] * s + im[m] * c;
            const double wr = gauss16(ws, 2 * (uint64_t)n), wi = gauss16(ws, 2 * (uint64_t)n + 1);
            out[2 * n] = quant(gs * rot_re + sn * wr);
            out[2 * n + 1] = quant(gs * rot_im + sn * wi);
        }
    }
    free(re); free(im);
}
