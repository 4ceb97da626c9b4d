// longblock.hpp -- cross-correlation for blocks longer than the LDS-resident limit
// (B = 2^15 .. 2^22 complex FFT points per row; BASELINE config 5: L = 2^20, B = 2^21).
//
// A row no longer fits in one CU's LDS (B = 2^21 is 16 MiB of cf32), so the B-point transform is
// split  B = N1 x N2,  N2 = 16384 (the LDS-resident size),  N1 = B / 16384 in {2..256}:
//
//   x[n1 N2 + n2] --A: N1-point column FFTs, x W_B^(n2 k1)--> Y[k1][n2]
//                 --B: 16384-point row FFTs (x conj(ref), inverse), xcorr14.hpp--> Z[k1][n2]
//                 --C: x conj W_B^(n2 k1), inverse N1-point column FFTs, |.|^2, argmax--> partials
//                 --D: reduce partials, parabolic neighbours from N1-term sums--> lag, mag, frac
//
// Every pass streams the row through HBM once with >= 256-byte contiguous segments (A reads the
// int8 input directly, zero half never read; C never writes the correlation back).  This is the
// HBM-streaming regime of the path: ~64 MiB of traffic per 2 MiB int8 row.
// Reference semantics are unchanged: fftwf_plan_many_dft of length B (src/ccoherent.cc:78-93),
// conjugate multiply (:177-179), |.|^2 and first-maximum argmax (:185-192), lag/mag (:204,:232).
#pragma once
#include "arith.hpp"
#include "fft_lds.hpp"
#include "plan_args.hpp"
#include <stdint.h>

namespace crsdr {
namespace lb {

constexpr int LOG2N2 = 14, N2 = 1 << LOG2N2;
// column-pass tile: 8192 elements = 64 KiB of LDS and 512 threads, so two workgroups share a CU and one's
// HBM traffic overlaps the other's column transforms (16384-element tiles, one workgroup per CU: +45 % time)
constexpr int LOG2TILE = 13, TILE = 1 << LOG2TILE, THREADS = 512;
__host__ __device__ constexpr int ntiles(int log2n1) { return 1 << (LOG2N2 + log2n1 - LOG2TILE); } // column tiles per row
constexpr int FBITS = 11;                                                    // fine twiddle table: 2^11 entries
constexpr size_t LDS_BYTES = sizeof(float2) * TILE + 512;

#ifdef CRSDR_LB_EXPERIMENT       // diagnostics only (a -DCRSDR_LB_EXPERIMENT build under tools/): bit 0 no transforms / reduction in stage C,
__device__ int lb_dbg = 0;       // bit 1 XCD-contiguous tile order, bit 2 stage A without transforms
#define LB_DBG(bit) (lb_dbg & (bit))
#else
#define LB_DBG(bit) 0
#endif
struct LongTw {
    const float2 *wc;  // W_B^(i << FBITS), i < B >> FBITS      (coarse)
    const float2 *wf;  // W_B^j,            j < 1 << FBITS      (fine)
    const float2 *tw1; // W_N1^k, k < N1                        (column transforms)
    const float2 *tw2; // W_16384^k, k < 16384                  (row transforms)
    uint32_t bmask;    // B - 1
};

// forward twiddle W_B^m (m taken mod B), two-level table: one product, ~1 ulp
__device__ __forceinline__ float2 tw_big(const LongTw &t, uint32_t m)
{
    m &= t.bmask;
    return cmul(t.wc[m >> FBITS], t.wf[m & ((1u << FBITS) - 1)]);
}

template <int LOG2N1>
__device__ __forceinline__ int rev_n1(int j) { return digit_reverse<LOG2N1>(j); }

// C = TILE / N1 independent N1-point DIF transforms along the rows of an LDS tile laid out
// [n1][C] (same orientation as global memory, lanes run along the C columns: conflict-free).
// Leaves X_col[rev(j)] in tile row j.  Caller syncs before; this syncs after every pass.
// Sink (optional): called as sink(v, R, base, stride, it) for every group of the LAST pass instead of storing it back (output i
// of the group is tile element base + i * stride; it = which of the thread's groups, a constant after unrolling) -- a consumer
// that only reduces or streams the outputs (stage C's |.|^2 / argmax, stage A's twiddle + store) takes them from the registers
// and saves the tile's last write, its re-read and one barrier.
// Source (optional): called as source(v, R, base, stride, it) to FILL the inputs of every group of the FIRST pass instead of
// reading them from the tile (stage A at N1 = 128: int8 straight from global memory) -- saves the tile's first write, a
// barrier and its first read.
struct NoSink {};
struct NoSource {};
template <int LOG2N1, int DIR, int P = 0, typename Sink = NoSink, typename Source = NoSource>
__device__ __forceinline__ void col_fft(float2 *T, const float2 *__restrict__ tw1, int tid, Sink sink = Sink(), Source source = Source())
{
    using G = FftGeom<LOG2N1>;
    constexpr bool SUNK = !__is_same(Sink, NoSink) && P == G::NPASS - 1;
    constexpr bool SOURCED = !__is_same(Source, NoSource) && P == 0;
    if constexpr (P < G::NPASS) {
        constexpr int LOG2C = LOG2TILE - LOG2N1, C = 1 << LOG2C;
        constexpr int LR = G::log2r(P), R = 1 << LR, LM = G::log2m(P), M = 1 << LM;
        constexpr int NG = TILE / R;
#pragma unroll SUNK ? NG / THREADS : 1
        for (int it = 0; it < NG / THREADS; ++it) {
            const int g = tid + it * THREADS;
            const int c = g & (C - 1), gi = g >> LOG2C;
            const int blk = gi >> LM, n2 = gi & (M - 1);
            const int base = (((blk << (LM + LR)) + n2) << LOG2C) + c;
            float2 v[R];
            if constexpr (SOURCED) {
                source(v, R, base, (1 << LM) << LOG2C, it);
            } else {
#pragma unroll
                for (int i = 0; i < R; ++i) v[i] = T[base + ((i << LM) << LOG2C)];
            }
            dft<R, DIR>(v);
            if constexpr (M > 1) {
                if constexpr (LOG2C >= 6) {
                    // a wave sweeps the C >= 64 columns of ONE tile row group: n2 is wave-uniform, so the twiddles are scalar
                    // loads into SGPRs.  That is more than cheaper: a VECTOR load here would be waited for with vmcnt, and
                    // vector loads return in order -- the persistent column kernels' prefetch of the next tile (issued just
                    // before these transforms) would have to land first.
                    // (the table is read through a constant-address-space pointer: only then does the compiler take the uniform
                    // address to the scalar unit -- through a plain global pointer it issued one vector load per twiddle and
                    // waited for each, vmcnt(0), before the product)
                    typedef const float __attribute__((address_space(4))) *ctab_t;
                    const ctab_t tc = (ctab_t)(uintptr_t)tw1;
                    const int n2u = __builtin_amdgcn_readfirstlane(n2);
                    float2 wk[R];
#pragma unroll
                    for (int k = 1; k < R; ++k) {
                        const int o = 2 * ((n2u * k) << (4 * P));
                        wk[k] = make_float2(tc[o], tc[o + 1]);
                    }
#pragma unroll
                    for (int k = 1; k < R; ++k) v[k] = ctw<DIR>(v[k], wk[k]);
                } else {
#pragma unroll
                    for (int k = 1; k < R; ++k) v[k] = ctw<DIR>(v[k], tw1[(n2 * k) << (4 * P)]);
                }
            }
            if constexpr (SUNK) {
                sink(v, R, base, (1 << LM) << LOG2C, it);
            } else {
#pragma unroll
                for (int i = 0; i < R; ++i) T[base + ((i << LM) << LOG2C)] = v[i];
            }
        }
        if constexpr (!SUNK) __syncthreads();
        col_fft<LOG2N1, DIR, P + 1, Sink, Source>(T, tw1, tid, sink, source);
    }
}

// ---- twiddles of the column passes ---------------------------------------------------------------------
// A thread owns the same column pair (n2, n2 + 1) in 8 tile rows j(i) = j0 + S i, so its 16 twiddles are
// W_B^(n2 k1(i)) with k1(i) = k0 + sum_b bit_b(i) ks[b]: four two-level table products and three levels of
// products (<= 4 roundings, ~3e-7) per column instead of 8 x 2 gathered table entries -- a quarter of the
// gathers.  (N1 >= 8; the two shortest block sizes keep the per-element table look-ups: their tiles are wider
// than a thread's stride.)
constexpr uint32_t FMASK = (1u << FBITS) - 1;
struct Ladder {
    float2 g[8];
};
__device__ __forceinline__ void ladder_issue(Ladder &l, const LongTw &t, uint32_t n2, uint32_t k0, uint32_t ks0, uint32_t ks1, uint32_t ks2)
{
    const uint32_t m[4] = {(n2 * k0) & t.bmask, (n2 * ks0) & t.bmask, (n2 * ks1) & t.bmask, (n2 * ks2) & t.bmask};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        l.g[2 * q] = t.wc[m[q] >> FBITS];
        l.g[2 * q + 1] = t.wf[m[q] & FMASK];
    }
}
__device__ __forceinline__ void ladder_expand(const Ladder &l, float2 *w)
{
    const float2 b = cmul(l.g[0], l.g[1]), s0 = cmul(l.g[2], l.g[3]), s1 = cmul(l.g[4], l.g[5]), s2 = cmul(l.g[6], l.g[7]);
    w[0] = b; w[1] = cmul(b, s0); w[2] = cmul(b, s1); w[3] = cmul(w[1], s1);
    w[4] = cmul(b, s2); w[5] = cmul(w[1], s2); w[6] = cmul(w[2], s2); w[7] = cmul(w[3], s2);
}

// the same for 16 twiddles W_B^(n2 (k0 + sum_b bit_b(i) ks[b])), i < 16: five table products, four levels
struct Ladder16 {
    float2 g[10];
};
__device__ __forceinline__ void ladder16_issue(Ladder16 &l, const LongTw &t, uint32_t n2, uint32_t k0, uint32_t ks0, uint32_t ks1, uint32_t ks2, uint32_t ks3)
{
    const uint32_t m[5] = {(n2 * k0) & t.bmask, (n2 * ks0) & t.bmask, (n2 * ks1) & t.bmask, (n2 * ks2) & t.bmask, (n2 * ks3) & t.bmask};
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        l.g[2 * q] = t.wc[m[q] >> FBITS];
        l.g[2 * q + 1] = t.wf[m[q] & FMASK];
    }
}
__device__ __forceinline__ void ladder16_expand(const Ladder16 &l, float2 *w)
{
    w[0] = cmul(l.g[0], l.g[1]);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const float2 sb = cmul(l.g[2 + 2 * b], l.g[3 + 2 * b]);
#pragma unroll
        for (int i = 0; i < (1 << b); ++i) w[i + (1 << b)] = cmul(w[i], sb);
    }
}

// ---- A: int8 -> column FFTs -> x W_B^(n2 k1) -> Y[k1][n2] -------------------------------------------
// Work item w = (row w / ntiles, tile w % ntiles); tile = columns [tile*C, (tile+1)*C), C = TILE / N1.  Signal rows carry
// samples at n < L (n1 < N1/2), the ref row at n >= L (src/crtlsdr.cc:205-207,215-218).
// Persistent form: the grid is two workgroups per CU (what the 64 KiB tiles allow) and a workgroup walks items w, w + grid, ...
// with the NEXT item's loads (int8 words, twiddle-table entries) issued before the column transforms of the current one, and
// the current item's 64 KiB of stores draining under the next item's loads and transforms -- a one-shot workgroup pays its
// load latency, its transforms and its store drain one after the other before the CU slot sees the next tile.
template <int LOG2N1, bool IS_REF>
__global__ __launch_bounds__(THREADS, 4) void k_long_fwd_cols(const int8_t *__restrict__ rows, int row_begin, uint32_t xor80,
                                                           LongTw tw, float2 *__restrict__ Y, int nwork)
{
    constexpr int N1 = 1 << LOG2N1, LOG2C = LOG2TILE - LOG2N1, C = 1 << LOG2C, H = N1 / 2;
    constexpr int LOG2NT = LOG2N2 + LOG2N1 - LOG2TILE, NT = 1 << LOG2NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *T = reinterpret_cast<float2 *>(smem);
    const int tid0 = threadIdx.x;
    const size_t B = (size_t)N1 * N2;
    // non-zero half: H tile rows of C samples; the other H rows are zero
    constexpr int WORDS = H * C / 2 / THREADS;     // = TILE / 4 / THREADS input words per thread, all loaded before the first use
    constexpr bool LADDER = LOG2N1 >= 3;
    constexpr int LOG2S = LADDER ? LOG2N1 - 3 : 0;                   // a thread's tile rows: j0 + (i << LOG2S)
    // STREAM (N1 >= 16: a radix-16 first pass with stride M0 = N1 / 16 rows, then radix RL = N1 / 16 -- or the one pass itself):
    // the first pass reads its 8 non-zero int8 samples (one column, rows n2 + M0 i) straight from global memory, and the
    // outputs go from the last pass's registers through their twiddles straight to Y: a thread then owns ONE column in 16 / RL
    // runs of RL consecutive tile rows, 8-byte stores that a wave still lays down as contiguous runs per k1 -- no tile fill, no
    // zero fill, no last write of the tile and re-read, two barriers less.  In k1 = rev(tile row) a thread's 16 outputs are
    // k0 + (i << KSH) + rev(q * GSTEP) (i < RL its position in a run, q its run): one four-level twiddle ladder covers them.
    using G = FftGeom<LOG2N1>;
    constexpr bool STREAM = LOG2N1 >= 4;
    constexpr int M0 = STREAM ? N1 / 16 : 1;
    constexpr int LRL = G::log2r(G::NPASS - 1), RL = 1 << LRL, KSH = 4 * (G::NPASS - 1);
    constexpr int GSTEP = ((THREADS >> LOG2C) << LRL);                // tile rows between a thread's consecutive runs
    uint32_t u[WORDS];
    uint16_t h[8];
    Ladder la, lb_;
    Ladder16 l16;
    // the loads of one work item: its int8 words and the table entries of its output twiddles (used after the column transforms)
    auto issue_words = [&](int w, int tid) {
        const int tile = w & (NT - 1), row = IS_REF ? 0 : row_begin + (w >> LOG2NT);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(rows + (size_t)row * B); // word = 2 samples
        if constexpr (STREAM) {
            // first pass (radix 16, stride M0 rows): group g = tid: column g % C, rows (g / C) + M0 i; storage rows 0 .. H - 1 hold
            // the non-zero half (tile rows 0 .. H - 1 of a signal row, H .. N1 - 1 of the reference row)
            const uint16_t *s16 = reinterpret_cast<const uint16_t *>(src);
            const int c = tid & (C - 1), n2 = tid >> LOG2C;
#pragma unroll
            for (int i = 0; i < 8; ++i) h[i] = s16[(size_t)(n2 + M0 * i) * N2 + (size_t)tile * C + c];
            return;
        }
#pragma unroll
        for (int i = 0; i < WORDS; ++i) {
            const int x = tid + i * THREADS, n1 = (2 * x) >> LOG2C, c = (2 * x) & (C - 1);
            u[i] = src[((size_t)n1 * N2 + (size_t)tile * C + c) >> 1];
        }
    };
    auto issue_ladder = [&](int w, int tid) {
        const int tile = w & (NT - 1);
        if constexpr (STREAM) {
            // run q of the last pass: group tid + q * THREADS = column tid % C, tile rows ((tid / C) << LRL) + q * GSTEP + i, i < RL;
            // ladder bit b < LRL is bit b of i, bit LRL + b' is bit b' of q
            constexpr auto step = [](int b) -> uint32_t {
                return b < LRL ? (uint32_t)((1 << b) << KSH) : (uint32_t)digit_reverse_c<LOG2N1>((GSTEP << (b - LRL)) & (N1 - 1));
            };
            const uint32_t c0 = (uint32_t)(tid & (C - 1)), jb0 = (uint32_t)(tid >> LOG2C) << LRL;
            ladder16_issue(l16, tw, (uint32_t)(tile * C) + c0, (uint32_t)rev_n1<LOG2N1>((int)jb0), step(0), step(1), step(2), step(3));
        } else if constexpr (LADDER) {
            const uint32_t j0 = (uint32_t)(2 * tid) >> LOG2C, n2 = (uint32_t)(tile * C + ((2 * tid) & (C - 1)));
            const uint32_t k0 = (uint32_t)rev_n1<LOG2N1>((int)j0);
            constexpr uint32_t ks0 = (uint32_t)digit_reverse_c<LOG2N1>(1 << LOG2S), ks1 = (uint32_t)digit_reverse_c<LOG2N1>(2 << LOG2S),
                               ks2 = (uint32_t)digit_reverse_c<LOG2N1>(4 << LOG2S);
            ladder_issue(la, tw, n2, k0, ks0, ks1, ks2);
            ladder_issue(lb_, tw, n2 + 1u, k0, ks0, ks1, ks2);
        }
    };
    int w = blockIdx.x;
    if (w >= nwork) return;
    issue_words(w, tid0);
    issue_ladder(w, tid0);
    for (;;) {
        // opaque per iteration: otherwise every LDS / global offset derived from the thread index is hoisted out of the item
        // loop and kept (spilled) across it -- the one-shot form of this kernel needed 82 registers, the loop 128 + 53 spills
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int tile = w & (NT - 1);
        if constexpr (!STREAM) {
#pragma unroll
            for (int i = 0; i < WORDS; ++i) {
                const int xw = tid + i * THREADS, n1 = (2 * xw) >> LOG2C, c = (2 * xw) & (C - 1);
                const uint32_t x = u[i] ^ xor80;
                const int r = IS_REF ? n1 + H : n1;
                *reinterpret_cast<float4 *>(T + (r << LOG2C) + c) =
                    make_float4(i8_to_f32(sext8(x, 0)), i8_to_f32(sext8(x, 1)), i8_to_f32(sext8(x, 2)), i8_to_f32(sext8(x, 3)));
            }
            for (int e = tid; e < H * C; e += THREADS) T[((IS_REF ? 0 : H) << LOG2C) + e] = make_float2(0.f, 0.f);
        }
        // a thread's 16 outputs as 8 column pairs, 16-byte stores; twiddles from the ladder (N1 >= 8) or, for the two
        // shortest sizes, 32 table entries all issued before the first product
        constexpr int PAIRS = TILE / 2 / THREADS;
        float2 wa[LADDER ? 8 : 2 * PAIRS], wb[LADDER ? 8 : 2 * PAIRS];
        const int wn = w + (int)gridDim.x;
        float2 *Yr = Y + (size_t)(IS_REF ? 0 : (w >> LOG2NT)) * B;
        if constexpr (STREAM) {
            // inputs of the first pass from the registers: 8 samples of one column, the other 8 rows of the group are the zero pad
            uint32_t hx[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) hx[i] = (uint32_t)h[i] ^ (xor80 & 0xFFFFu);
            if (wn < nwork) issue_words(wn, tid);                            // the next item's: in flight during the column transforms
            auto get = [&](float2 *v, int r, int, int, int) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[IS_REF ? i + 8 : i] = make_float2(i8_to_f32(sext8(hx[i], 0)), i8_to_f32(sext8(hx[i], 1)));
                    v[IS_REF ? i : i + 8] = make_float2(0.f, 0.f);
                }
            };
            float2 w16[16];
            auto put = [&](const float2 *v, int r, int base, int stride, int it) {
                if (it == 0) ladder16_expand(l16, w16);        // once per item: the runs are unrolled, `it` is a constant
                const int c = base & (C - 1), jb = base >> LOG2C;
                float2 *dst = Yr + (size_t)rev_n1<LOG2N1>(jb) * N2 + (tile * C + c);      // k1 = rev(jb) + (i << KSH)
#pragma unroll
                for (int i = 0; i < RL; ++i) dst[(size_t)i * ((size_t)N2 << KSH)] = cmul(v[i], w16[(it << LRL) + i]);
            };
            col_fft<LOG2N1, -1, 0, decltype(put), decltype(get)>(T, tw.tw1, tid, put, get);
            if (wn < nwork) issue_ladder(wn, tid);
            if (wn >= nwork) break;
            w = wn;
            __syncthreads();
            continue;
        }
        __syncthreads();
        if (wn < nwork) issue_words(wn, tid);                                // in flight during the column transforms
        if (!LB_DBG(4)) col_fft<LOG2N1, -1>(T, tw.tw1, tid);
        if constexpr (LADDER) {
            ladder_expand(la, wa);
            ladder_expand(lb_, wb);
            if (wn < nwork) issue_ladder(wn, tid);                           // the next item's table entries: used after ITS column transforms
        }
        if constexpr (!LADDER) {
#pragma unroll
            for (int i = 0; i < PAIRS; ++i) {
                const int e = 2 * (tid + i * THREADS), j = e >> LOG2C, c = e & (C - 1);
                const uint32_t k1 = (uint32_t)rev_n1<LOG2N1>(j), n2 = (uint32_t)(tile * C + c);
                const uint32_t m0 = (n2 * k1) & tw.bmask, m1 = ((n2 + 1u) * k1) & tw.bmask;
                wa[2 * i] = tw.wc[m0 >> FBITS]; wb[2 * i] = tw.wf[m0 & FMASK];
                wa[2 * i + 1] = tw.wc[m1 >> FBITS]; wb[2 * i + 1] = tw.wf[m1 & FMASK];
            }
        }
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            const int e = 2 * (tid + i * THREADS), j = e >> LOG2C, c = e & (C - 1);
            const int k1 = rev_n1<LOG2N1>(j), n2 = tile * C + c;
            const float4 t = *reinterpret_cast<const float4 *>(T + e);
            float2 y0, y1;
            if constexpr (LADDER) {
                y0 = cmul(make_float2(t.x, t.y), wa[i]);
                y1 = cmul(make_float2(t.z, t.w), wb[i]);
            } else {
                y0 = cmul(make_float2(t.x, t.y), cmul(wa[2 * i], wb[2 * i]));
                y1 = cmul(make_float2(t.z, t.w), cmul(wa[2 * i + 1], wb[2 * i + 1]));
            }
            *reinterpret_cast<float4 *>(Yr + (size_t)k1 * N2 + n2) = make_float4(y0.x, y0.y, y1.x, y1.y);
        }
        if (wn >= nwork) break;
        w = wn;
        __syncthreads();                                // every thread has read its outputs from the tile: the next item may fill it
    }
}

// ---- B: 16384-point row transforms: x14::k_rows14_cf32 (xcorr14.hpp), the K1 structure on cf32 lines ---------

// ---- C: x conj W_B^(n2 k1) -> inverse column FFTs -> |.|^2 -> per-tile argmax ----------------------------
struct LongPartial {
    float m;
    int idx;
};

// OUTPUT (the apply pass of crsdr_plan_set_frac_apply): Z is the inverse row transform of a row's RESAMPLED spectrum; the
// column transforms then yield its time samples, and instead of |.|^2 / argmax the first L of them (n1 < N1/2: the second
// half is the zero pad the shift wrapped into) are quantised like cdsp::convto8bit (src/cdsp.cc:51-54) into the row at
// out + blockIdx.y * B -- the cpacketize::write(complex<float>*) of src/cpacketizer.cc:158-172 for this mode.
template <int LOG2N1, bool OUTPUT = false>
__global__ __launch_bounds__(THREADS, 4) void k_long_inv_cols(const float2 *__restrict__ Z, LongTw tw, LongPartial *__restrict__ part, int8_t *__restrict__ out,
                                                           int nwork)
{
    // persistent, like k_long_fwd_cols: work item w = (row w / ntiles, tile w % ntiles); the next item's 64 KiB of Z and its
    // twiddle-table entries are in flight during the column transforms and the reduction of the current one
    constexpr int N1 = 1 << LOG2N1, LOG2C = LOG2TILE - LOG2N1, C = 1 << LOG2C;
    constexpr int LOG2NT = LOG2N2 + LOG2N1 - LOG2TILE, NT = 1 << LOG2NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *T = reinterpret_cast<float2 *>(smem);
    LongPartial *wred = reinterpret_cast<LongPartial *>(smem + sizeof(float2) * TILE);
    const int tid0 = threadIdx.x;
    const size_t B = (size_t)N1 * N2;
    // a thread's 16 inputs as 8 column pairs: every load (8 x 16 bytes of Z, the twiddle-table entries) is in
    // flight before the first product -- in a rolled loop each few elements cost a memory round trip
    constexpr int PAIRS = TILE / 2 / THREADS;
    constexpr bool LADDER = LOG2N1 >= 3;
    constexpr int LOG2S = LADDER ? LOG2N1 - 3 : 0;
    // STREAM (N1 >= 16: the first pass is a radix 16 with stride M = N1 / 16 rows): it takes its inputs from global memory -- a
    // thread's group is one column in the tile rows r + M i: 16 loads of 8 bytes that a wave lays down as contiguous runs per
    // k1 -- times their twiddles (a four-level ladder), instead of filling the tile first: one tile write, one barrier and one
    // tile read less
    constexpr bool STREAM = LOG2N1 >= 4;
    constexpr int M0 = STREAM ? N1 / 16 : 1;
    float4 z[PAIRS];
    float2 zz[STREAM ? 16 : 1];
    Ladder la, lb_;
    Ladder16 l16;
    float2 ta[LADDER ? 1 : 2 * PAIRS], tb[LADDER ? 1 : 2 * PAIRS];
    auto tile_of = [&](int w) { const int t = w & (NT - 1); return LB_DBG(2) ? ((t & 7) * (NT / 8) + (t >> 3)) : t; };
    auto issue_z = [&](int w, int tid) {
        const int tile = tile_of(w);
        const float2 *Zr = Z + (size_t)(w >> LOG2NT) * B;
        if constexpr (STREAM) {
            const int c = tid & (C - 1), r = tid >> LOG2C;
#pragma unroll
            for (int i = 0; i < 16; ++i) zz[i] = Zr[(size_t)(r + M0 * i) * N2 + tile * C + c];
            return;
        }
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            const int e = 2 * (tid + i * THREADS);
            int k1 = e >> LOG2C, c = e & (C - 1), n2 = tile * C + c;
            if (LB_DBG(8)) {          // diagnostic (with bit 0): the same bytes as 2C-column x N1/2-row pieces -- twice the segment length
                k1 = (e >> (LOG2C + 1)) + (N1 / 2) * (tile & 1);
                n2 = (tile >> 1) * 2 * C + (e & (2 * C - 1));
            }
            z[i] = *reinterpret_cast<const float4 *>(Zr + (size_t)k1 * N2 + n2);
        }
    };
    auto issue_tw = [&](int w, int tid) {
        const int tile = tile_of(w);
        if constexpr (STREAM) {
            ladder16_issue(l16, tw, (uint32_t)(tile * C + (tid & (C - 1))), (uint32_t)(tid >> LOG2C), (uint32_t)M0, 2u * M0, 4u * M0, 8u * M0);
        } else if constexpr (LADDER) {
            const uint32_t k0 = (uint32_t)(2 * tid) >> LOG2C, n2 = (uint32_t)(tile * C + ((2 * tid) & (C - 1)));
            ladder_issue(la, tw, n2, k0, 1u << LOG2S, 2u << LOG2S, 4u << LOG2S);
            ladder_issue(lb_, tw, n2 + 1u, k0, 1u << LOG2S, 2u << LOG2S, 4u << LOG2S);
        } else {
#pragma unroll
            for (int i = 0; i < PAIRS; ++i) {
                const int e = 2 * (tid + i * THREADS);
                const uint32_t k1 = (uint32_t)(e >> LOG2C), n2 = (uint32_t)(tile * C + (e & (C - 1)));
                const uint32_t m0 = (n2 * k1) & tw.bmask, m1 = ((n2 + 1u) * k1) & tw.bmask;
                ta[2 * i] = tw.wc[m0 >> FBITS]; tb[2 * i] = tw.wf[m0 & FMASK];
                ta[2 * i + 1] = tw.wc[m1 >> FBITS]; tb[2 * i + 1] = tw.wf[m1 & FMASK];
            }
        }
    };
    int w = blockIdx.x;
    if (w >= nwork) return;
    issue_z(w, tid0);
    issue_tw(w, tid0);
    for (;;) {
        int tid = tid0;                                 // opaque per iteration (see k_long_fwd_cols)
        asm volatile("" : "+v"(tid));
        const int tile = tile_of(w), rowi = w >> LOG2NT;
        if constexpr (!STREAM) {
            float2 wa[8], wb[8];
            if constexpr (LADDER) {
                ladder_expand(la, wa);
                ladder_expand(lb_, wb);
            }
#pragma unroll
            for (int i = 0; i < PAIRS; ++i) {
                const int e = 2 * (tid + i * THREADS);
                float2 y0, y1;
                if constexpr (LADDER) {
                    y0 = cmulc(make_float2(z[i].x, z[i].y), wa[i]);
                    y1 = cmulc(make_float2(z[i].z, z[i].w), wb[i]);
                } else {
                    y0 = cmulc(make_float2(z[i].x, z[i].y), cmul(ta[2 * i], tb[2 * i]));
                    y1 = cmulc(make_float2(z[i].z, z[i].w), cmul(ta[2 * i + 1], tb[2 * i + 1]));
                }
                *reinterpret_cast<float4 *>(T + e) = make_float4(y0.x, y0.y, y1.x, y1.y);
            }
            __syncthreads();
        }
        const int wn = w + (int)gridDim.x;
        if constexpr (!STREAM) {
            if (wn < nwork) issue_z(wn, tid);                                // 64 KiB in flight during the column transforms
        }
        // STREAM: the first pass's inputs = the loaded values times their twiddles; the next item's loads go out right behind
        auto get = [&](float2 *v, int, int, int, int) {
            float2 w16[16];
            ladder16_expand(l16, w16);
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = cmulc(zz[STREAM ? i : 0], w16[i]);
            if (wn < nwork) issue_z(wn, tid);
        };
        float bm = -1.0f;
        int bi = 0x7fffffff;
        // (a radix-16 last pass, N1 = 16 or 256, keeps the tile round trip: 16 outputs plus their indices in flight spilled 24-31 registers)
        constexpr bool SMALL_LAST = FftGeom<LOG2N1>::log2r(FftGeom<LOG2N1>::NPASS - 1) <= 3;
        constexpr bool FROM_REGS = !OUTPUT && SMALL_LAST, EMIT = OUTPUT && SMALL_LAST && STREAM;
        if constexpr (EMIT) {
            // the apply pass: the first L samples quantised like cdsp::convto8bit (src/cdsp.cc:51-54) straight from the last pass's
            // registers, one sample (two bytes) per lane: a wave's 64 columns of one time row are 128 contiguous bytes
            int8_t *orow = out + (size_t)rowi * B;
            auto emit = [&](const float2 *v, int r, int base, int stride, int) {
#pragma unroll
                for (int i = 0; i < r; ++i) {
                    const int e = base + i * stride, j = e >> LOG2C, c = e & (C - 1), n1 = rev_n1<LOG2N1>(j);
                    if (n1 < N1 / 2)
                        *reinterpret_cast<uint16_t *>(orow + 2 * ((size_t)n1 * N2 + (size_t)(tile * C + c))) =
                            (uint16_t)((uint32_t)(uint8_t)f32_to_i8(v[i].x) | ((uint32_t)(uint8_t)f32_to_i8(v[i].y) << 8));
                }
            };
            col_fft<LOG2N1, +1, 0, decltype(emit), decltype(get)>(T, tw.tw1, tid, emit, get);
        } else if constexpr (!FROM_REGS) {
            if constexpr (STREAM) col_fft<LOG2N1, +1, 0, NoSink, decltype(get)>(T, tw.tw1, tid, NoSink(), get);
            else col_fft<LOG2N1, +1>(T, tw.tw1, tid);
        } else {
            // |.|^2 and the running (first) maximum straight from the last pass's registers
            auto take = [&](const float2 *v, int r, int base, int stride, int) {
#pragma unroll
                for (int i = 0; i < r; ++i) {
                    const int e = base + i * stride, j = e >> LOG2C, c = e & (C - 1);
                    const int n = rev_n1<LOG2N1>(j) * N2 + tile * C + c; // natural sample index of this output
                    const float m = fmaf(v[i].x, v[i].x, v[i].y * v[i].y);
                    if (m > bm || (m == bm && n < bi)) { bm = m; bi = n; }
                }
            };
            if constexpr (STREAM) col_fft<LOG2N1, +1, 0, decltype(take), decltype(get)>(T, tw.tw1, tid, take, get);
            else if (!LB_DBG(1)) col_fft<LOG2N1, +1, 0, decltype(take)>(T, tw.tw1, tid, take);
        }
        if (wn < nwork) issue_tw(wn, tid);                                   // table entries (cache hits): under the reduction below
        if constexpr (EMIT) {
        } else if constexpr (OUTPUT) {
            int8_t *orow = out + (size_t)rowi * B;
            for (int e = 2 * tid; e < TILE; e += 2 * THREADS) {          // column pairs: one 32-bit store = samples n, n + 1
                const int j = e >> LOG2C, c = e & (C - 1), n1 = rev_n1<LOG2N1>(j);
                if (n1 >= N1 / 2) continue;
                const float4 y = *reinterpret_cast<const float4 *>(T + e);
                const uint32_t wd = (uint32_t)(uint8_t)f32_to_i8(y.x) | ((uint32_t)(uint8_t)f32_to_i8(y.y) << 8) |
                                    ((uint32_t)(uint8_t)f32_to_i8(y.z) << 16) | ((uint32_t)(uint8_t)f32_to_i8(y.w) << 24);
                *reinterpret_cast<uint32_t *>(orow + 2 * ((size_t)n1 * N2 + (size_t)(tile * C + c))) = wd;
            }
        } else {
            if constexpr (!FROM_REGS) {
                for (int e = tid; e < TILE; e += THREADS) {
                    const int j = e >> LOG2C, c = e & (C - 1);
                    const int n = rev_n1<LOG2N1>(j) * N2 + tile * C + c; // natural sample index of this output
                    const float2 y = T[e];
                    const float m = fmaf(y.x, y.x, y.y * y.y);
                    if (m > bm || (m == bm && n < bi)) { bm = m; bi = n; }
                }
            } else if (LB_DBG(1)) {
                for (int e = tid; e < 2 * THREADS; e += THREADS) { const float2 y = T[e]; const float m = fmaf(y.x, y.x, y.y * y.y); if (m > bm) { bm = m; bi = e; } }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float om = __shfl_xor(bm, off, 64);
                const int oi = __shfl_xor(bi, off, 64);
                if (om > bm || (om == bm && oi < bi)) { bm = om; bi = oi; }
            }
            if ((tid & 63) == 0) wred[tid >> 6] = LongPartial{bm, bi};
            __syncthreads();
            if (tid == 0) {
                LongPartial b = wred[0];
                for (int x = 1; x < THREADS / 64; ++x)
                    if (wred[x].m > b.m || (wred[x].m == b.m && wred[x].idx < b.idx)) b = wred[x];
                part[(size_t)rowi * NT + tile] = b;                       // [row][tile]
            }
        }
        if (wn >= nwork) break;
        w = wn;
        __syncthreads();                                // the tile and the reduction scratch are free again
    }
}

// ---- D: reduce the per-tile partials, parabolic neighbours, publish -----------------------------------
// y[n] for a single n is an N1-term sum over the column of Z that holds it:
//   y[n1 N2 + n2] = sum_k1 Z[k1][n2] conj(W_B^(n2 k1)) conj(W_N1^(n1 k1))
__global__ __launch_bounds__(256) void k_long_finalize(const float2 *__restrict__ Z, const LongPartial *__restrict__ part, LongTw tw,
                                                       int N1, int ntile, XcorrArgs a, long long *__restrict__ corr_zero)
{
    __shared__ LongPartial sp[256];
    __shared__ float snb[2][2 * 256];
    const int tid = threadIdx.x, row = a.row_begin + (int)blockIdx.x, t = 0;
    // the long rows' dot product accumulates its chunks into corr[row] with integer atomics: zeroed here, one kernel ahead of it,
    // instead of by a memset of its own between the two (a launch and a gap per block)
    if (corr_zero && tid < 2) corr_zero[2 * (size_t)row + tid] = 0;
    if (xcorr_skip(a, row, t, tid)) return;
    const size_t B = (size_t)N1 * N2;
    LongPartial b = {-1.0f, 0x7fffffff};
    for (int i = tid; i < ntile; i += 256) {
        const LongPartial p = part[(size_t)blockIdx.x * ntile + i];
        if (p.m > b.m || (p.m == b.m && p.idx < b.idx)) b = p;
    }
    sp[tid] = b;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) {
            const LongPartial o = sp[tid + s];
            if (o.m > sp[tid].m || (o.m == sp[tid].m && o.idx < sp[tid].idx)) sp[tid] = o;
        }
        __syncthreads();
    }
    const float gm = sp[0].m;
    int gi = sp[0].idx;
    if ((unsigned)gi >= (unsigned)B) gi = 0;
    const float2 *Zr = Z + (size_t)blockIdx.x * B;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const long n = (long)gi + (side ? 1 : -1);
        float2 acc = make_float2(0.f, 0.f);
        if (n >= 0 && n < (long)B) {
            const uint32_t n1 = (uint32_t)(n >> LOG2N2), n2 = (uint32_t)(n & (N2 - 1));
            for (int k1 = tid; k1 < N1; k1 += 256) {
                float2 z = cmulc(Zr[(size_t)k1 * N2 + n2], tw_big(tw, n2 * (uint32_t)k1));
                z = cmulc(z, tw.tw1[(n1 * (uint32_t)k1) & (uint32_t)(N1 - 1)]);
                acc = cadd(acc, z);
            }
        }
        snb[side][2 * tid] = acc.x;
        snb[side][2 * tid + 1] = acc.y;
    }
    __syncthreads();
    // the two neighbours' 256 partial sums: pairwise tree (it was one thread walking 1024 LDS words: 9.2 -> 6.0 us per block)
    for (int st = 128; st >= 1; st >>= 1) {
        if (tid < st) {
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                snb[side][2 * tid] += snb[side][2 * (tid + st)];
                snb[side][2 * tid + 1] += snb[side][2 * (tid + st) + 1];
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        float mn[2];
        for (int side = 0; side < 2; ++side) {
            const float sx = snb[side][0], sy = snb[side][1];
            mn[side] = fmaf(sx, sx, sy * sy);
        }
        float D = 0.0f;
        if (gi > 0 && gi < (int)B - 1) {
            const float den = (mn[0] - 2.0f * gm) + mn[1];
            if (den != 0.0f) D = (0.5f * (mn[0] - mn[1])) / den;
        }
        const int L = (int)(B >> 1);
        xcorr_publish(a, row, t, gi - L /* src/ccoherent.cc:232 */, sqrtf(gm / (float)L) /* :204 */, D);
    }
}

} // namespace lb
} // namespace crsdr
