// xcorr14q.hpp -- K1 with TWO rows per CU in opposite phases (experiment, CRSDR_K1_VARIANT=q).
//
// tools/lds_valu_overlap.hip: on one SIMD the LDS stores of one wave overlap the packed VALU stream of the other,
// but K1's two waves per SIMD run the same row and hit their store bursts together, so a row costs VALU time PLUS
// store time.  Here one persistent 512-thread workgroup runs two rows at once: waves 0-3 (one per SIMD) are group 0,
// waves 4-7 group 1, every thread does the work of two threads of xcorr14p.hpp one after the other (virtual threads
// tid and tid + 256: same passes, same LDS image, same arithmetic -> identical bits).  A row needs the one 132 KiB
// image only from its P0 stores to its P0' loads; outside that window (P0' arithmetic, epilogue, the next row's loads
// and first butterflies) a group touches no LDS, and that is when the other group owns the image and does its
// store-heavy middle section.  Hand-over: one owner word in LDS (compare-and-swap by every wave, tagged with the
// row's sequence number), released by the last wave of the owner group that has its P0' values in registers.
// Barriers inside a group (4 waves) are LDS counters: s_barrier would join both groups.
// Every wait is bounded; running out sets an error word (rows then hold garbage, the kernel still terminates).
#pragma once
#include "xcorr14p.hpp"

namespace crsdr {
namespace x14p {

constexpr int QG = 256;                                       // threads per row group
// per-group scratch behind the image: the four waves' argmax candidates (max, first index, the peak's in-wave neighbours: 64 B)
// and the magnitudes of every wave's first and last lane (4 waves x 2 lanes x 64 values: a peak on a wave edge has its
// neighbour there); k_rows14_cf32q only uses the sync words
constexpr int QSCR_BYTES = 64 + 4 * 2 * 64 * 4;
constexpr int LDSQ_BYTES = LDS_ELEMS * 8 + 2 * QSCR_BYTES + 64;      // image + scratch per group + sync words (sizeof(QSync) <= 64)
constexpr int kQSpinLimit = 1 << 18;
#ifndef Q_PRIO
#define Q_PRIO 3
#endif
#ifndef Q_ACQ_SLEEP
#define Q_ACQ_SLEEP 2
#endif

struct QSync {
    int owner;      // 0 = image free, else 2 * seq + group + 1
    int relcnt;     // waves of the owner group that hold their P0' values
    int bar[2];     // arrivals at each group's barrier (monotonic)
    int err;        // a bounded wait ran out
    int next[2];    // the item each group takes after its current one (drawn one row ahead by the group's first lane)
    int spin_limit; // polls a wait may take (kQSpinLimit; a test forces 0 to see the error path)
    int arrive[2];  // waves of each group that have published their argmax candidate (monotonic: the fourth of a row finishes it)
};
static_assert(sizeof(QSync) <= 64, "QSync must fit the 64 bytes reserved behind the scratch");

#ifdef CRSDR_QDEBUG      // tools/k1_pair.hip: cycles workgroup 0's waves spend in group barriers [1] / waiting for the image [2], total [0]
__device__ unsigned long long *dbg__ = nullptr;
#define QDBG_T0 const unsigned long long t0__ = __builtin_readcyclecounter();
#define QDBG_ADD(slot) if (dbg__ && blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicAdd(dbg__ + (slot), (unsigned long long)(__builtin_readcyclecounter() - t0__));
#else
#define QDBG_T0
#define QDBG_ADD(slot)
#endif
__device__ __forceinline__ void q_barrier(QSync *s, int g, int &gen, int site = 0)
{
    QDBG_T0
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&s->bar[g], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    gen += QG / 64;
    int spins = 0;
    while (__hip_atomic_load(&s->bar[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < gen) {
        if (++spins > s->spin_limit || __hip_atomic_load(&s->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
            __hip_atomic_store(&s->err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    QDBG_ADD(1)
    QDBG_ADD(3 + site)
}

__device__ __forceinline__ void q_acquire(QSync *s, int tag)
{
    QDBG_T0
    int spins = 0;
    for (;;) {
        int old = tag;
        if ((threadIdx.x & 63) == 0) {
            int expected = 0;
            __hip_atomic_compare_exchange_strong(&s->owner, &expected, tag, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            old = expected;                    // 0 if this wave took it, else the current owner
        }
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == 0 || old == tag) break;
        __builtin_amdgcn_s_sleep(Q_ACQ_SLEEP);
        if (++spins > s->spin_limit || __hip_atomic_load(&s->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
            __hip_atomic_store(&s->err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    QDBG_ADD(2)
}

__device__ __forceinline__ void q_release(QSync *s)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // this wave's image loads have returned
    if ((threadIdx.x & 63) == 0) {
        const int n = __hip_atomic_fetch_add(&s->relcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (n == QG / 64 - 1) {
            __hip_atomic_store(&s->relcnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&s->owner, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// P0 of one virtual thread without the stores (pass0_forward<false> of xcorr14p.hpp)
__device__ __forceinline__ void q_p0_compute(c2 *v, const int8_t *__restrict__ row, const c2 *__restrict__ twA, uint32_t xor80, int vt)
{
    const uint16_t *src = reinterpret_cast<const uint16_t *>(row);
    const uint32_t x16 = xor80 & 0xFFFFu;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t u = (uint32_t)src[i * 512 + vt] ^ x16;
        v[i] = mk((float)sext8(u, 0), (float)sext8(u, 1));
    }
    dft32_stage1_pruned<0, false>(v);
    dft16p<-1>(v);
    dft16p<-1>(v + 16);
    c2 w[32];
    tw_load(w, twA, TWA_STRIDE, vt);
    tw_apply<-1, true, 1>(v, w);
}
__device__ __forceinline__ void q_p0_store(c2 *A, const c2 *v, int vt)
{
    int base = p0_base(vt);
    asm volatile("" : "+v"(base));       // formed here, not hoisted out of the row loop and kept (spilled) as 32 addresses
#pragma unroll
    for (int k = 0; k < 32; ++k) A[base + k * 528] = v[xpos(k)];
}
// P0' of one virtual thread: loads, then (separately) the arithmetic down to |.|^2
__device__ __forceinline__ void q_p0i_load(c2 *v, const c2 *A, int vt)
{
    int base = p0_base(vt);
    asm volatile("" : "+v"(base));
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = A[base + k * 528];
}
__device__ __forceinline__ void q_p0i_compute(float *m, c2 *v, const c2 *__restrict__ twA, int vt)
{
    c2 w[32];
    tw_load(w, twA, TWA_STRIDE, vt);
    tw_dft32_inv(v, w);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const c2 x = v[xpos(i)];
        m[i] = fmaf(x.x, x.x, x.y * x.y);
    }
}
// junction of one virtual thread, one half (the J loop of xcorr_row14); the reference-spectrum values are loaded a half ahead
__device__ __forceinline__ void q_refspec_load(float4 *r, const float4 *__restrict__ refspec4, int vt, int h)
{
    const int g = ((vt >> 6) << 7) + 64 * h + (vt & 63);
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = refspec4[j * 1024 + g];
}
__device__ __forceinline__ void q_junction_half(float4 *A4, const float4 *r, int vt, int h)
{
    const int g = ((vt >> 6) << 7) + 64 * h + (vt & 63), key = g & 7;
    int base = j_base(g);
    asm volatile("" : "+v"(base));
    c2 u[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float4 q = A4[base + (j ^ key)];
        u[2 * j] = mk(q.x, q.y);
        u[2 * j + 1] = mk(q.z, q.w);
    }
    dft16p<-1>(u);
#if CRSDR_K1_FUSED_TW
    {
        c2 rr[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) { rr[2 * j] = mk(r[j].x, r[j].y); rr[2 * j + 1] = mk(r[j].z, r[j].w); }
        dft16_inv_mul(u, rr);
    }
#else
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        u[2 * j] = cmul(u[2 * j], mk(r[j].x, r[j].y));
        u[2 * j + 1] = cmul(u[2 * j + 1], mk(r[j].z, r[j].w));
    }
    dft16p<+1>(u);
#endif
#pragma unroll
    for (int j = 0; j < 8; ++j)
        A4[base + (j ^ key)] = make_float4(u[2 * j].x, u[2 * j].y, u[2 * j + 1].x, u[2 * j + 1].y);
}

__device__ __forceinline__ float q_wave_max63(float wm)
{
    wm = fmaxf(wm, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(wm), __float_as_int(wm), 0x111, 0xf, 0xf, false)));
    wm = fmaxf(wm, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(wm), __float_as_int(wm), 0x112, 0xf, 0xf, false)));
    wm = fmaxf(wm, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(wm), __float_as_int(wm), 0x114, 0xf, 0xf, false)));
    wm = fmaxf(wm, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(wm), __float_as_int(wm), 0x118, 0xf, 0xf, false)));
    wm = fmaxf(wm, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(wm), __float_as_int(wm), 0x142, 0xa, 0xf, false)));
    wm = fmaxf(wm, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(wm), __float_as_int(wm), 0x143, 0xc, 0xf, false)));
    return wm;      // valid in lane 63
}

// minimum of v over the 64 lanes (wave-uniform result): the same six DPP steps as q_wave_max63, then lane 63 read back
__device__ __forceinline__ int q_wave_min(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}
// (idx < 32 ? m0[idx] : m1[idx - 32]) for a WAVE-UNIFORM idx: a tree of scalar branches ending in one move.  (Written as an
// array subscript -- or as a tree of ?: the optimiser recognises as one -- the two arrays go to scratch memory; 64 compares and
// selects would do, at 64 vector instructions per wave and row.)  The asm keeps every leaf a plain register read.
__device__ __forceinline__ float q_pick(const float (&m0)[32], const float (&m1)[32], int idx)
{
    float r = 0.0f;
    switch (__builtin_amdgcn_readfirstlane(idx)) {
    case 0: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[0])); break;
    case 1: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[1])); break;
    case 2: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[2])); break;
    case 3: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[3])); break;
    case 4: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[4])); break;
    case 5: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[5])); break;
    case 6: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[6])); break;
    case 7: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[7])); break;
    case 8: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[8])); break;
    case 9: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[9])); break;
    case 10: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[10])); break;
    case 11: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[11])); break;
    case 12: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[12])); break;
    case 13: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[13])); break;
    case 14: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[14])); break;
    case 15: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[15])); break;
    case 16: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[16])); break;
    case 17: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[17])); break;
    case 18: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[18])); break;
    case 19: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[19])); break;
    case 20: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[20])); break;
    case 21: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[21])); break;
    case 22: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[22])); break;
    case 23: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[23])); break;
    case 24: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[24])); break;
    case 25: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[25])); break;
    case 26: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[26])); break;
    case 27: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[27])); break;
    case 28: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[28])); break;
    case 29: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[29])); break;
    case 30: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[30])); break;
    case 31: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m0[31])); break;
    case 32: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[0])); break;
    case 33: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[1])); break;
    case 34: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[2])); break;
    case 35: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[3])); break;
    case 36: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[4])); break;
    case 37: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[5])); break;
    case 38: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[6])); break;
    case 39: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[7])); break;
    case 40: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[8])); break;
    case 41: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[9])); break;
    case 42: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[10])); break;
    case 43: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[11])); break;
    case 44: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[12])); break;
    case 45: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[13])); break;
    case 46: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[14])); break;
    case 47: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[15])); break;
    case 48: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[16])); break;
    case 49: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[17])); break;
    case 50: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[18])); break;
    case 51: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[19])); break;
    case 52: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[20])); break;
    case 53: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[21])); break;
    case 54: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[22])); break;
    case 55: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[23])); break;
    case 56: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[24])); break;
    case 57: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[25])); break;
    case 58: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[26])); break;
    case 59: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[27])); break;
    case 60: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[28])); break;
    case 61: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[29])); break;
    case 62: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[30])); break;
    case 63: asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(m1[31])); break;
    default: break;
    }
    return r;
}

// grid: one workgroup per CU; items = owned rows x blocks, item (2k + g) * gridDim.x + blockIdx.x goes to group g
__global__ __launch_bounds__(2 * QG, 1) void k_xcorr_lag14q(XcorrArgs a, const float2 *__restrict__ twA_, const float2 *__restrict__ twB_,
                                                           int row_count, int *__restrict__ errflag, unsigned int *__restrict__ work,
                                                           unsigned int work_base, int spin_limit)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const c2 *twA = reinterpret_cast<const c2 *>(twA_), *twB = reinterpret_cast<const c2 *>(twB_);
    c2 *A = reinterpret_cast<c2 *>(smem);
    float4 *A4 = reinterpret_cast<float4 *>(smem);
    const int g = threadIdx.x >> 8, tid = threadIdx.x & (QG - 1);
    float *red = reinterpret_cast<float *>(smem + (size_t)LDS_ELEMS * 8 + QSCR_BYTES * g);      // [0..3] wave max, [4..7] its first index, [8..11] / [12..15] the neighbours
    int *redi = reinterpret_cast<int *>(red);
    float *edge = red + 16;                                                                      // [wave][first lane, last lane][half * 32 + i]
    QSync *sy = reinterpret_cast<QSync *>(smem + (size_t)LDS_ELEMS * 8 + 2 * QSCR_BYTES);
    if (threadIdx.x == 0) { sy->owner = 0; sy->relcnt = 0; sy->bar[0] = 0; sy->bar[1] = 0; sy->err = 0; sy->next[0] = 0; sy->next[1] = 0; sy->spin_limit = spin_limit; sy->arrive[0] = 0; sy->arrive[1] = 0; }
    __syncthreads();
    int gen = 0;
#ifdef CRSDR_QDEBUG
    const unsigned long long tk0__ = __builtin_readcyclecounter();
#endif
    const int nitems = row_count * a.nblocks;
    // Work order: each group starts on item g * gridDim.x + blockIdx.x and draws every further one from a global counter
    // (*work counts on from work_base; it advances by exactly nitems per launch -- every item after the first 2 * grid
    // is one draw, and every group that had an item makes one draw that comes back empty -- so the host never resets it).
    // A workgroup that gets its CU late (a collective's kernel, K0 or another process is resident there) then simply takes
    // fewer rows; with a fixed share per workgroup it would run its whole share after everybody else had finished.
    int item = g * (int)gridDim.x + (int)blockIdx.x;
    for (int k = 0;; ++k) {
        // opaque per iteration: otherwise every LDS / table offset derived from the two virtual thread ids is hoisted out
        // of the row loop and ~40 registers' worth of them live (spilled) across it
        int vt0 = tid, vt1 = tid + QG;
        asm volatile("" : "+v"(vt0), "+v"(vt1));
        if ((unsigned)item >= (unsigned)nitems) break;     // also ends on a negative item (host / device counters out of step)
        const int item_u = __builtin_amdgcn_readfirstlane(item);           // one item per group: scalar row / block addresses
        const int t = item_u / row_count, row = a.row_begin + item_u % row_count;
        if (xcorr_skip(a, row, t, tid)) {
            q_barrier(sy, g, gen, 5);           // every wave of the group has read this item from next[g] before it is overwritten
            if (tid == 0) sy->next[g] = 2 * (int)gridDim.x + (int)(atomicAdd(work, 1u) - work_base);
            q_barrier(sy, g, gen, 5);
            item = sy->next[g];                 // (the next write follows a group barrier on either path)
            continue;
        }
        const int8_t *src = a.rows + (size_t)t * a.block_stride + (size_t)row * N;
        const float4 *__restrict__ refspec4 = reinterpret_cast<const float4 *>(a.refspec) + (size_t)t * (N / 2);
        c2 wB[32];
        {
            c2 v[32], v2[32];
            q_p0_compute(v, src, twA, a.xor80, vt0);      // no LDS yet: both halves run beside the other group's middle section
            q_p0_compute(v2, src, twA, a.xor80, vt1);
            tw_load(wB, twB, TWB_STRIDE, tid & 15);       // P1 / P1' twiddles: table loads and product chain outside the ownership window
            q_acquire(sy, 2 * k + g + 1);
            __builtin_amdgcn_s_setprio(Q_PRIO);                // the owner's window is what the pair's period is made of
            q_p0_store(A, v, vt0);
            q_p0_store(A, v2, vt1);
        }
        float4 ra[8], rb[8];
        q_refspec_load(ra, refspec4, vt0, 0);
        q_barrier(sy, g, gen, 0);
        // the group's next item, drawn one row ahead: the atomic goes out here, its result is parked in LDS before the window's
        // second barrier and read by everybody after it (the next write follows the next row's first barrier on either path)
        int drawn = 0;
        if (tid == 0) drawn = 2 * (int)gridDim.x + (int)(atomicAdd(work, 1u) - work_base);
        pass1_forward(A, wB, vt0);
        pass1_forward(A, wB, vt1);
        wave_lds_sync();
        q_refspec_load(rb, refspec4, vt0, 1);
        q_junction_half(A4, ra, vt0, 0);
        q_refspec_load(ra, refspec4, vt1, 0);
        q_junction_half(A4, rb, vt0, 1);
        q_refspec_load(rb, refspec4, vt1, 1);
        q_junction_half(A4, ra, vt1, 0);
        q_junction_half(A4, rb, vt1, 1);
        wave_lds_sync();
        pass1_inverse(A, wB, vt0);
        pass1_inverse(A, wB, vt1);
        if (tid == 0) sy->next[g] = drawn;
        q_barrier(sy, g, gen, 1);
        float m0[32], m1[32];
        {
            c2 v[32], v2[32];
            q_p0i_load(v, A, vt0);
            q_p0i_load(v2, A, vt1);
            q_release(sy);                               // from here on this row is in registers
            __builtin_amdgcn_s_setprio(0);
            int vb0 = tid, vb1 = tid + QG;               // opaque again: the column-twiddle addresses of the first pass would otherwise be kept (spilled) for this one
            asm volatile("" : "+v"(vb0), "+v"(vb1));
            q_p0i_compute(m0, v, twA, vb0);
            q_p0i_compute(m1, v2, twA, vb1);
        }
        // ---- argmax without a group barrier ------------------------------------------------------------------------------------
        // Each wave reduces its own 64 x 64 magnitudes to a candidate -- maximum VALUE first (v_max3, six DPP steps), then the first
        // index at which its lanes hold it (natural index of output i of virtual thread vt: i * 512 + vt; volk_32f_index_max_32u keeps
        // the first strict maximum) -- and parks it in LDS with the peak's two neighbours where they sit in the same wave; the
        // magnitudes of the wave's first and last lane go to the edge table, where a peak on a wave edge finds its neighbour.  The
        // LAST of the four waves to arrive combines the candidates and publishes; nobody waits for anybody (r02: two group barriers per
        // row, 12 % of a wave's time, spun through beside the other group's window).  A wave cannot overwrite its candidate before
        // the finisher has read it: the next row's epilogue lies behind that row's two window barriers, which the finisher joins.
        float tm = fmaxf(m0[0], m1[0]);
#pragma unroll
        for (int i = 1; i < 32; ++i) tm = fmaxf(tm, fmaxf(m0[i], m1[i]));
        const int lane = tid & 63, wv = tid >> 6;
        const float wm = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q_wave_max63(tm)), 63));
        int bi = 0x7fffffff;
        if (tm == wm) {
#pragma unroll
            for (int i = 31; i >= 0; --i) {
                bi = (m1[i] == wm) ? i * 512 + vt1 : bi;
                bi = (m0[i] == wm) ? i * 512 + vt0 : bi;      // vt0 < vt1: the lower index wins at the same i
            }
        }
        const int wbi = q_wave_min(bi);                       // wave-uniform; 0x7fffffff: no lane compared equal (an all-NaN wave)
        if (lane == 0 || lane == 63) {
            float4 *e4 = reinterpret_cast<float4 *>(edge + (wv * 2 + (lane ? 1 : 0)) * 64);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                e4[i] = make_float4(m0[4 * i], m0[4 * i + 1], m0[4 * i + 2], m0[4 * i + 3]);
                e4[8 + i] = make_float4(m1[4 * i], m1[4 * i + 1], m1[4 * i + 2], m1[4 * i + 3]);
            }
        }
        {
            const int pcw = wbi & 511, piw = (wbi >> 9) & 31;
            const float mine = q_pick(m0, m1, (pcw >> 8) * 32 + piw);     // every lane's magnitude at the candidate's output index and half
            const float ym = __shfl_up(mine, 1, 64), yp = __shfl_down(mine, 1, 64);
            if (lane == (pcw & 63)) { red[wv] = wm; redi[4 + wv] = wbi; red[8 + wv] = ym; red[12 + wv] = yp; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        int arrived = 0;
        if (lane == 0) arrived = __hip_atomic_fetch_add(&sy->arrive[g], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        arrived = __builtin_amdgcn_readfirstlane(arrived);
        if ((arrived & 3) == 3) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            float gm = red[0];
#pragma unroll
            for (int w = 1; w < QG / 64; ++w) gm = fmaxf(gm, red[w]);
            int gi = 0x7fffffff;
#pragma unroll
            for (int w = 0; w < QG / 64; ++w) gi = (red[w] == gm) ? min(gi, redi[4 + w]) : gi;
            if ((unsigned)gi >= (unsigned)N) gi = 0;          // all-NaN row: defined as index 0
            const int pc = gi & 511, ow = (pc & (QG - 1)) >> 6, ln = pc & 63;
            float ym = red[8 + ow], yp = red[12 + ow];        // the owner wave's own lanes pc -+ 1 ...
            if (ln == 0 && gi > 0) {                          // ... unless the peak sits on a wave edge: the neighbour is another wave's last / first lane
                const int nl = gi - 1, cl = nl & 511;
                ym = edge[(((cl & (QG - 1)) >> 6) * 2 + 1) * 64 + (cl >> 8) * 32 + (nl >> 9)];
            }
            if (ln == 63 && gi < N - 1) {
                const int nr = gi + 1, cr = nr & 511;
                yp = edge[(((cr & (QG - 1)) >> 6) * 2 + 0) * 64 + (cr >> 8) * 32 + (nr >> 9)];
            }
            float D = 0.0f;
            if (gi > 0 && gi < N - 1) {
                const float den = (ym - 2.0f * gm) + yp;
                if (den != 0.0f) D = (0.5f * (ym - yp)) / den;
            }
            if (lane == 0) xcorr_publish(a, row, t, gi - L /* src/ccoherent.cc:232 */, sqrtf(gm / (float)L) * kInvScale2 /* :204 */, D);
        }
        item = sy->next[g];       // written before this row's second barrier; rewritten behind the next row's first
    }
#ifdef CRSDR_QDEBUG
    if (dbg__ && blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicAdd(dbg__ + 0, (unsigned long long)(__builtin_readcyclecounter() - tk0__));
#endif
    if (threadIdx.x == 0 && errflag) {
        // both groups may still be running: the flag is only ever set, so reading it here can miss a late error of the
        // other group -- thread 256 reports as well
        if (__hip_atomic_load(&sy->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) atomicAdd(errflag, 1);
    }
    if (threadIdx.x == QG && errflag) {
        if (__hip_atomic_load(&sy->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) atomicAdd(errflag, 1);
    }
}

// ---- stage B of the long-block path (longblock.hpp) on the same two-lines-per-CU structure -----------------------------
// k_rows14_cf32p<false> costs a CU its line's load phase + ~10 us of transforms + its store phase one after the other (one
// 132 KiB image per CU, so no second workgroup hides them): 18.6 us per 128 KiB line.  Here a line's loads, first radix-32
// pass, last pass and stores are the OUTER section of a group and run beside the other group's image-owning middle section,
// exactly as the int8 loads and the epilogue do in k_xcorr_lag14q.  cf32 in and out, in place; forward, x conj(ref), inverse.
// items = lines: item i is line i of Y (row i / n1, frequency k1 = i % n1 -> its slice of the reference spectrum).
__device__ __forceinline__ void q_line_p0_transform(c2 *v, const c2 *__restrict__ twA, int vt)
{
    dft32<-1>(v);
    c2 w[32];
    tw_load(w, twA, TWA_STRIDE, vt);
    tw_apply<-1, true, 1>(v, w);
}
// the column pair (2m, 2m + 1) of a line / of the image as 16 bytes per lane
__device__ __forceinline__ void q_line_load2(c2 *v, c2 *v2, const c2 *__restrict__ line, int m)
{
    const float4 *l4 = reinterpret_cast<const float4 *>(line);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const float4 x = l4[i * 256 + m];
        v[i] = mk(x.x, x.y);
        v2[i] = mk(x.z, x.w);
    }
}
__device__ __forceinline__ void q_p0_store2(c2 *A, const c2 *v, const c2 *v2, int m)
{
    int base = p0_base(2 * m);            // even: the pair is 16-byte aligned in the image
    asm volatile("" : "+v"(base));
#pragma unroll
    for (int k = 0; k < 32; ++k)
        *reinterpret_cast<float4 *>(A + base + k * 528) = make_float4(v[xpos(k)].x, v[xpos(k)].y, v2[xpos(k)].x, v2[xpos(k)].y);
}
__device__ __forceinline__ void q_p0i_load2(c2 *v, c2 *v2, const c2 *A, int m)
{
    int base = p0_base(2 * m);
    asm volatile("" : "+v"(base));
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const float4 x = *reinterpret_cast<const float4 *>(A + base + k * 528);
        v[k] = mk(x.x, x.y);
        v2[k] = mk(x.z, x.w);
    }
}
__device__ __forceinline__ void q_line_p0i_transform(c2 *v, const c2 *__restrict__ twA, int vt)
{
    c2 w[32];
    tw_load(w, twA, TWA_STRIDE, vt);
    tw_dft32_inv(v, w);
}
__device__ __forceinline__ void q_line_store2(c2 *__restrict__ line, const c2 *v, const c2 *v2, int m)
{
    float4 *l4 = reinterpret_cast<float4 *>(line);
#pragma unroll
    for (int i = 0; i < 32; ++i) l4[i * 256 + m] = make_float4(v[xpos(i)].x, v[xpos(i)].y, v2[xpos(i)].x, v2[xpos(i)].y);      // natural order, coalesced
}

// Work order (static): line (row, k1) needs the k1-th 128 KiB slice of the reference spectrum, 16 MiB in all -- more than an XCD's
// L2 holds, so with lines dealt out in memory order every line fetched its slice from the memory-side cache: a third of the
// stage's traffic (measured with every line reading slice 0: 94 -> 83 us per launch).  Hence one queue per XCD: workgroup b runs
// on XCD b % 8 (round-robin dispatch) and takes lines with k1 % 8 == b % 8, k1-major, so that the rows of one k1 pass through
// one L2 at about the same time and its slice is fetched once per launch and XCD.  (The mapping only matters for speed: whatever
// XCD a workgroup really runs on, every line is done exactly once.)  nq = 8 when n1 and the grid are multiples of 8, else 1.
// RAMP (the apply pass of crsdr_plan_set_frac_apply): the "spectrum" is the row's own response G (k_ramp_rowspec: one 128 KiB
// slice per row of the launch) and a line's inputs are scaled by its c(k1) -- the same kernel otherwise.
template <bool RAMP>
__global__ __launch_bounds__(2 * QG, 1) void k_rows14_cf32q(c2 *__restrict__ Y, const c2 *__restrict__ twA, const c2 *__restrict__ twB,
                                                           const float4 *__restrict__ refspec_base, int n1, int rows, int nq, int *__restrict__ errflag,
                                                           int spin_limit, RampArgs ra, c2 *__restrict__ Yout)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c2 *A = reinterpret_cast<c2 *>(smem);
    float4 *A4 = reinterpret_cast<float4 *>(smem);
    const int g = threadIdx.x >> 8, tid = threadIdx.x & (QG - 1);
    QSync *sy = reinterpret_cast<QSync *>(smem + (size_t)LDS_ELEMS * 8 + 2 * QSCR_BYTES);
    if (threadIdx.x == 0) { sy->owner = 0; sy->relcnt = 0; sy->bar[0] = 0; sy->bar[1] = 0; sy->err = 0; sy->next[0] = 0; sy->next[1] = 0; sy->spin_limit = spin_limit; sy->arrive[0] = 0; sy->arrive[1] = 0; }
    __syncthreads();
    int gen = 0;
    const int q = (int)blockIdx.x % nq, nlocal = 2 * ((int)gridDim.x / nq);           // this workgroup's queue; groups working on it
    const int lg = g * ((int)gridDim.x / nq) + (int)blockIdx.x / nq;                  // this group's place among them
    const int nqueue = (n1 / nq) * rows;                                              // lines in the queue, k1-major
    for (int k = 0;; ++k) {
        int vt0 = tid, vt1 = tid + QG;
        asm volatile("" : "+v"(vt0), "+v"(vt1));
        const int j = lg + k * nlocal;
        if (j >= nqueue) break;
        const int item_u = __builtin_amdgcn_readfirstlane((j % rows) * n1 + (j / rows) * nq + q);      // line index in Y: row * n1 + k1 (scalar base addresses)
        c2 *line = Y + (size_t)item_u * N;
        c2 *line_out = Yout ? Yout + (size_t)item_u * N : line;      // out of place when the caller keeps the forward column transforms (the apply pass reuses them)
#ifdef CRSDR_B_SPEC0      // diagnostic: every line reads spectrum slice 0 (wrong results; what the slices' traffic costs)
        const float4 *__restrict__ refspec4 = refspec_base;
#else
        const float4 *__restrict__ refspec4 = refspec_base + (size_t)(RAMP ? item_u / n1 : item_u % n1) * (N / 2);
#endif
        c2 wB[32];
        {
            c2 v[32], v2[32];
            // the outer passes work on the column PAIR (2 tid, 2 tid + 1) instead of (tid, tid + 256): neighbours in the line and
            // in the image (p0_base(2m + 1) = p0_base(2m) + 1), so a pair travels as 16 bytes per lane -- global loads / stores
            // and the image stores / loads of P0 / P0' are half as many instructions, twice as wide.  The image itself does not
            // change, so the middle section keeps its columns (tid, tid + 256).  (Time per launch unchanged, 94.2 against 94.7 us,
            // as it was with both halves' loads issued up front, 97.0: this stage is not waiting for its memory instructions.)
            const int pc0 = 2 * vt0, pc1 = 2 * vt0 + 1;
            q_line_load2(v, v2, line, vt0);
            if constexpr (RAMP) {
                const c2 cl = ramp_line_scale(ra, ra.row0 + item_u / n1, (uint32_t)(item_u % n1));
#pragma unroll
                for (int i = 0; i < 32; ++i) { v[i] = cmul(v[i], cl); v2[i] = cmul(v2[i], cl); }
            }
            __builtin_amdgcn_sched_barrier(0);
            q_line_p0_transform(v, twA, pc0);
            __builtin_amdgcn_sched_barrier(0);
            q_line_p0_transform(v2, twA, pc1);
            __builtin_amdgcn_sched_barrier(0);
            tw_load(wB, twB, TWB_STRIDE, tid & 15);
            q_acquire(sy, 2 * k + g + 1);
            __builtin_amdgcn_s_setprio(Q_PRIO);
            q_p0_store2(A, v, v2, vt0);
        }
        float4 ra[8], rb[8];
        q_refspec_load(ra, refspec4, vt0, 0);
        q_barrier(sy, g, gen, 0);
        pass1_forward(A, wB, vt0);
        pass1_forward(A, wB, vt1);
        wave_lds_sync();
        q_refspec_load(rb, refspec4, vt0, 1);
        q_junction_half(A4, ra, vt0, 0);
        q_refspec_load(ra, refspec4, vt1, 0);
        q_junction_half(A4, rb, vt0, 1);
        q_refspec_load(rb, refspec4, vt1, 1);
        q_junction_half(A4, ra, vt1, 0);
        q_junction_half(A4, rb, vt1, 1);
        wave_lds_sync();
        pass1_inverse(A, wB, vt0);
        pass1_inverse(A, wB, vt1);
        q_barrier(sy, g, gen, 1);
        {
            c2 v[32], v2[32];
            q_p0i_load2(v, v2, A, vt0);
            q_release(sy);                               // the line is in registers: the image goes to the other group
            __builtin_amdgcn_s_setprio(0);
            int vs0 = tid;                               // opaque again: store addresses kept from the loads would be 64 registers
            asm volatile("" : "+v"(vs0));
            q_line_p0i_transform(v, twA, 2 * vs0);
            __builtin_amdgcn_sched_barrier(0);
            q_line_p0i_transform(v2, twA, 2 * vs0 + 1);
            __builtin_amdgcn_sched_barrier(0);
            q_line_store2(line_out, v, v2, vs0);
        }
    }
    if (threadIdx.x == 0 && errflag) {
        if (__hip_atomic_load(&sy->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) atomicAdd(errflag, 1);
    }
    if (threadIdx.x == QG && errflag) {
        if (__hip_atomic_load(&sy->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) atomicAdd(errflag, 1);
    }
}

} // namespace x14p
} // namespace crsdr
