// plan_args.hpp -- kernel argument blocks of the batched plan (plain structs passed by value).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crsdr {

// K1 (cross-correlation): grid = (owned signal rows, T blocks)
struct XcorrArgs {
    const int8_t *rows;     // batch base: block t at rows + t*block_stride, layout [nrows][B]
    size_t block_stride;    // bytes
    const float2 *refspec;  // [T][B] conj reference spectra of this batch (kernel-specific order)
    const uint8_t *lag_mask; // [nrows] or nullptr = every owned row
    int row_begin, nrows, nblocks, stagger;
    uint32_t xor80;
    int32_t *lag;           // [T][nrows] per-block outputs
    float *mag, *frac;
    int32_t *lag_state;     // [nrows] last known lag per row (carried across batches)
    float *mag_state, *frac_state;
    // B = 16384 kernels with the reference spectra folded in (fold = 1): the launch's first work items / workgroups transform the
    // blocks' reference rows themselves (what k_ref_spectrum14p does on the aux stream otherwise), store conj(spectrum) through
    // refspec_w and publish refflag[t] = refgen; a signal row of block t waits for that word before it reads the spectrum.
    int fold = 0;
    float2 *refspec_w = nullptr;
    unsigned int *refflag = nullptr;   // [T] device words, monotonic: refgen of the last launch that published block t's spectrum
    unsigned int refgen = 0;
    int *errflag = nullptr;            // a bounded wait ran out (the two-row kernel's word: the host rolls the plan back)
};

// wave-level wait for a published word (every lane leaves with the same verdict); false = the poll budget ran out.
// `early` is what the wave's first lane read from the word at the START of its row (word_peek): a load whose latency has long
// been paid by the time the row needs the verdict -- only the first rows of a launch ever find the word not yet there and poll.
__device__ __forceinline__ unsigned int word_peek(const unsigned int *word)
{
    unsigned int v = 0u;
    if ((threadIdx.x & 63) == 0) v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}
__device__ __forceinline__ bool wait_word(const unsigned int *word, unsigned int want, unsigned int early, int budget)
{
    unsigned int v = (unsigned int)__builtin_amdgcn_readfirstlane((int)early);
    int spins = 0;
    while (v != want) {
        if (++spins > budget) return false;
        __builtin_amdgcn_s_sleep(16);      // ~0.5 us between polls: one poller per workgroup, and it must not crowd the publisher's own traffic out of the word's L2 channel
        unsigned int w = want;
        if ((threadIdx.x & 63) == 0) w = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = (unsigned int)__builtin_amdgcn_readfirstlane((int)w);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // what the publisher stored before the word is visible to this wave's loads
    // the L1 invalidate completes asynchronously: the OTHER waves of the workgroup load behind the barrier this wave joins next, so it
    // waits the invalidate out first (MI355X_MICROARCH.md, inter-workgroup visibility: poll -> acquire -> vmcnt(0) -> barrier -> loads)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return true;
}
constexpr int kRefWaitBudget = 1 << 20;      // polls of >= 0.3 us: a reference item takes ~10 us, this is a third of a second

// A row that did not request a lag keeps its previous lag/mag/frac (src/ccoherent.cc:266: only
// is_lagrequested() devices are queued): the workgroup just republishes the state and leaves.
__device__ __forceinline__ bool xcorr_skip(const XcorrArgs &a, int row, int t, int tid)
{
    if (!a.lag_mask || a.lag_mask[row]) return false;
    if (tid == 0) {
        const size_t o = (size_t)t * a.nrows + row;
        a.lag[o] = a.lag_state[row];
        a.mag[o] = a.mag_state[row];
        a.frac[o] = a.frac_state[row];
    }
    return true;
}

__device__ __forceinline__ void xcorr_publish(const XcorrArgs &a, int row, int t, int lag, float mag, float frac)
{
    const size_t o = (size_t)t * a.nrows + row;
    a.lag[o] = lag;
    a.mag[o] = mag;
    a.frac[o] = frac;
    if (t == a.nblocks - 1) { // what csdrdevice::set_lag leaves in lagp (include/csdrdevice.h:138-151)
        a.lag_state[row] = lag;
        a.mag_state[row] = mag;
        a.frac_state[row] = frac;
    }
}

// K2a / K2b (phase estimate, rotate, quantise): grid = (rows, T blocks)
struct AlignArgs {
    const int8_t *rows;       // batch base, block t at rows + t*block_stride
    size_t block_stride;
    int8_t *packet;           // block t at packet + t*packet_stride: hdr + readcnt + matrix
    size_t packet_stride;
    const uint32_t *readcnt;  // device [T][nrows] or nullptr (use seq + t)
    const uint8_t *lag_mask;  // [nrows] or nullptr
    const int32_t *lag;       // [T][nrows] (K1 outputs of this batch)
    const int32_t *lag_state; // [nrows]
    long long *corr;          // [T][nrows][2] exact integer correlation sums
    const float2 *phase_in;   // [nrows] csdrdevice::phasecorr before this batch
    float2 *phase_out;        // [nrows] ... after it (double-buffered: other blocks still read phase_in)
    float2 *phasor;           // [T][nrows] get_phasecorrect() after each block
    int nrows, B, row_begin, nblocks;
    int digital, refnoise, xcorr_ran;
    int inline_chain;         // one-block batches of long rows: k_align_quant folds the block's phasor itself (no k_phase_chain launch)
    int nt;                   // bit 0: non-temporal stores of the output rows, bit 1: non-temporal loads of the signal rows (both are touched once)
    uint32_t seq, xor80;
    // slab output (sharded plans, crsdr_plan_bind_slab): when slab != nullptr the owned rows of block t go to
    // slab + t*slab_stride (row_begin first, B-byte pitch) instead of the packet matrix, and header + readcnt +
    // row 0 are written only for blocks hdr_first <= t < hdr_first + hdr_count, into packet (t - hdr_first)
    int8_t *slab;
    size_t slab_stride;
    int hdr_first, hdr_count;
    // per-block {lag, mag, frac} outputs and the carried state: a batch whose cross-correlation did not run
    // (CRSDR_NO_LAG / nothing requested) republishes the state for every block (k_phase_chain)
    int32_t *lag_out;
    float *mag_out, *frac_out;
    const float *mag_state, *frac_state;
};

} // namespace crsdr
