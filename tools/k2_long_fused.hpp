// tools/k2_long_fused.hpp -- DEAD END 16 (r03), kept for the record; not part of the product.
// The long rows' phase path (one block per submit) as ONE kernel instead of k_phase_dot + k_align_quant.  Bit-identical to the
// three-kernel path in a variants test (B = 2^15 ... 2^21, track / locked / no reference noise, also with the wait cut to zero polls),
// and slower on cfg5 (21 rows x 2 MiB, A/B in one call: bench.py --cfg5 with the kernel switched on and off, two rounds):
//     two kernels                           23.9 + 21.1 = 45 us   2 540 - 2 580 blocks/s
//     this kernel, 32 KiB chunks (118 VGPRs)        54 us          2 480 - 2 500
//     this kernel, 16 KiB chunks ( 86 VGPRs)     88 - 91 us        2 270 - 2 290
// and on the way: with a release fence between the sums and the arrival 187 us (the fence writes back the XCD's whole L2, i.e. the other
// workgroups' freshly stored rows, once per workgroup); without fences (returning atomics + vmcnt(0)) but with every row's {re, im,
// arrivals} packed into two cache lines 102 us (every waiting workgroup of the launch polls the same lines); one 128-byte record per
// row: 54.  What it saves -- the second read of 44 MB that the memory-side cache serves anyway, and one launch boundary -- is less than
// what a thousand workgroups parked on a counter cost: they hold their CU slots with nothing in flight, and the launch degenerates into
// a load phase for everybody followed by a store phase for everybody.
// To rebuild: paste behind rotq_word in kernels.hpp; launch k_align_long_fused<8>(aa, row_count, B / 32768, records, 16, status, polls) on
// (row_count + 1) * (B / 32768) workgroups with the records zeroed (k_long_finalize or a memset).
#pragma once

// ---- long rows, one block per submit: k_phase_dot + k_align_quant as ONE kernel, the row read once -------------------------
// A long row (B > 16 KiB) is cut into 32 KiB chunks, one workgroup each: 8 x 16 B of the row per thread stay in registers while
// the chunk's exact integer dot product goes into the row's accumulator (two 64-bit atomics) and the workgroup waits for the row's
// other chunks -- an arrival counter next to the accumulators -- before it forms the phasor (the single chain step of a one-block
// batch, as k_align_quant's inline_chain does: same operands, same operations) and rotates what it holds.  Saves the second read of
// the rows and a launch boundary (cfg5: 23 + 20 us of kernels -> one).
//   * work order = workgroup index, row-major: the chunks of a row are dispatched together, so a row's workgroups only ever wait
//     for workgroups that were dispatched with them (a row has at most 128 chunks; the device holds > 1000 workgroups);
//   * nothing depends on the wait: after `spin_limit` polls a workgroup forms the row's whole dot product itself (same integers),
//     counted in *status like the fused kernel's look-back fallback.
// Grid: row_count * nchunk workgroups for the rows, then nchunk for header + readcnt + the raw reference row.
// rec[rec_stride * row + {0, 1, 2}] = {sum re, sum im, arrivals} are zero at launch (k_long_finalize, or the host's memset).
template <int U>
__global__ __launch_bounds__(kAlignThreads) void k_align_long_fused(AlignArgs a, int row_count, int nchunk, unsigned long long *rec, int rec_stride,
                                                                    unsigned int *status, int spin_limit)
{
    __shared__ long long sred[2 * (kAlignThreads / 64)];
    __shared__ float2 sp;
    __shared__ int sfall;
    const int tid = threadIdx.x;
    const int B = a.B, L = B >> 1;
    const size_t moff = 16 + 4 * (size_t)a.nrows;
    const int8_t *blk = a.rows;                                   // one block per submit (t = 0)
    int8_t *packet = a.packet;
    const unsigned int id = blockIdx.x;
    const int x = (int)(id / (unsigned)nchunk), z = (int)(id % (unsigned)nchunk);
    const int v_lo = (int)(((long long)(B / 16) * z) / nchunk), v_hi = (int)(((long long)(B / 16) * (z + 1)) / nchunk);
    if (x >= row_count) {
        if (a.slab && (0 < a.hdr_first || 0 >= a.hdr_first + a.hdr_count)) return;      // a sharded plan that does not root this block
        uint32_t *h = reinterpret_cast<uint32_t *>(packet);
        if (z == 0) {
            // header hdr0{globalseqn,N,L,unused} src/cpacketizer.cc:112-116 and readcnt words :142,163
            if (tid == 0) { h[0] = a.seq; h[1] = (uint32_t)a.nrows; h[2] = (uint32_t)L; h[3] = 0u; }
            for (int r = tid; r < a.nrows; r += kAlignThreads) h[4 + r] = a.readcnt ? a.readcnt[r] : a.seq;
        }
        const uint4 *src = reinterpret_cast<const uint4 *>(blk);
        uint4 *dst = reinterpret_cast<uint4 *>(packet + moff);
        for (int i = v_lo + tid; i < v_hi; i += kAlignThreads) {
            const uint4 v = src[i];
            dst[i] = make_uint4(v.x ^ a.xor80, v.y ^ a.xor80, v.z ^ a.xor80, v.w ^ a.xor80);
        }
        return;
    }
    const int row = a.row_begin + x;
    const int d = align_shift(a, row, 0);
    const int8_t *srow = blk + (size_t)row * B;
    const uint4 *r128 = reinterpret_cast<const uint4 *>(blk);
    const int i0 = v_lo + tid;                                    // the chunk is exactly U * kAlignThreads vectors (host-checked)
    uint4 sv[U];
    // the row's record, alone in its 128-byte line ({re, im} packed 16 bytes apart per row with all arrival counters in one line, every
    // waiting workgroup of the launch polled the same two lines: 102 us against 45 for the two kernels this one replaces)
    unsigned long long *c = rec + (size_t)rec_stride * row, *arrive = c + 2;
    if (a.refnoise) {
        int re = 0, cr = 0, nq = 0;
        {
            uint4 rv[U];
#pragma unroll
            for (int q = 0; q < U; ++q) rv[q] = r128[i0 + q * kAlignThreads];
            shifted_vecs<U>(sv, srow, i0, kAlignThreads, v_hi, d, L, a.xor80);
#pragma unroll
            for (int q = 0; q < U; ++q) {
                dot_word3(sv[q].x, rv[q].x ^ a.xor80, re, cr, nq);
                dot_word3(sv[q].y, rv[q].y ^ a.xor80, re, cr, nq);
                dot_word3(sv[q].z, rv[q].z ^ a.xor80, re, cr, nq);
                dot_word3(sv[q].w, rv[q].w ^ a.xor80, re, cr, nq);
            }
        }
        long long acc_re = re, acc_im = cr - 2 * nq;              // per thread: <= 8 * 4 words * 2^16: exact in 32 bits
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            acc_re += __shfl_xor(acc_re, off, 64);
            acc_im += __shfl_xor(acc_im, off, 64);
        }
        if ((tid & 63) == 0) { sred[2 * (tid >> 6)] = acc_re; sred[2 * (tid >> 6) + 1] = acc_im; }
        __syncthreads();
        if (tid == 0) {
            long long sr = 0, si = 0;
            for (int w = 0; w < kAlignThreads / 64; ++w) { sr += sred[2 * w]; si += sred[2 * w + 1]; }
            // integer partial sums: the order of the adds does not matter
            // RETURNING agent-scope atomics, performed at the memory side: once both have come back (vmcnt(0)) the sums are in place, and
            // only then does the arrival go out.  No fence: a release fence here writes back the XCD's whole L2 -- the other workgroups'
            // freshly stored rows -- once per workgroup (measured: 187 us for this kernel against 45 for the two it replaces).
            const unsigned long long o0 = __hip_atomic_fetch_add(c, (unsigned long long)sr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long o1 = __hip_atomic_fetch_add(c + 1, (unsigned long long)si, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" : : "v"(o0), "v"(o1) : "memory");
            __hip_atomic_fetch_add(arrive, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)nchunk && spins < spin_limit) {
                __builtin_amdgcn_s_sleep(32);
                ++spins;
            }
            sfall = __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)nchunk ? 1 : 0;
        }
        __syncthreads();
    } else {
        shifted_vecs<U>(sv, srow, i0, kAlignThreads, v_hi, d, L, a.xor80);
        if (tid == 0) sfall = 0;
        __syncthreads();
    }
    long long tot_re = 0, tot_im = 0;                             // thread 0's
    if (sfall) {
        // a chunk of this row has not arrived within the poll budget (its workgroup was never dispatched beside this one: another
        // process shares the GPU, ...): the whole row's dot product here, then this chunk's vectors once more
        int re = 0, cr = 0, nq = 0;
        for (int j0 = tid; j0 < B / 16; j0 += 4 * kAlignThreads) {
            uint4 fv[4], rv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) rv[q] = r128[min(j0 + q * kAlignThreads, B / 16 - 1)];
            shifted_vecs<4>(fv, srow, j0, kAlignThreads, B / 16, d, L, a.xor80);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (j0 + q * kAlignThreads < B / 16) {
                    dot_word3(fv[q].x, rv[q].x ^ a.xor80, re, cr, nq);
                    dot_word3(fv[q].y, rv[q].y ^ a.xor80, re, cr, nq);
                    dot_word3(fv[q].z, rv[q].z ^ a.xor80, re, cr, nq);
                    dot_word3(fv[q].w, rv[q].w ^ a.xor80, re, cr, nq);
                }
        }
        long long acc_re = re, acc_im = (long long)cr - 2 * (long long)nq;      // <= 2^13 words per thread at B = 2^22: int32 partials are safe
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            acc_re += __shfl_xor(acc_re, off, 64);
            acc_im += __shfl_xor(acc_im, off, 64);
        }
        __syncthreads();
        if ((tid & 63) == 0) { sred[2 * (tid >> 6)] = acc_re; sred[2 * (tid >> 6) + 1] = acc_im; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 0; w < kAlignThreads / 64; ++w) { tot_re += sred[2 * w]; tot_im += sred[2 * w + 1]; }
            atomicAdd(status, 1u);                                // counted, not an error
        }
        shifted_vecs<U>(sv, srow, i0, kAlignThreads, v_hi, d, L, a.xor80);
    } else if (tid == 0 && a.refnoise) {
        tot_re = (long long)__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tot_im = (long long)__hip_atomic_load(c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) {
        // the single chain step (k_phase_chain / k_align_quant's inline_chain): csdrdevice::est_phasecorrect, src/csdrdevice.cc:58-69
        float2 p = a.phase_in[row];
        if (a.refnoise && (tot_re != 0 || tot_im != 0)) {
            const double cr = (double)tot_re, ci = (double)tot_im;
            const double inv = 1.0 / sqrt(cr * cr + ci * ci);
            const float pr = (float)(cr * inv), pi = (float)(-ci * inv);
            p = make_float2(__fadd_rn(__fmul_rn(0.5f, pr), __fmul_rn(0.5f, p.x)), __fadd_rn(__fmul_rn(0.5f, pi), __fmul_rn(0.5f, p.y)));
        }
        sp = p;
        if (z == 0) {
            a.phasor[row] = p;
            a.phase_out[row] = p;
            if (!a.xcorr_ran) { a.lag_out[row] = a.lag_state[row]; a.mag_out[row] = a.mag_state[row]; a.frac_out[row] = a.frac_state[row]; }
        }
    }
    __syncthreads();
    const float2 p = sp;
    int8_t *orow = a.slab ? a.slab + (size_t)(row - a.row_begin) * B : packet + moff + (size_t)row * B;
    uint4 *o128 = reinterpret_cast<uint4 *>(orow);
#pragma unroll
    for (int q = 0; q < U; ++q)
        o128[i0 + q * kAlignThreads] = make_uint4(rotq_word(sv[q].x, p), rotq_word(sv[q].y, p), rotq_word(sv[q].z, p), rotq_word(sv[q].w, p));
}

