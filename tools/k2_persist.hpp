// tools/k2_persist.hpp -- DEAD END 15 (r03), kept for the record: k_align_fused as PERSISTENT workgroups (B = 16384, 16-byte aligned rows).
// Not part of the product.  To rebuild the experiment: include this file from kernels.hpp behind k_align_fused and launch
// k_align_fused_p<xor>(aa, fs, total items) on min(items, CUs x n) workgroups.  Bit-identical to k_align_fused in
// test_phase_path_variants_are_bit_identical (also with every look-back wait cut short and with 1 or 2 workgroups per CU), and slower:
// locked cadence, 64-block launches, A/B in one call (profiles/r03/k2_persist_ab.log): 3.10 TB/s at 3 workgroups per CU, 3.59 at 4,
// 3.64 at 5 (96 VGPRs with 10 - 18 spilled) against 4.76 for the one-shot kernel at 8.  An item takes a resident workgroup ~9 us --
// one memory round trip under load, whatever is prefetched one item ahead -- so the bytes in flight per CU are what counts, and
// 99 VGPRs (the prefetched row, the edge-patch and fallback code kept live across the loop) allow half the workgroups the one-shot
// kernel's 59 do.  What would be needed is the prefetch without registers (LDS-DMA into 16 KiB per workgroup, eight workgroups per CU)
// and look-back polls that do not queue behind it on the in-order vector-memory counter (scalar loads): not built.
//
//
// k_align_fused runs one workgroup per (row, block): load phase -> dot product, two barriers, the chain wave's serial section
// (nothing outstanding) -> stores.  What bounds it (dead end 13) is the part of a workgroup's life without
// memory requests in flight, not instruction issue.  Here a workgroup walks the items w, w + G, w + 2G, ... of the same
// block-major order and issues the NEXT item's signal-row loads (HBM, 4 x 16 B per thread) as soon as the current item's dot
// product has let go of the reference row's registers, so every resident workgroup has 16 KiB of HBM reads outstanding through
// its silent part.  The reference row of an item comes
// from L2 and is loaded when the item starts.
//
// Order and progress: a workgroup takes its items in increasing order and item (row, t) depends only on items (row, u < t), i.e.
// on LOWER indices; all G workgroups are resident (G <= what the device holds), so the lowest unfinished item is always the next
// one of some resident workgroup and has nothing left to wait for.  The bounded look-back wait with its local fallback is kept.
// Results: the same integers, the same phasors, the same rounding sequence -- bit-identical to k_align_fused
// (tests/test_gpu_plan.py::test_phase_path_variants_are_bit_identical).
#pragma once

template <bool XOR>
__global__ __launch_bounds__(kAlignThreads, 4) void k_align_fused_p(AlignArgs a_, FusedSync fs, unsigned int total)
{
    AlignArgs a = a_;
    a.xor80 = XOR ? a_.xor80 : 0u;                 // a compile-time zero without XOR
    __shared__ long long sred[2 * (kAlignThreads / 64)];
    __shared__ float2 sp;
    __shared__ unsigned long long smiss, sfix[64];
    __shared__ int sstar;
    int tid = threadIdx.x;
    const unsigned int per = (unsigned)fs.row_count + 1u;
    constexpr int B = 16384, L = B >> 1, nvec = B / 16;
    const size_t moff = 16 + 4 * (size_t)a.nrows;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef u32x4 u32x4_u2 __attribute__((aligned(2)));

    // the four signal-row vectors of an item, issued back to back (k_align_fused: an interior vector is ONE load at a 2-byte
    // aligned address, one that straddles or lies outside [0, L) loads the row start and is patched when the item is processed).
    // A reference item (x == 0) has no signal row: it loads the first owned row's start and drops it.
    auto issue = [&](unsigned int ticket, uint4 (&v)[4], int &d) {
        const int t = (int)(ticket / per), x = (int)(ticket % per);
        const int row = a.row_begin + max(x, 1) - 1;
        d = x ? align_shift(a, row, t) : 0;
        const int8_t *srow = a.rows + (size_t)t * a.block_stride + (size_t)row * B;
        if (a.nt & 2) {            // ONE uniform branch around all four loads
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m0 = 8 * (tid + q * kAlignThreads) + d;
                const bool inside = m0 >= 0 && m0 + 8 <= L;
                const u32x4 u = __builtin_nontemporal_load(reinterpret_cast<const u32x4_u2 *>(srow + 2 * (ptrdiff_t)(inside ? m0 : 0)));
                v[q] = make_uint4(u.x, u.y, u.z, u.w);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m0 = 8 * (tid + q * kAlignThreads) + d;
                const bool inside = m0 >= 0 && m0 + 8 <= L;
                const u4_unaligned u = *reinterpret_cast<const u4_unaligned *>(srow + 2 * (ptrdiff_t)(inside ? m0 : 0));
                v[q] = make_uint4(u.x, u.y, u.z, u.w);
            }
        }
    };
    // unit phasor conj(corr)/|corr| from the integer sums; 0 = "|corr| == 0, hold the previous phasor"
    auto unit_bits = [](long long sr, long long si) -> unsigned long long {
        if (sr == 0 && si == 0) return 0ull;
        const double cr = (double)sr, ci = (double)si;
        const double inv = 1.0 / sqrt(cr * cr + ci * ci);
        return (unsigned long long)__float_as_uint((float)(cr * inv)) | ((unsigned long long)__float_as_uint((float)(-ci * inv)) << 32);
    };

    unsigned int ticket = blockIdx.x;
    if (ticket >= total) return;
    uint4 sv[4];
    int d;
    issue(ticket, sv, d);
    for (;;) {
        // opaque per item: otherwise every address offset derived from the thread index is hoisted out of the loop and kept across it
        // (111 VGPRs against the one-shot kernel's 59)
        asm volatile("" : "+v"(tid));
        const unsigned int next = ticket + gridDim.x;
        const bool more = next < total;
        const int t = (int)(ticket / per), x = (int)(ticket % per);
        const int8_t *blk = a.rows + (size_t)t * a.block_stride;
        int8_t *packet = a.packet + (size_t)t * a.packet_stride;
        bool skip = false;                                   // a sharded plan writes headers for the blocks it roots only
        if (a.slab) {
            skip = x == 0 && (t < a.hdr_first || t >= a.hdr_first + a.hdr_count);
            packet = a.packet + (size_t)(t - a.hdr_first) * a.packet_stride;
        }
        // this item's reference row (L2) first, then the NEXT item's signal row (HBM): both before anything waits
        uint4 rv[4];
        if ((a.refnoise || x == 0) && !skip) {
            const uint4 *r128 = reinterpret_cast<const uint4 *>(blk);
#pragma unroll
            for (int q = 0; q < 4; ++q) rv[q] = r128[tid + q * kAlignThreads];
        }
        uint4 nv[4];
        int nd;
        if (x == 0) {
            issue(more ? next : ticket, nv, nd);             // (the last item reloads its own row: a branch here would split the load clause)
            if (!skip) {
                // header hdr0{globalseqn,N,L,unused} + readcnt words + the raw reference row (src/cpacketizer.cc:112-116,137-156)
                uint32_t *h = reinterpret_cast<uint32_t *>(packet);
                const uint32_t seq = a.seq + (uint32_t)t;
                if (tid == 0) { h[0] = seq; h[1] = (uint32_t)a.nrows; h[2] = (uint32_t)L; h[3] = 0u; }
                for (int r = tid; r < a.nrows; r += kAlignThreads) h[4 + r] = a.readcnt ? a.readcnt[(size_t)t * a.nrows + r] : seq;
                uint4 *dst = reinterpret_cast<uint4 *>(packet + moff);
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[tid + q * kAlignThreads] = make_uint4(rv[q].x ^ a.xor80, rv[q].y ^ a.xor80, rv[q].z ^ a.xor80, rv[q].w ^ a.xor80);
            }
        } else {
            const int row = a.row_begin + x - 1;
            const size_t o = (size_t)t * a.nrows + row;
            const int8_t *srow = blk + (size_t)row * B;
            {
                bool edge = false;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m0 = 8 * (tid + q * kAlignThreads) + d;
                    edge |= !(m0 >= 0 && m0 + 8 <= L);
                }
                if (edge) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = tid + q * kAlignThreads, m0 = 8 * i + d;
                        if (!(m0 >= 0 && m0 + 8 <= L)) {
                            const uint4 e = shifted_vec(srow, i, d, L, a.xor80);
                            sv[q] = make_uint4(e.x ^ a.xor80, e.y ^ a.xor80, e.z ^ a.xor80, e.w ^ a.xor80);
                        }
                    }
                }
            }
            // wave 0 starts what the fold will need: the carried phasor and what earlier blocks of this row have published
            float2 p_in = make_float2(0.f, 0.f);
            unsigned long long bits = 0ull;          // wave 0, lane u <= t: unit phasor of block u of this row
            unsigned long long cv = kChainEmpty;     // wave 0, lane u < t: the chain value after block u, where its workgroup has got that far
            if (tid < 64) {
                p_in = a.phase_in[row];
                if (a.refnoise && tid < t) {
                    bits = fs.spin_limit < 0 ? kChainEmpty
                                             : __hip_atomic_load(fs.chain + 2 * ((size_t)tid * a.nrows + row), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (fs.spin_limit >= 0) cv = __hip_atomic_load(fs.chainv + 2 * ((size_t)tid * a.nrows + row), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) sv[q] = make_uint4(sv[q].x ^ a.xor80, sv[q].y ^ a.xor80, sv[q].z ^ a.xor80, sv[q].w ^ a.xor80);
            if (a.refnoise) {
                int re = 0, cr = 0, nq = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    dot_word3(sv[q].x, rv[q].x ^ a.xor80, re, cr, nq);
                    dot_word3(sv[q].y, rv[q].y ^ a.xor80, re, cr, nq);
                    dot_word3(sv[q].z, rv[q].z ^ a.xor80, re, cr, nq);
                    dot_word3(sv[q].w, rv[q].w ^ a.xor80, re, cr, nq);
                }
                int im = cr - 2 * nq;
                re = wave_sum_lane63(re);
                im = wave_sum_lane63(im);
                if ((tid & 63) == 63) { sred[2 * (tid >> 6)] = re; sred[2 * (tid >> 6) + 1] = im; }
            }
            // the reference row's registers are free again: the next item's signal row goes out here and flies through the barriers,
            // the chain wave's serial section, the rotation and the stores (issued before the dot product it cost 43 spilled VGPRs at
            // six workgroups per CU)
            issue(more ? next : ticket, nv, nd);
            __syncthreads();
            if (tid < 64) {
                if (tid == 0 && !a.xcorr_ran) {   // no lag measured in this batch: republish the carried one (include/csdrdevice.h:161)
                    a.lag_out[o] = a.lag_state[row]; a.mag_out[o] = a.mag_state[row]; a.frac_out[o] = a.frac_state[row];
                }
                if (a.refnoise) {
                    unsigned long long mine = 0ull;
                    if (tid == 0) {
                        long long sr = 0, si = 0;
                        for (int w = 0; w < kAlignThreads / 64; ++w) { sr += sred[2 * w]; si += sred[2 * w + 1]; }
                        mine = unit_bits(sr, si);
                        __hip_atomic_store(fs.chain + 2 * o, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (fs.rearm) fs.rearm[2 * o] = kChainEmpty;    // nobody reads this slot before the next launch
                    }
                    int ustar = -1;
                    {
                        const unsigned long long *srcu = fs.chain + 2 * ((size_t)tid * a.nrows + row);
                        const unsigned long long *srcv = fs.chainv + 2 * ((size_t)tid * a.nrows + row);
                        int spins = 0;
                        for (;;) {
                            const unsigned long long have = __ballot(tid < t && cv != kChainEmpty);
                            ustar = have ? 63 - __builtin_clzll(have) : -1;
                            const bool need = tid < t && tid > ustar && bits == kChainEmpty;
                            if (!__ballot(need) || spins >= fs.spin_limit) break;
                            if (need) bits = __hip_atomic_load(srcu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (tid < t && tid > ustar && cv == kChainEmpty) cv = __hip_atomic_load(srcv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __builtin_amdgcn_s_sleep(8);
                            ++spins;
                        }
                    }
                    const unsigned long long own = __shfl(mine, 0, 64);
                    if (tid == t) bits = own;
                    const unsigned long long missing = __ballot(tid < t && tid > ustar && bits == kChainEmpty);
                    if (tid == 0) { smiss = missing; sstar = ustar; }
                } else if (tid == 0) { smiss = 0ull; sstar = -1; }
            }
            __syncthreads();
            // Fallback, normally never taken (see k_align_fused): the whole workgroup forms a missing block's dot product itself
            for (unsigned long long miss = smiss; miss != 0ull; miss &= miss - 1ull) {
                const int u = __builtin_ctzll(miss);
                const int8_t *blku = a.rows + (size_t)u * a.block_stride;
                const int du = align_shift(a, row, u);
                const uint4 *r128 = reinterpret_cast<const uint4 *>(blku);
                int re = 0, im = 0;
                for (int i = tid; i < nvec; i += kAlignThreads) {
                    const uint4 s = shifted_vec(blku + (size_t)row * B, i, du, L, a.xor80);
                    const uint4 rr = r128[i];
                    dot_word(s.x, rr.x ^ a.xor80, re, im);
                    dot_word(s.y, rr.y ^ a.xor80, re, im);
                    dot_word(s.z, rr.z ^ a.xor80, re, im);
                    dot_word(s.w, rr.w ^ a.xor80, re, im);
                }
                long long acc_re = re, acc_im = im;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    acc_re += __shfl_xor(acc_re, off, 64);
                    acc_im += __shfl_xor(acc_im, off, 64);
                }
                __syncthreads();                       // sred / sfix of the previous round are consumed
                if ((tid & 63) == 0) { sred[2 * (tid >> 6)] = acc_re; sred[2 * (tid >> 6) + 1] = acc_im; }
                __syncthreads();
                if (tid == 0) {
                    long long sr = 0, si = 0;
                    for (int w = 0; w < kAlignThreads / 64; ++w) { sr += sred[2 * w]; si += sred[2 * w + 1]; }
                    sfix[u] = unit_bits(sr, si);
                    atomicAdd(fs.status, 1u);          // counted, not an error: how often the fallback ran
                }
                __syncthreads();
            }
            if (tid < 64) {
                float2 p = p_in;
                if (a.refnoise) {
                    const int ustar = sstar;           // wave-uniform: the block whose chain value the fold starts from (-1: the carried phasor)
                    if (tid < t && tid > ustar && bits == kChainEmpty) bits = sfix[tid];
                    if (ustar >= 0) {
                        const unsigned sl = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(cv & 0xffffffffull), ustar),
                                       sh = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(cv >> 32), ustar);
                        p = make_float2(__uint_as_float(sl), __uint_as_float(sh));
                    }
                    const int blo = (int)(unsigned)(bits & 0xffffffffull), bhi = (int)(unsigned)(bits >> 32);
                    for (int u = ustar + 1; u <= t; ++u) {     // one sequential fold in block order (src/csdrdevice.cc:66-67)
                        const unsigned rl = (unsigned)__builtin_amdgcn_readlane(blo, u), rh = (unsigned)__builtin_amdgcn_readlane(bhi, u);
                        if ((rl | rh) != 0u) {
                            const float pr = __uint_as_float(rl), pi = __uint_as_float(rh);
                            p = make_float2(__fadd_rn(__fmul_rn(0.5f, pr), __fmul_rn(0.5f, p.x)),
                                            __fadd_rn(__fmul_rn(0.5f, pi), __fmul_rn(0.5f, p.y)));
                        }
                    }
                }
                if (tid == 0) {
                    a.phasor[o] = p;                                    // get_phasecorrect() after block t
                    if (t == a.nblocks - 1) a.phase_out[row] = p;       // state carried to the next batch
                    sp = p;
                    if (a.refnoise) {                                   // later blocks of this row may start their fold here
                        if (t < a.nblocks - 1)
                            __hip_atomic_store(fs.chainv + 2 * o, (unsigned long long)__float_as_uint(p.x) | ((unsigned long long)__float_as_uint(p.y) << 32),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (fs.rearmv) fs.rearmv[2 * o] = kChainEmpty;
                    }
                }
            }
            __syncthreads();
            const float2 p = sp;
            int8_t *orow = a.slab ? a.slab + (size_t)t * a.slab_stride + (size_t)(row - a.row_begin) * B : packet + moff + (size_t)row * B;
            uint4 *o128 = reinterpret_cast<uint4 *>(orow);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = tid + q * kAlignThreads;
                const uint4 v = make_uint4(rotq_word(sv[q].x, p), rotq_word(sv[q].y, p), rotq_word(sv[q].z, p), rotq_word(sv[q].w, p));
                if (a.nt & 1) __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4 *>(o128 + i));   // written once, read by nobody here
                else o128[i] = v;
            }
        }
        if (!more) break;
        ticket = next;
        d = nd;
#pragma unroll
        for (int q = 0; q < 4; ++q) sv[q] = nv[q];
    }
}
