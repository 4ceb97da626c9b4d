// k2_pair.hpp -- the fused phase kernel with TWO rows per workgroup (r03 experiment, measured slower, not in the product).
// Parity-green on the first run (bit-identical to k_align_fused in tests/test_gpu_plan.py::test_phase_path_variants_are_bit_identical,
// odd slabs included); A/B in one call, locked cadence, T = 64 (profiles/r03/k2_pair_ab.log):
//     one row per workgroup   0.443 ms per launch = 4.85 TB/s read+write   (59 VGPRs, 8 waves per SIMD)
//     row pairs (this file)   0.531 ms            = 4.05 TB/s              (92 VGPRs, 5 waves per SIMD)
// and the dot product's new three-instruction form alone changed nothing (4.85 against 4.84 with the six-instruction form): the
// kernel is not bound by vector-instruction issue (r03's own premise, from 829 instructions per wave and row) but by how much of a
// workgroup's life it has memory requests in flight -- fewer, fatter workgroups lengthen the part without any.  To include it:
// paste into csrc/kernels.hpp before k_assemble_slabs and launch with grid (1 + ceil(rows / 2)) * nblocks.

// ---- K2 fused, TWO rows per workgroup (B == 16384) ----------------------------------------------------------------------------
// The same kernel for the row pair (2p, 2p + 1) of the plan's slab: a thread's four 16-byte vectors of the REFERENCE row -- and their
// swapped / masked forms for the integer dot product -- serve both rows, and everything a workgroup pays once (block / row index
// arithmetic, address forming, the scalar-register traffic of two large argument blocks, three barriers) is paid per pair instead of
// per row.  Waves 0 and 1 run the phasor chain of rows A and B side by side (the code of k_align_fused with the row taken from the
// wave index).  Same integers, same chain arithmetic, same rotation: bit-identical to k_align_fused (test_phase_path_variants_are_
// bit_identical).  The locked cadence is bound by vector-instruction issue as much as by HBM (r02: 829 VALU instructions per wave and
// row at ~87 % issue occupancy), so instructions per row are what its bandwidth is made of.
// Grid: (1 + ceil(rows / 2)) * nblocks workgroups, block-major like k_align_fused: (pair, t) depends on lower indices only.
template <bool XOR>
__global__ __launch_bounds__(kAlignThreads, 5) void k_align_fused2(AlignArgs a_, FusedSync fs)
{
    AlignArgs a = a_;
    a.xor80 = XOR ? a_.xor80 : 0u;
    constexpr int B = 16384, L = B >> 1, nvec = B / 16;
    __shared__ int sred[4 * (kAlignThreads / 64)];      // per wave: re / im of row A, re / im of row B (|sum| < 2^28: exact in 32 bits)
    __shared__ float2 sp[2];
    __shared__ unsigned long long smiss[2], sfix[2][64];
    __shared__ int sstar[2];
    const int tid = threadIdx.x;
    const unsigned int npairs = ((unsigned)fs.row_count + 1u) >> 1, per = npairs + 1u;
    const unsigned int ticket = blockIdx.x;
    const int t = (int)(ticket / per), x = (int)(ticket % per);
    const size_t moff = 16 + 4 * (size_t)a.nrows;
    const int8_t *blk = a.rows + (size_t)t * a.block_stride;
    int8_t *packet = a.packet + (size_t)t * a.packet_stride;
    if (a.slab) {
        if (x == 0 && (t < a.hdr_first || t >= a.hdr_first + a.hdr_count)) return;
        packet = a.packet + (size_t)(t - a.hdr_first) * a.packet_stride;
    }
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef u32x4 u32x4_u2 __attribute__((aligned(2)));
    if (x == 0) {
        // header hdr0{globalseqn,N,L,unused} + readcnt words + the raw reference row (src/cpacketizer.cc:112-116,137-156)
        uint32_t *h = reinterpret_cast<uint32_t *>(packet);
        const uint32_t seq = a.seq + (uint32_t)t;
        if (tid == 0) { h[0] = seq; h[1] = (uint32_t)a.nrows; h[2] = (uint32_t)L; h[3] = 0u; }
        for (int r = tid; r < a.nrows; r += kAlignThreads) h[4 + r] = a.readcnt ? a.readcnt[(size_t)t * a.nrows + r] : seq;
        const uint4 *src = reinterpret_cast<const uint4 *>(blk);
        uint4 *dst = reinterpret_cast<uint4 *>(packet + moff);
        for (int i = tid; i < nvec; i += kAlignThreads) {
            const uint4 v = src[i];
            dst[i] = make_uint4(v.x ^ a.xor80, v.y ^ a.xor80, v.z ^ a.xor80, v.w ^ a.xor80);
        }
        return;
    }
    const int rowA = a.row_begin + 2 * (x - 1);
    const bool hasB = 2 * (x - 1) + 1 < fs.row_count;               // an odd slab ends in a pair of one
    const int rowB = hasB ? rowA + 1 : rowA;                          // (row B then repeats row A's loads and is never stored)
    const int dA = align_shift(a, rowA, t), dB = align_shift(a, rowB, t);
    const int8_t *srowA = blk + (size_t)rowA * B, *srowB = blk + (size_t)rowB * B;
    // all twelve 16-byte loads of a thread go out back to back: the reference row's four (they do not wait for a lag), then the
    // rows' own.  An interior vector is ONE load at a 2-byte-aligned address; one that straddles or lies outside [0,L) loads from the
    // row start instead and is patched afterwards (k_align_fused)
    uint4 rv[4], sa[4], sb[4];
    if (a.refnoise) {
        const uint4 *r128 = reinterpret_cast<const uint4 *>(blk);
#pragma unroll
        for (int q = 0; q < 4; ++q) rv[q] = r128[tid + q * kAlignThreads];
    }
    bool edgeA = false, edgeB = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int mA = 8 * (tid + q * kAlignThreads) + dA, mB = 8 * (tid + q * kAlignThreads) + dB;
        const bool inA = mA >= 0 && mA + 8 <= L, inB = mB >= 0 && mB + 8 <= L;
        const u32x4 ua = __builtin_nontemporal_load(reinterpret_cast<const u32x4_u2 *>(srowA + 2 * (ptrdiff_t)(inA ? mA : 0)));
        const u32x4 ub = __builtin_nontemporal_load(reinterpret_cast<const u32x4_u2 *>(srowB + 2 * (ptrdiff_t)(inB ? mB : 0)));
        sa[q] = make_uint4(ua.x, ua.y, ua.z, ua.w);
        sb[q] = make_uint4(ub.x, ub.y, ub.z, ub.w);
        edgeA |= !inA; edgeB |= !inB;
    }
    if (edgeA) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * kAlignThreads, m0 = 8 * i + dA;
            if (!(m0 >= 0 && m0 + 8 <= L)) { const uint4 e = shifted_vec(srowA, i, dA, L, a.xor80); sa[q] = make_uint4(e.x ^ a.xor80, e.y ^ a.xor80, e.z ^ a.xor80, e.w ^ a.xor80); }
        }
    }
    if (edgeB) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * kAlignThreads, m0 = 8 * i + dB;
            if (!(m0 >= 0 && m0 + 8 <= L)) { const uint4 e = shifted_vec(srowB, i, dB, L, a.xor80); sb[q] = make_uint4(e.x ^ a.xor80, e.y ^ a.xor80, e.z ^ a.xor80, e.w ^ a.xor80); }
        }
    }
    // waves 0 / 1 are the chain waves of rows A / B: they start what their fold will need together with the row loads
    const int cw = tid >> 6, lane = tid & 63;
    const bool chainwave = cw == 0 || (cw == 1 && hasB);
    const int crow = cw == 0 ? rowA : rowB;
    const size_t oc = (size_t)t * a.nrows + crow;
    float2 p_in = make_float2(0.f, 0.f);
    unsigned long long bits = 0ull, cv = kChainEmpty;
    if (chainwave) {
        p_in = a.phase_in[crow];
        if (a.refnoise && lane < t) {
            bits = fs.spin_limit < 0 ? kChainEmpty : __hip_atomic_load(fs.chain + 2 * ((size_t)lane * a.nrows + crow), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (fs.spin_limit >= 0) cv = __hip_atomic_load(fs.chainv + 2 * ((size_t)lane * a.nrows + crow), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sa[q] = make_uint4(sa[q].x ^ a.xor80, sa[q].y ^ a.xor80, sa[q].z ^ a.xor80, sa[q].w ^ a.xor80);
        sb[q] = make_uint4(sb[q].x ^ a.xor80, sb[q].y ^ a.xor80, sb[q].z ^ a.xor80, sb[q].w ^ a.xor80);
    }
    if (a.refnoise) {
        int reA = 0, crA = 0, nqA = 0, reB = 0, crB = 0, nqB = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t rw[4] = {rv[q].x ^ a.xor80, rv[q].y ^ a.xor80, rv[q].z ^ a.xor80, rv[q].w ^ a.xor80};
            const uint32_t wa[4] = {sa[q].x, sa[q].y, sa[q].z, sa[q].w}, wb[4] = {sb[q].x, sb[q].y, sb[q].z, sb[q].w};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t rs = ref_swap(rw[w]), rsq = rs & 0x00FF00FFu;       // once per reference word, for both rows
                dot_word3(wa[w], rw[w], rs, rsq, reA, crA, nqA);
                dot_word3(wb[w], rw[w], rs, rsq, reB, crB, nqB);
            }
        }
        const int s0 = wave_sum_lane63(reA), s1 = wave_sum_lane63(crA - 2 * nqA), s2 = wave_sum_lane63(reB), s3 = wave_sum_lane63(crB - 2 * nqB);
        if (lane == 63) { sred[4 * cw] = s0; sred[4 * cw + 1] = s1; sred[4 * cw + 2] = s2; sred[4 * cw + 3] = s3; }
    }
    __syncthreads();
    auto unit_bits = [](long long sr, long long si) -> unsigned long long {
        if (sr == 0 && si == 0) return 0ull;
        const double cr = (double)sr, ci = (double)si;
        const double inv = 1.0 / sqrt(cr * cr + ci * ci);
        return (unsigned long long)__float_as_uint((float)(cr * inv)) | ((unsigned long long)__float_as_uint((float)(-ci * inv)) << 32);
    };
    if (chainwave) {
        if (lane == 0 && !a.xcorr_ran) {   // no lag measured in this batch: republish the carried one (include/csdrdevice.h:161)
            a.lag_out[oc] = a.lag_state[crow]; a.mag_out[oc] = a.mag_state[crow]; a.frac_out[oc] = a.frac_state[crow];
        }
        if (a.refnoise) {
            unsigned long long mine = 0ull;
            if (lane == 0) {
                long long sr = 0, si = 0;
                for (int w = 0; w < kAlignThreads / 64; ++w) { sr += sred[4 * w + 2 * cw]; si += sred[4 * w + 2 * cw + 1]; }
                mine = unit_bits(sr, si);
                __hip_atomic_store(fs.chain + 2 * oc, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (fs.rearm) fs.rearm[2 * oc] = kChainEmpty;
            }
            int ustar = -1;
            {
                const unsigned long long *srcu = fs.chain + 2 * ((size_t)lane * a.nrows + crow);
                const unsigned long long *srcv = fs.chainv + 2 * ((size_t)lane * a.nrows + crow);
                int spins = 0;
                for (;;) {
                    const unsigned long long have = __ballot(lane < t && cv != kChainEmpty);
                    ustar = have ? 63 - __builtin_clzll(have) : -1;
                    const bool need = lane < t && lane > ustar && bits == kChainEmpty;
                    if (!__ballot(need) || spins >= fs.spin_limit) break;
                    if (need) bits = __hip_atomic_load(srcu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (lane < t && lane > ustar && cv == kChainEmpty) cv = __hip_atomic_load(srcv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __builtin_amdgcn_s_sleep(8);
                    ++spins;
                }
            }
            const unsigned long long own = __shfl(mine, 0, 64);
            if (lane == t) bits = own;
            const unsigned long long missing = __ballot(lane < t && lane > ustar && bits == kChainEmpty);
            if (lane == 0) { smiss[cw] = missing; sstar[cw] = ustar; }
        } else if (lane == 0) { smiss[cw] = 0ull; sstar[cw] = -1; }
    } else if (cw == 1 && lane == 0) { smiss[1] = 0ull; sstar[1] = -1; }
    __syncthreads();
    // Fallback, normally never taken (k_align_fused): an earlier block's unit phasor did not arrive within the poll budget -- the
    // whole workgroup forms that block's dot product itself, for row A's missing blocks and then for row B's
#pragma unroll 1
    for (int rr = 0; rr < 2; ++rr) {
        const int frow = rr ? rowB : rowA;
        for (unsigned long long miss = smiss[rr]; miss != 0ull; miss &= miss - 1ull) {
            const int u = __builtin_ctzll(miss);
            const int8_t *blku = a.rows + (size_t)u * a.block_stride;
            const int du = align_shift(a, frow, u);
            const uint4 *r128 = reinterpret_cast<const uint4 *>(blku);
            int re = 0, im = 0;
            for (int i = tid; i < nvec; i += kAlignThreads) {
                const uint4 s = shifted_vec(blku + (size_t)frow * B, i, du, L, a.xor80);
                const uint4 rvu = r128[i];
                dot_word(s.x, rvu.x ^ a.xor80, re, im);
                dot_word(s.y, rvu.y ^ a.xor80, re, im);
                dot_word(s.z, rvu.z ^ a.xor80, re, im);
                dot_word(s.w, rvu.w ^ a.xor80, re, im);
            }
            re = wave_sum_lane63(re);
            im = wave_sum_lane63(im);
            __syncthreads();                       // sred / sfix of the previous round are consumed
            if (lane == 63) { sred[4 * cw] = re; sred[4 * cw + 1] = im; }
            __syncthreads();
            if (tid == 0) {
                long long sr = 0, si = 0;
                for (int w = 0; w < kAlignThreads / 64; ++w) { sr += sred[4 * w]; si += sred[4 * w + 1]; }
                sfix[rr][u] = unit_bits(sr, si);
                atomicAdd(fs.status, 1u);          // counted, not an error: how often the fallback ran
            }
            __syncthreads();
        }
    }
    if (chainwave) {
        float2 p = p_in;
        if (a.refnoise) {
            const int ustar = sstar[cw];           // wave-uniform: the block whose chain value the fold starts from (-1: the carried phasor)
            if (lane < t && lane > ustar && bits == kChainEmpty) bits = sfix[cw][lane];
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
                    p = make_float2(__fadd_rn(__fmul_rn(0.5f, pr), __fmul_rn(0.5f, p.x)), __fadd_rn(__fmul_rn(0.5f, pi), __fmul_rn(0.5f, p.y)));
                }
            }
        }
        if (lane == 0) {
            a.phasor[oc] = p;                                   // get_phasecorrect() after block t
            if (t == a.nblocks - 1) a.phase_out[crow] = p;      // state carried to the next batch
            sp[cw] = p;
            if (a.refnoise) {
                if (t < a.nblocks - 1)
                    __hip_atomic_store(fs.chainv + 2 * oc, (unsigned long long)__float_as_uint(p.x) | ((unsigned long long)__float_as_uint(p.y) << 32),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (fs.rearmv) fs.rearmv[2 * oc] = kChainEmpty;
            }
        }
    }
    __syncthreads();
    const float2 pA = sp[0];
    int8_t *orowA = a.slab ? a.slab + (size_t)t * a.slab_stride + (size_t)(rowA - a.row_begin) * B : packet + moff + (size_t)rowA * B;
    uint4 *oA = reinterpret_cast<uint4 *>(orowA);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 v = make_uint4(rotq_word(sa[q].x, pA), rotq_word(sa[q].y, pA), rotq_word(sa[q].z, pA), rotq_word(sa[q].w, pA));
        if (a.nt & 1) __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4 *>(oA + tid + q * kAlignThreads));
        else oA[tid + q * kAlignThreads] = v;
    }
    if (hasB) {
        const float2 pB = sp[1];
        uint4 *oB = reinterpret_cast<uint4 *>(orowA + B);      // row B is the next row of the matrix / the slab
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 v = make_uint4(rotq_word(sb[q].x, pB), rotq_word(sb[q].y, pB), rotq_word(sb[q].z, pB), rotq_word(sb[q].w, pB));
            if (a.nt & 1) __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4 *>(oB + tid + q * kAlignThreads));
            else oB[tid + q * kAlignThreads] = v;
        }
    }
}

